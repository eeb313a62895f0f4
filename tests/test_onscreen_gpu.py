"""The reference's live call shape -- render_voxels with offscreen == false (src/renderer/cuda_renderer.cpp:141-142): a per-pixel
t_max from the depth attachment (renderer_kernel.cu:277-280) and a composite over the pixel already in the image (:230-234, :260-264)
-- through mnv_render_voxels_ex / mnv_render_voxels_accel_ex, both kernels, against the oracle (bit for bit), the reference's own
device code (goldens + the live build) and through the eleven-parameter binding of include/mnv_reference_binding.hpp."""
import os

import numpy as np
import pytest

import cases
from test_parity_gpu import require_live_reference

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _setup(mnv, name):
    base, _, _ = cases.ONSCREEN[name]
    spec = cases.CASES[base]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    tmax, image = cases.onscreen_inputs(name, cam)
    return tree, cam, opt, tmax, image


def _dev(torch, a):
    return None if a is None else torch.from_numpy(a).cuda()


@pytest.mark.parametrize("name", list(cases.ONSCREEN))
@pytest.mark.parametrize("which", ["ref_layout", "ref_layout_table", "ref_layout_walk", "accel", "tree_cache"])
def test_onscreen_inputs_bit_exact_vs_oracle(mnv, orc, torch_gpu, name, which):
    torch = torch_gpu
    tree, cam, opt, tmax, image = _setup(mnv, name)
    want = orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt, want_rgba8=True, tmax_px=tmax, rgba8_init=image)
    plain = orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt)["rgba"]
    assert (want["rgba"] != plain).any(axis=-1).mean() > 0.05, "the inputs of this case change too few pixels to test anything"
    tree.move_to_device()
    h, w = cam.height, cam.width
    rgba = torch.full((h, w, 4), float("nan"), dtype=torch.float32, device="cuda")
    rgba8 = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    t_dev, i_dev = _dev(torch, tmax), _dev(torch, image)
    if which == "accel":
        mnv.render_voxels_accel(tree.accel, cam, opt, rgba=rgba, rgba8=rgba8, tmax_px=t_dev, rgba8_init=i_dev)
    else:
        if which == "tree_cache":
            mnv.set_tree_cache(True)
        try:
            mnv.render_voxels(tree.device_view(), cam, opt, rgba=rgba, rgba8=rgba8, tmax_px=t_dev, rgba8_init=i_dev,
                              table_min_rays={"ref_layout": None, "ref_layout_table": 0, "ref_layout_walk": -1, "tree_cache": None}[which])
            torch.cuda.synchronize()
        finally:
            mnv.set_tree_cache(False)
    torch.cuda.synchronize()
    got, got8 = rgba.cpu().numpy(), rgba8.cpu().numpy()
    assert not np.isnan(got).any()
    assert np.array_equal(cases.bits(got), cases.bits(want["rgba"])), f"{int((cases.bits(got) != cases.bits(want['rgba'])).any(axis=-1).sum())} pixels differ"
    assert np.array_equal(got8, want["rgba8"])


@pytest.mark.parametrize("which", ["ref_layout", "accel"])
def test_onscreen_image_in_place(mnv, orc, torch_gpu, which):
    """The reference reads and writes ONE surface: rgba8_init may be the output buffer."""
    torch = torch_gpu
    tree, cam, opt, tmax, image = _setup(mnv, "onscreen_both")
    want = orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt, want_rgba8=True, tmax_px=tmax, rgba8_init=image)["rgba8"]
    tree.move_to_device()
    buf = torch.from_numpy(image).cuda()
    if which == "accel":
        mnv.render_voxels_accel(tree.accel, cam, opt, rgba8=buf, tmax_px=_dev(torch, tmax), rgba8_init=buf)
    else:
        mnv.render_voxels(tree.device_view(), cam, opt, rgba8=buf, tmax_px=_dev(torch, tmax), rgba8_init=buf)
    torch.cuda.synchronize()
    assert np.array_equal(buf.cpu().numpy(), want)


def test_onscreen_tile_and_null_members(mnv, orc, torch_gpu):
    """Inputs are indexed like the outputs (tile-relative); a struct with both members NULL is the offscreen call."""
    torch = torch_gpu
    tree, cam, opt, tmax, image = _setup(mnv, "onscreen_both")
    tile = (40, 24, 131, 77)
    x0, y0, w, h = tile
    t_tile = np.ascontiguousarray(tmax[y0:y0 + h, x0:x0 + w])
    i_tile = np.ascontiguousarray(image[y0:y0 + h, x0:x0 + w])
    t = orc.tree_from_view(tree.host_view())
    want = orc.render(t, cam.c, opt, tile=tile, tmax_px=t_tile, rgba8_init=i_tile)["rgba"]
    full = orc.render(t, cam.c, opt, tmax_px=tmax, rgba8_init=image)["rgba"]
    assert np.array_equal(cases.bits(want), cases.bits(full[y0:y0 + h, x0:x0 + w]))
    tree.move_to_device()
    for fn, arg in ((mnv.render_voxels, tree.device_view()), (mnv.render_voxels_accel, tree.accel)):
        out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
        fn(arg, cam, opt, tile=tile, rgba=out, tmax_px=_dev(torch, t_tile), rgba8_init=_dev(torch, i_tile))
        torch.cuda.synchronize()
        assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(want))
    import ctypes as C
    empty = mnv.FrameInputs(None, None)
    plain = orc.render(t, cam.c, opt)["rgba"]
    out = torch.empty((cam.height, cam.width, 4), dtype=torch.float32, device="cuda")
    rc = mnv.lib().mnv_render_voxels_accel_ex(C.c_void_p(tree.accel), C.byref(cam.c), C.byref(opt), mnv.Rect(0, 0, cam.width, cam.height),
                                              C.byref(empty), C.c_void_p(out.data_ptr()), None, None)
    torch.cuda.synchronize()
    assert rc == 0 and np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(plain))


@pytest.mark.parametrize("name", list(cases.ONSCREEN))
def test_onscreen_against_reference_goldens_and_live_build(mnv, orc, torch_gpu, tmp_path, name):
    """HIP vs the frames the reference's own device code produced for the same inputs (committed goldens; and again live when
    oracle/_ref is present): <= 1e-6, no outliers (contract 1e-4)."""
    torch = torch_gpu
    tree, cam, opt, tmax, image = _setup(mnv, name)
    g = np.load(os.path.join(GOLD, f"ref_{name}.npz"))["rgba"]
    tree.move_to_device()
    out = torch.empty((cam.height, cam.width, 4), dtype=torch.float32, device="cuda")
    mnv.render_voxels_accel(tree.accel, cam, opt, rgba=out, tmax_px=_dev(torch, tmax), rgba8_init=_dev(torch, image))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.abs(got.astype(np.float64) - g.astype(np.float64)).max() <= 1e-6
    mnv_ref = require_live_reference()
    path = str(tmp_path / "t.npz")
    tree.save_npz(path)
    live = mnv_ref.render_onscreen_npz(path, cam.c, opt, tmax_px=tmax, rgba8_init=image)
    assert np.array_equal(cases.bits(live), cases.bits(g)), "the live reference build no longer reproduces the committed golden"


@pytest.mark.parametrize("path_kind", [0, 1])
@pytest.mark.parametrize("offscreen", [False, True])
def test_eleven_parameter_binding_is_a_drop_in(mnv, orc, torch_gpu, tmp_path, path_kind, offscreen):
    """viewer::render_voxels(tree, cam, opt, image_arr, depth_arr, stream, to_split, to_sample, visited, track_visit, offscreen) as
    include/mnv_reference_binding.hpp declares it, compiled inside a build of the reference: the reference's loader, N3Tree and Camera
    feed libmnv.so; the image comes back equal to the oracle's RGBA8 frame byte for byte, offscreen honoured both ways."""
    mnv_ref = require_live_reference()
    name = "onscreen_both"
    base = cases.ONSCREEN[name][0]
    spec = cases.CASES[base]
    tree = cases.make_tree(mnv, spec["tree"])
    cs = spec["camera"]
    cam = cases.make_camera(mnv, cs)
    opt = cases.make_options(mnv, spec["options"])
    opt.basis_minmax[0], opt.basis_minmax[1] = 0, max(tree.host_view().basis_dim - 1, 0)
    tmax, image = cases.onscreen_inputs(name, cam)
    path = str(tmp_path / "t.npz")
    tree.save_npz(path)
    t = orc.tree_from_view(tree.host_view())
    if offscreen:
        want = orc.render(t, cam.c, opt, want_rgba8=True)["rgba8"]
    else:
        want = orc.render(t, cam.c, opt, want_rgba8=True, tmax_px=tmax, rgba8_init=image)["rgba8"]
    cam_spec = dict(width=cs["width"], height=cs["height"], fx=cs["fx"], center=cs["center"], back=cs["back"], up=cs.get("up", (0.0, 0.0, 1.0)))
    got = mnv_ref.dropin_onscreen_npz(path, cam_spec, opt, image, tmax, path=path_kind, offscreen=offscreen)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("need_viewdir,embedding", [(True, 7), (False, -1)])
def test_guided_samples_with_the_depth_image_bit_exact_on_both_layouts(mnv, orc, torch_gpu, need_viewdir, embedding):
    """get_samples_from_voxels with offscreen == false (renderer_kernel.cu:354-357: t_max from the depth attachment): the walking kernel on
    the reference's arrays (mnv_get_samples_from_voxels_ex, with visit marks) and the tuned kernel on the packed accel
    (mnv_get_samples_from_voxels_accel_visit_ex) against the oracle -- counts, rows, cluster ids, both trackers, marks."""
    import guided_cases
    torch = torch_gpu
    tree, cam, opt, _ = guided_cases.get_samples_setup(mnv)
    opt.need_viewdir, opt.appearance_embedding, opt.max_guided_samples = need_viewdir, embedding, 8
    opt.max_depth, opt.max_sample_count = 5, 9
    dim = 4 + (3 if need_viewdir else 0) + (1 if embedding != -1 else 0)
    grid = guided_cases.cluster_grid(mnv.ClusterGrid)
    tmax = guided_cases.onscreen_tmax(cam)
    v = tree.host_view()
    sc = np.random.default_rng(4).integers(0, 14, size=(v.capacity, 8)).astype(np.int16)
    visited_ref = np.zeros(v.capacity, np.int32)
    want = orc.get_samples(orc.tree_from_view(v, sample_counts=sc), cam.c, opt, grid, dim, visited=visited_ref, track_visit=True, tmax_px=tmax)
    off = orc.get_samples(orc.tree_from_view(v, sample_counts=sc), cam.c, opt, grid, dim)
    assert int((want["num_samples"] != off["num_samples"]).sum()) > 1000  # the limits do something
    tree.move_to_device(need_parent=True, need_sample_counts=True)
    dv = tree.device_view()
    sc_dev = torch.from_numpy(sc).cuda()
    dv.sample_counts = sc_dev.data_ptr()
    d_tmax = torch.from_numpy(tmax).cuda()
    n = cam.width * cam.height
    k = np.arange(8)[None, :] < want["num_samples"][:, None]

    def buffers():
        return (torch.zeros(n, dtype=torch.int16, device="cuda"), torch.full((n, 8, dim), -1.0, dtype=torch.float32, device="cuda"),
                torch.full((n, 8), -1, dtype=torch.int16, device="cuda"), torch.full((n, 3), -1.0, dtype=torch.float32, device="cuda"),
                torch.full((n, 3), -1.0, dtype=torch.float32, device="cuda"), torch.zeros(v.capacity, dtype=torch.int32, device="cuda"))

    def check(num, samples, clusters, split, sample, visited, who):
        torch.cuda.synchronize()
        assert np.array_equal(num.cpu().numpy(), want["num_samples"]), who
        got_s, got_c = samples.cpu().numpy(), clusters.cpu().numpy()
        assert np.array_equal(cases.bits(got_s[k]), cases.bits(want["samples"][k])) and np.all(got_s[~k][:, 0] == -1.0), who
        assert np.array_equal(got_c[k], want["cluster_indices"][k]), who
        assert np.array_equal(cases.bits(split.cpu().numpy()), cases.bits(want["split"])) and np.array_equal(cases.bits(sample.cpu().numpy()), cases.bits(want["sample"])), who
        assert np.array_equal(visited.cpu().numpy(), visited_ref), who

    num, samples, clusters, split, sample, visited = buffers()
    mnv.get_samples_from_voxels(dv, cam, opt, num, samples, clusters, grid, split_track=split, sample_track=sample, visited=visited, track_visit=True,
                                tmax_px=d_tmax)
    check(num, samples, clusters, split, sample, visited, "mnv_get_samples_from_voxels_ex")
    num, samples, clusters, split, sample, visited = buffers()
    mnv.get_samples_from_voxels_accel_visit(tree.accel, cam, opt, visited, dv.parent, num, samples, clusters, grid, split_track=split, sample_track=sample,
                                            sample_counts=sc_dev, tmax_px=d_tmax)
    check(num, samples, clusters, split, sample, visited, "mnv_get_samples_from_voxels_accel_visit_ex")


def test_guided_samples_with_the_depth_image_against_the_live_reference_and_through_the_binding(mnv, orc, torch_gpu, tmp_path):
    """The reference's own get_samples_trace_ray with t_max from a depth image, run here (oracle/_ref), reproduces the committed golden; and
    viewer::get_samples_from_voxels with its sixteen original parameters (include/mnv_reference_binding.hpp, offscreen == false), served by
    libmnv.so on the reference's N3Tree / Camera / tensors, returns the same arrays."""
    import guided_cases
    mnv_ref = require_live_reference()
    g = np.load(os.path.join(GOLD, "ref_guided_get_samples_onscreen.npz"))
    tree, cam, opt, dim = guided_cases.get_samples_setup(mnv)
    opt.max_depth, opt.max_sample_count = 5, 9
    grid = guided_cases.cluster_grid(mnv.ClusterGrid)
    tmax = guided_cases.onscreen_tmax(cam)
    path = str(tmp_path / "t.npz")
    tree.save_npz(path)
    k = np.arange(opt.max_guided_samples)[None, :] < g["num_samples"][:, None]
    for dropin in (False, True):
        got = mnv_ref.get_samples_npz(path, cam.c, opt, grid, dim, tmax_px=tmax, dropin=dropin)
        assert np.array_equal(got["num_samples"], g["num_samples"]), dropin
        assert np.array_equal(got["cluster_indices"][k], g["cluster_indices"][k]) and np.array_equal(cases.bits(got["samples"][k]), cases.bits(g["samples"][k])), dropin
        assert np.array_equal(cases.bits(got["split"]), cases.bits(g["split"])) and np.array_equal(cases.bits(got["sample"]), cases.bits(g["sample"])), dropin
    # offscreen == true through the binding: the offscreen golden
    g0 = np.load(os.path.join(GOLD, "ref_guided_get_samples.npz"))
    tree, cam, opt, dim = guided_cases.get_samples_setup(mnv)
    got = mnv_ref.get_samples_npz(path, cam.c, opt, grid, dim, dropin=True)
    k0 = np.arange(opt.max_guided_samples)[None, :] < g0["num_samples"][:, None]
    assert np.array_equal(got["num_samples"], g0["num_samples"]) and np.array_equal(cases.bits(got["samples"][k0]), cases.bits(g0["samples"][k0]))


def test_host_renderer_renders_the_live_frame(mnv, orc, torch_gpu):
    """VolumeRenderer with set_frame_inputs: the reference's render loop passes offscreen == false to every launcher (cuda_renderer.cpp:111-113,
    135-136, 141-142).  A plain frame through the host renderer, with the depth image and the image under the volume, equals the oracle's
    frame of the same inputs bit for bit (float and RGBA8); a guided-sampling frame equals the fused entry point called with the same depth
    image, on the fused and on the four-step path of the renderer; without inputs it is the offline renderer again.
    (One render() per renderer: Camera::_update normalises v_back on every call, as the reference's does (camera.cpp:54-82), and a
    normalised vector normalised again may move by an ulp -- this case's camera alternates between two matrices from frame to frame.)"""
    import ctypes as C

    import mlp_cases
    from test_renderer_refine_gpu import make_grid
    torch = torch_gpu
    name = "onscreen_both"
    base = cases.ONSCREEN[name][0]
    spec = cases.CASES[base]
    tree = cases.make_tree(mnv, spec["tree"])
    v = tree.host_view()
    cs = spec["camera"]
    cam = cases.make_camera(mnv, cs)
    opt = cases.make_options(mnv, spec["options"])
    opt.basis_minmax[0], opt.basis_minmax[1] = 0, max(v.basis_dim - 1, 0)
    tmax, image = cases.onscreen_inputs(name, cam)
    d_tmax, d_image = torch.from_numpy(tmax).cuda(), torch.from_numpy(image).cuda()
    desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, dir_octaves=2, need_viewdir=False, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    params = mlp_cases.make_params(mnv, desc, seed=21)
    grid = make_grid(mnv)

    def renderer(model=False, **over):
        r = mnv.Renderer()
        r.resize(cs["width"], cs["height"])
        r.set(tree, v.capacity)
        if model:
            r.set_model(desc, params, grid)
        r.set_camera(cs["center"], cs["back"], fx=cs["fx"], up=cs.get("up", (0.0, 0.0, 1.0)))
        C.memmove(C.byref(r.options), C.byref(opt), C.sizeof(opt))  # every option of the case
        for k, val in over.items():
            setattr(r.options, k, val)
        return r

    t = orc.tree_from_view(v)
    want = orc.render(t, cam.c, opt, want_rgba8=True, tmax_px=tmax, rgba8_init=image)
    off = orc.render(t, cam.c, opt, want_rgba8=True)
    assert int((cases.bits(want["rgba"]) != cases.bits(off["rgba"])).any(axis=-1).sum()) > 5000
    r = renderer()
    r.set_frame_inputs(d_tmax, d_image)
    r.render()
    f32, u8 = r.download(want_rgba8=True)
    assert np.array_equal(cases.bits(f32), cases.bits(want["rgba"])) and np.array_equal(u8, want["rgba8"])
    r = renderer()
    r.set_frame_inputs(d_tmax, d_image)
    r.set_frame_inputs(None, None)
    r.render()
    f32, u8 = r.download(want_rgba8=True)
    assert np.array_equal(cases.bits(f32), cases.bits(off["rgba"])) and np.array_equal(u8, off["rgba8"])
    # guided sampling: the renderer's frame against the fused entry point with the same depth image
    opt2 = mnv.RenderOptions()
    C.memmove(C.byref(opt2), C.byref(opt), C.sizeof(opt2))
    opt2.use_guided_sampling, opt2.max_guided_samples = True, 16
    tree.move_to_device()
    out = torch.empty((cam.height, cam.width, 4), dtype=torch.float32, device="cuda")
    mnv.render_guided_fused(tree.accel, cam, opt2, mnv.Mlp(desc, params), grid, rgba=out, tmax_px=d_tmax)
    torch.cuda.synchronize()
    direct = out.cpu().numpy()
    mnv.render_guided_fused(tree.accel, cam, opt2, mnv.Mlp(desc, params), grid, rgba=out)
    torch.cuda.synchronize()
    assert int((cases.bits(out.cpu().numpy()) != cases.bits(direct)).any(axis=-1).sum()) > 1000  # the depth image matters
    for fused in (True, False):
        r = renderer(model=True, use_guided_sampling=True, max_guided_samples=16)
        r.set_frame_inputs(d_tmax, d_image)
        r.set_fused_guided(fused)
        st = r.render()
        assert bool(st["fused"]) == fused
        assert np.array_equal(cases.bits(r.download()), cases.bits(direct)), f"renderer's guided frame (fused {fused})"


@pytest.mark.parametrize("name", ["onscreen_both", "onscreen_terrain"])
def test_tracker_frame_of_the_live_call_on_both_layouts(mnv, orc, torch_gpu, name):
    """The frame the reference's render loop actually launches (cuda_renderer.cpp:141-142): render_voxels with the refinement trackers, visit marks
    AND offscreen == false.  mnv_render_voxels_ex on the reference's arrays and mnv_render_voxels_accel_visit_ex on the packed accel against the
    oracle: float and RGBA8 frame, both tracker arrays, the marks."""
    torch = torch_gpu
    base = cases.ONSCREEN[name][0]
    spec = cases.CASES[base]
    tree = cases.make_tree(mnv, spec["tree"])
    v = tree.host_view()
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.basis_minmax[0], opt.basis_minmax[1] = 0, max(v.basis_dim - 1, 0)
    opt.max_depth, opt.max_sample_count = 5, 9
    tmax, image = cases.onscreen_inputs(name, cam)
    sc = np.random.default_rng(7).integers(0, 14, size=(v.capacity, 8)).astype(np.int16)
    visited_ref = np.zeros(v.capacity, np.int32)
    want = orc.render(orc.tree_from_view(v, sample_counts=sc), cam.c, opt, want_rgba8=True, want_trackers=True, visited=visited_ref, track_visit=True,
                      tmax_px=tmax, rgba8_init=image)
    off = orc.render(orc.tree_from_view(v, sample_counts=sc), cam.c, opt, want_trackers=True)
    assert int((cases.bits(want["split"]) != cases.bits(off["split"])).any(axis=-1).sum()) > 100  # the depth image changes what the trackers see
    tree.move_to_device(need_parent=True, need_sample_counts=True)
    dv = tree.device_view()
    sc_dev = torch.from_numpy(sc).cuda()
    dv.sample_counts = sc_dev.data_ptr()
    d_tmax, d_image = torch.from_numpy(tmax).cuda(), torch.from_numpy(image).cuda()
    n = cam.width * cam.height
    for who in ("ref_layout", "accel"):
        rgba = torch.full((cam.height, cam.width, 4), float("nan"), dtype=torch.float32, device="cuda")
        rgba8 = torch.zeros((cam.height, cam.width, 4), dtype=torch.uint8, device="cuda")
        split = torch.full((n, 3), -1.0, dtype=torch.float32, device="cuda")
        sample = torch.full((n, 3), -1.0, dtype=torch.float32, device="cuda")
        visited = torch.zeros(v.capacity, dtype=torch.int32, device="cuda")
        if who == "accel":
            mnv.render_voxels_accel_visit(tree.accel, cam, opt, visited, dv.parent, rgba=rgba, rgba8=rgba8, split_track=split, sample_track=sample,
                                          sample_counts=sc_dev, tmax_px=d_tmax, rgba8_init=d_image)
        else:
            mnv.render_voxels(dv, cam, opt, rgba=rgba, rgba8=rgba8, split_track=split, sample_track=sample, visited=visited, track_visit=True,
                              tmax_px=d_tmax, rgba8_init=d_image)
        torch.cuda.synchronize()
        assert np.array_equal(cases.bits(rgba.cpu().numpy()), cases.bits(want["rgba"])) and np.array_equal(rgba8.cpu().numpy(), want["rgba8"]), who
        assert np.array_equal(cases.bits(split.cpu().numpy().reshape(want["split"].shape)), cases.bits(want["split"])), who
        assert np.array_equal(cases.bits(sample.cpu().numpy().reshape(want["sample"].shape)), cases.bits(want["sample"])), who
        assert np.array_equal(visited.cpu().numpy(), visited_ref), who


def test_live_reference_reproduces_the_tracker_frame_golden(mnv, tmp_path):
    """oracle/_ref run here: the reference's render_voxels_trace_ray with trackers, marks, depth image and image gives the committed arrays."""
    mnv_ref = require_live_reference()
    g = np.load(os.path.join(GOLD, "ref_onscreen_trackers_both.npz"))
    name = "onscreen_both"
    spec = cases.CASES[cases.ONSCREEN[name][0]]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.max_depth, opt.max_sample_count = 5, 9
    tmax, image = cases.onscreen_inputs(name, cam)
    v = tree.host_view()
    counts = np.random.default_rng(7).integers(0, 14, size=(v.capacity, 8)).astype(np.int16)
    path = str(tmp_path / "t.npz")
    tree.save_npz(path)
    live = mnv_ref.render_track_npz(path, cam.c, opt, v.capacity, sample_counts=counts, track_visit=True, tmax_px=tmax, rgba8_init=image)
    for k in ("rgba", "split", "sample"):
        assert np.array_equal(cases.bits(live[k]), cases.bits(g[k])), k
    assert np.array_equal(live["visited"], g["visited"])

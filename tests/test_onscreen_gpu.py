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
        mnv.set_ref_table_min_rays({"ref_layout": 1 << 16, "ref_layout_table": 0, "ref_layout_walk": -1, "tree_cache": 1 << 16}[which])
        if which == "tree_cache":
            mnv.set_tree_cache(True)
        try:
            mnv.render_voxels(tree.device_view(), cam, opt, rgba=rgba, rgba8=rgba8, tmax_px=t_dev, rgba8_init=i_dev)
            torch.cuda.synchronize()
        finally:
            mnv.set_ref_table_min_rays(1 << 16)
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

"""The VolumeRenderer refinement loop (host/volume_renderer.cpp; reference src/renderer/cuda_renderer.cpp:98-156,205-381)
driven through the mnv_renderer_* C ABI.

Frame-level checks: the first refinement step is emulated with the checkers (CPU march oracle for the trackers,
refine_oracle for the vote, the C oracle for add_children / sample generation / the MLP, refine_oracle for the mean)
and must give the same topology exactly and the same new rows to the MLP tolerance; guided sampling is compared
bit for bit with the reference's own tensor expressions (cumsum + boolean-mask compaction) evaluated by torch on the
device; longer runs are checked through tree invariants and accel-vs-reference-layout equality on the grown tree."""
import numpy as np
import pytest

import cases
import mlp_cases
import refine_oracle as ro

pytestmark = pytest.mark.gpu

M64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9e3779b97f4a7c15) & M64
    x = ((x ^ (x >> 30)) * 0xbf58476d1ce4e5b9) & M64
    x = ((x ^ (x >> 27)) * 0x94d049bb133111eb) & M64
    return x ^ (x >> 31)


def uniform_numpy(n, seed):
    """mnv_fill_uniform on the host (include/mnv.h)."""
    s = splitmix64(seed & M64)
    out = np.empty(n, np.float32)
    for i in range(n):
        out[i] = np.float32((splitmix64(s ^ i) >> 40) * 2.0 ** -24)
    return out


def make_grid(mnv):
    g = mnv.ClusterGrid()
    g.grid_dim[0], g.grid_dim[1] = 3, 2
    for i, (lo, rng) in enumerate([(-1.0, 2.0), (-1.1, 2.2), (-0.9, 1.8)]):
        g.min_position[i], g.range[i] = lo, rng
    return g


def setup(mnv, case, extra_capacity, need_viewdir=False, seed=5, **opt_over):
    spec = cases.CASES[case]
    tree = cases.make_tree(mnv, spec["tree"])
    v = tree.host_view()
    cam_spec = spec["camera"]
    desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, dir_octaves=2, need_viewdir=need_viewdir, hidden_width=64, hidden_layers=2,
                        out_dim=v.data_dim + 1)
    params = mlp_cases.make_params(mnv, desc, seed=21)
    r = mnv.Renderer()
    r.resize(cam_spec["width"], cam_spec["height"])
    r.set(tree, v.capacity + extra_capacity)
    r.set_model(desc, params, make_grid(mnv))
    # Camera ctor defaults (camera.cpp:29-45) when the case does not pose the camera
    r.set_camera(cam_spec.get("center", (-3.55, 0.0, 3.55)), cam_spec.get("back", (-0.7071068, 0.0, 0.7071068)), fx=cam_spec["fx"])
    r.set_seed(seed)
    opt = cases.make_options(mnv, spec["options"])
    o = r.options
    for name in ("step_size", "sigma_thresh", "stop_thresh", "background_brightness"):
        setattr(o, name, getattr(opt, name))
    for k, val in opt_over.items():
        setattr(o, k, val)
    return r, tree, desc, params, cam_spec


def check_tree_links(child, parent, cap):
    """Every chunk but the root hangs under exactly the slot its parent word names."""
    tgt = np.arange(cap)[:, None] + child[:cap]
    nz = child[:cap] != 0
    assert np.all((tgt[nz] > 0) & (tgt[nz] < cap))
    rows, slots = np.nonzero(nz)
    assert np.array_equal(parent[tgt[nz]], rows * 8 + slots)
    assert np.array_equal(np.sort(tgt[nz]), np.arange(1, cap))  # each chunk referenced once


def test_first_refinement_step_matches_checkers(mnv, orc, torch_gpu):
    seed = 5
    r, tree, desc, params, cam_spec = setup(mnv, "rgba_d5", 5000, seed=seed, use_splitting=True, max_depth=7, split_batch_size=64,
                                            samples_per_corner=4, max_sample_count=64)
    tree0 = cases.make_tree(mnv, cases.CASES["rgba_d5"]["tree"])  # untouched twin for the checkers (sync_tree reallocates the host arrays)
    v = tree0.host_view()
    cap, dd = v.capacity, v.data_dim
    data0, child0, parent0 = (a.copy() for a in tree0.host_arrays())
    st = r.render()
    frame = r.download()
    r.sync_tree()

    # --- the same step with the checkers
    cam = cases.make_camera(mnv, cam_spec)
    opt = mnv.RenderOptions()
    import ctypes as C
    C.memmove(C.byref(opt), C.byref(r.options), C.sizeof(opt))
    counts = np.full((cap, 8), 8, np.int16)
    ref = orc.render(orc.tree_from_view(v, sample_counts=counts), cam.c, opt, want_trackers=True)
    assert np.array_equal(cases.bits(frame), cases.bits(ref["rgba"]))
    nodes, n_cand = ro.select_split_candidates(ref["split"].reshape(-1, 3), 64)
    n = nodes.shape[0]
    assert n > 0 and st["split_candidates"] == n_cand and st["added"] == n and st["capacity"] == cap + n
    assert st["used_accel"] == 1 and st["track_visit"] == 0 and st["pruned"] == 0
    max_cap = cap + 5000
    child = np.zeros((max_cap, 8), np.int32)
    child[:cap] = child0
    parent = np.zeros(max_cap, np.int32)
    parent[:cap] = parent0
    visited = np.zeros(max_cap, np.int32)
    visited[0] = 1
    spc, dim = 4, 3
    samples = uniform_numpy(n * 8 * spc * dim, seed).reshape(n * 8, spc, dim)
    clusters = np.full((n * 8, spc), -1, np.int16)
    orc.add_children_and_generate_samples(child, parent, list(v.offset), list(v.scale), cap, opt, nodes, samples, clusters, visited, make_grid(mnv))
    results = orc.mlp_forward(desc, params, clusters.reshape(-1), samples.reshape(-1, dim), out_cols=dd + 1).reshape(n * 8, spc, dd + 1)
    data = np.zeros((max_cap, 8, dd), np.float16)
    data[:cap] = data0.view(np.float16)
    big_counts = np.zeros((max_cap, 8), np.int16)
    big_counts[:cap] = counts
    ro.apply_split_results(data, big_counts, cap, results, spc)

    got_data, got_child, got_parent = tree.host_arrays()
    assert tree.capacity == cap + n
    assert np.array_equal(got_child, child[:cap + n]) and np.array_equal(got_parent, parent[:cap + n])
    check_tree_links(got_child, got_parent, cap + n)
    assert np.array_equal(got_data[:cap], data0)  # existing rows untouched
    new_got = got_data[cap:].view(np.float16).astype(np.float32)
    new_want = data[cap:cap + n].astype(np.float32)
    err = np.abs(new_got - new_want) / (1.0 + np.abs(new_want))
    assert err.max() < 5e-3 and np.abs(new_want).mean() > 0.02


def _accel_equals_reference_layout(mnv, torch, tree, cam_spec, case):
    """The packed accel the renderer keeps and the reference-layout kernel agree bit for bit on the tree as it is now."""
    cam = cases.make_camera(mnv, cam_spec)
    opt = cases.make_options(mnv, cases.CASES[case]["options"])
    opt.basis_minmax[1] = max(tree.host_view().basis_dim - 1, 0)
    h, w = cam.height, cam.width
    a = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    mnv.render_voxels(tree.device_view(), cam, opt, rgba=a)
    mnv.render_voxels_accel(tree.accel, cam, opt, rgba=b)
    torch.cuda.synchronize()
    return torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_refinement_run_keeps_a_valid_tree_and_the_accel_follows(mnv, torch_gpu):
    torch = torch_gpu
    r, tree, desc, params, cam_spec = setup(mnv, "sh4_d6", 40000, need_viewdir=True, use_splitting=True, max_depth=8, split_batch_size=512,
                                            samples_per_corner=4, max_sample_count=24)
    r.set_seed(9)
    cap0 = tree.capacity
    caps, added = [], 0
    for f in range(10):
        a = 0.1 * f
        r.set_camera((-2.4 * np.cos(a) - 1.1 * np.sin(a), 1.1 * np.cos(a) - 2.4 * np.sin(a), 1.6), (-0.72 * np.cos(a) - 0.33 * np.sin(a), 0.33 * np.cos(a) - 0.72 * np.sin(a), 0.48))
        st = r.render()
        caps.append(st["capacity"])
        added += st["added"]
        # every split is patched into the packed accel (mnv_accel_refresh): the tuned kernel stays in use
        assert st["added"] <= 512 and st["pruned"] == 0 and st["used_accel"] == 1 and st["track_visit"] == 0
    assert caps == sorted(caps) and caps[-1] == cap0 + added and added > 1000
    assert np.isfinite(r.download()).all()
    r.sync_tree()
    data, child, parent = tree.host_arrays()
    assert child.shape[0] == caps[-1]
    check_tree_links(child, parent, caps[-1])
    assert _accel_equals_reference_layout(mnv, torch, tree, cam_spec, "sh4_d6")
    # no split candidates left -> get_more_samples rewrites rows of existing leaves; the accel follows those too
    r.options.max_depth = 1
    resampled = 0
    for f in range(3):
        st = r.render()
        resampled += st["resampled"]
        assert st["added"] == 0 and st["used_accel"] == 1
    assert resampled > 0
    assert _accel_equals_reference_layout(mnv, torch, tree, cam_spec, "sh4_d6")


def test_accel_refresh_entry_point(mnv, orc, torch_gpu):
    """mnv_accel_create_reserved + mnv_accel_refresh on their own: split shallow and deep leaves (grid rebuild and plain patch),
    rewrite rows, and compare with an accel built from scratch and with the reference-layout kernel."""
    torch = torch_gpu
    import refine_kernel_cases as rk
    spec = cases.CASES["sh4_d6"]
    tree = cases.make_tree(mnv, spec["tree"])
    v = tree.host_view()
    cap, dd = v.capacity, v.data_dim
    data, child, parent = tree.host_arrays()
    depth = np.zeros(cap, np.int32)
    depth[0] = 1
    for c in range(1, cap):
        depth[c] = depth[parent[c] >> 3] + 1
    leaves = np.argwhere(child == 0)
    rng = np.random.default_rng(3)
    for pick, label in ((leaves[depth[leaves[:, 0]] >= 5], "deep"), (leaves[depth[leaves[:, 0]] <= 2], "shallow")):
        n_new = min(24, len(pick))
        nodes = np.ascontiguousarray(pick[rng.choice(len(pick), n_new, replace=False)], dtype=np.int32)
        max_cap = cap + n_new
        big = lambda a: np.concatenate([a, np.zeros((n_new,) + a.shape[1:], a.dtype)])  # noqa: E731
        d = dict(data=torch.from_numpy(big(data).view(np.int16)).cuda(), child=torch.from_numpy(big(child)).cuda(), parent=torch.from_numpy(big(parent)).cuda())
        tv = mnv.TreeView()
        for f in ("offset", "scale", "N", "data_dim", "format", "basis_dim"):
            setattr(tv, f, getattr(v, f))
        tv.data, tv.child, tv.parent, tv.capacity = d["data"].data_ptr(), d["child"].data_ptr(), d["parent"].data_ptr(), cap
        accel = mnv.accel_create(tv, max_cap)
        # grow: link the children, give them rows
        opt = mnv.RenderOptions.cli_defaults()
        opt.samples_per_corner = 2
        samples = torch.rand((n_new * 8, 2, 3), device="cuda")
        clusters = torch.zeros((n_new * 8, 2), dtype=torch.int16, device="cuda")
        visited = torch.zeros(max_cap, dtype=torch.int32, device="cuda")
        edit = mnv.tree_edit(d["child"], d["parent"], list(v.offset), list(v.scale), cap)
        mnv.add_children_and_generate_samples(edit, opt, torch.from_numpy(nodes).cuda(), samples, clusters, visited, rk.grid(mnv))
        new_rows = torch.from_numpy((rng.standard_normal((n_new * 8, dd)) * 0.7).astype(np.float16).view(np.int16)).cuda()
        new_rows.view(torch.float16)[:, dd - 1] = torch.from_numpy(rng.uniform(0, 40, n_new * 8).astype(np.float16)).cuda()
        d["data"][cap:] = new_rows.view(n_new, 8, dd)
        tv.capacity = max_cap
        mnv.accel_refresh(accel, tv, cap)
        # rewrite rows of some existing leaves as well
        changed = np.ascontiguousarray(leaves[rng.choice(len(leaves), 30, replace=False)], dtype=np.int32)
        rows = torch.from_numpy((rng.standard_normal((30, dd)) * 0.7).astype(np.float16).view(np.int16)).cuda()
        rows.view(torch.float16)[:, dd - 1] = 25.0
        d["data"].view(-1, dd)[torch.from_numpy(changed[:, 0].astype(np.int64) * 8 + changed[:, 1]).cuda()] = rows
        mnv.accel_refresh(accel, tv, max_cap, torch.from_numpy(changed).cuda())
        fresh = mnv.accel_create(tv)
        cam = cases.make_camera(mnv, spec["camera"])
        ropt = cases.make_options(mnv, spec["options"])
        ropt.basis_minmax[1] = 3
        imgs = [torch.empty((cam.height, cam.width, 4), dtype=torch.float32, device="cuda") for _ in range(3)]
        mnv.render_voxels_accel(accel, cam, ropt, rgba=imgs[0])
        mnv.render_voxels_accel(fresh, cam, ropt, rgba=imgs[1])
        mnv.render_voxels(tv, cam, ropt, rgba=imgs[2])
        torch.cuda.synchronize()
        assert torch.equal(imgs[0].view(torch.int32), imgs[2].view(torch.int32)), label
        assert torch.equal(imgs[1].view(torch.int32), imgs[2].view(torch.int32)), label
        # refresh refuses what it cannot follow
        with pytest.raises(mnv.MnvError):
            mnv.accel_refresh(accel, tv, cap)  # stale old_capacity
        # an in-place rebuild (what follows a prune) gives the same accel again
        mnv.accel_rebuild(accel, tv)
        mnv.render_voxels_accel(accel, cam, ropt, rgba=imgs[0])
        torch.cuda.synchronize()
        assert torch.equal(imgs[0].view(torch.int32), imgs[2].view(torch.int32)), label
        mnv.accel_destroy(accel)
        mnv.accel_destroy(fresh)


def test_prune_runs_inside_the_loop(mnv, torch_gpu):
    # max_tree_capacity - capacity < split_batch_size from the start, but capacity <= 3/4 max: the camera change of the first
    # frame does not collect visit marks, so the prune waits one track_visit frame (host/volume_renderer.cpp)
    r, tree, desc, params, cam_spec = setup(mnv, "sh4_d6", 600, use_splitting=True, max_depth=8, split_batch_size=700, samples_per_corner=2,
                                            max_sample_count=24)
    cap0 = tree.capacity
    log = [r.render() for _ in range(4)]
    assert log[0]["track_visit"] == 0 and log[0]["pruned"] == 0 and 0 < log[0]["added"] <= 600
    assert log[1]["track_visit"] == 1 and log[1]["pruned"] > 0 and log[1]["capacity"] < cap0
    assert log[2]["track_visit"] == 1  # prune_happened (cuda_renderer.cpp:101-102)
    # visit-mark frames run on the packed accel too (leaf chunks marked by the march, ancestors by the closure pass), and the prune
    # rebuilds the accel in place: no frame of the loop falls back to the reference-layout kernel
    assert [st["used_accel"] for st in log] == [1, 1, 1, 1]
    r.sync_tree()
    data, child, parent = tree.host_arrays()
    assert child.shape[0] == tree.capacity == log[-1]["capacity"]
    check_tree_links(child, parent, tree.capacity)
    assert np.isfinite(r.download()).all()
    r.options.max_depth, r.options.max_sample_count = 1, -30000
    used = [r.render()["used_accel"] for _ in range(8)]
    assert used == [1] * 8
    assert _accel_equals_reference_layout(mnv, torch_gpu, tree, cam_spec, "sh4_d6")


@pytest.mark.parametrize("guided", [False, True])
def test_rank_mode_of_the_renderer_equals_the_plain_one(mnv, torch_gpu, guided):
    """Renderer.set_ranks with a one-rank communicator (RCCL itself: this is what a one-GPU box can run of it): partitioned tracker march,
    tile gather to the root, un-permute, all-gather of the tracker rows -- the frames, the per-frame decisions and the refined tree equal
    the plain renderer's over a run with splits and a prune.  Several ranks: tests/test_cli_gpu.py (transport stand-in)."""
    kw = dict(use_splitting=True, max_depth=8, split_batch_size=700, samples_per_corner=2, max_sample_count=24)
    if guided:
        kw.update(use_guided_sampling=True, max_guided_samples=24)
    runs = []
    for ranks in (False, True):
        r, tree, desc, params, cam_spec = setup(mnv, "sh4_d6", 600, **kw)
        if ranks:
            r.set_ranks(mnv.Comm(mnv.comm_get_unique_id(), 1, 0), 64, 24)
        log, frames = [], []
        for _ in range(5):
            st = r.render()
            log.append({k: st[k] for k in ("track_visit", "split_candidates", "added", "sample_candidates", "resampled", "pruned", "capacity")})
            frames.append(r.download().copy())
        r.sync_tree()
        runs.append((log, frames, [a.copy() for a in tree.host_arrays()]))
        if ranks:
            r.set_ranks(None)
    assert runs[0][0] == runs[1][0] and any(st["pruned"] > 0 for st in runs[0][0]) and any(st["added"] > 0 for st in runs[0][0])
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for a, b in zip(runs[0][2], runs[1][2]):
        assert np.array_equal(a, b)


def test_prune_on_the_first_frame_when_the_tree_is_nearly_full(mnv, orc, torch_gpu):
    """capacity > 3/4 max and a changed camera: the first frame marks visits (cuda_renderer.cpp:101-102) and prunes right
    after its split step; the picture of that frame is the unpruned tree's."""
    r, tree, desc, params, cam_spec = setup(mnv, "sh4_d6", 36, use_splitting=True, max_depth=8, split_batch_size=24, samples_per_corner=2,
                                            max_sample_count=24)
    tree0 = cases.make_tree(mnv, cases.CASES["sh4_d6"]["tree"])
    cap0 = tree.capacity
    st = r.render()
    assert st["track_visit"] == 1 and st["used_accel"] == 1 and st["added"] == 24 and st["pruned"] > 0
    assert st["capacity"] == cap0 + 24 - st["pruned"]
    frame = r.download()
    cam = cases.make_camera(mnv, cam_spec)
    import ctypes as C
    opt = mnv.RenderOptions()
    C.memmove(C.byref(opt), C.byref(r.options), C.sizeof(opt))
    ref = orc.render(orc.tree_from_view(tree0.host_view()), cam.c, opt)
    assert np.array_equal(cases.bits(frame), cases.bits(ref["rgba"]))
    r.sync_tree()
    data, child, parent = tree.host_arrays()
    check_tree_links(child, parent, tree.capacity)


@pytest.mark.parametrize("extra,accel_path", [(0, 1), (4000, 1)])
def test_guided_sampling_frame_matches_reference_tensor_ops(mnv, torch_gpu, extra, accel_path):
    """use_guided_sampling: get_samples -> compaction -> networks -> composite.  The compaction is compared with the
    reference's own expressions (cuda_renderer.cpp:116-121: cumsum, boolean mask on column 0) run by torch."""
    torch = torch_gpu
    # extra = 0: capacity > 3/4 max, the first frame also tracks visits (four-step path, sample march on the accel with visit marks);
    # with room to grow it is the fused kernel -- the same picture either way
    r, tree, desc, params, cam_spec = setup(mnv, "rgba_d5", extra, use_guided_sampling=True, max_guided_samples=16)
    st = r.render()
    frame = r.download()
    assert st["guided_samples"] > 0 and st["used_accel"] == accel_path

    cam = cases.make_camera(mnv, cam_spec)
    import ctypes as C
    opt = mnv.RenderOptions()
    C.memmove(C.byref(opt), C.byref(r.options), C.sizeof(opt))
    dv = tree.device_view()
    n_px, max_g, dim = cam.width * cam.height, 16, 4
    num = torch.zeros(n_px, dtype=torch.int16, device="cuda")
    guided = torch.zeros((n_px, max_g, dim), dtype=torch.float32, device="cuda")
    guided[:, :, 0] = -1
    clusters = torch.zeros((n_px, max_g), dtype=torch.int16, device="cuda")
    mnv.get_samples_from_voxels(dv, cam, opt, num, guided, clusters, make_grid(mnv))
    offsets = torch.cumsum(num, 0)
    flat = guided.view(-1, dim)
    mask = flat[:, 0] >= 0
    valid, valid_clusters = flat[mask], clusters.view(-1)[mask]
    assert valid.shape[0] == st["guided_samples"] == int(offsets[-1])
    # the build's compaction equals the boolean-mask form
    off2 = torch.empty(n_px, dtype=torch.int64, device="cuda")
    total = valid.shape[0]
    z2 = torch.empty(total, dtype=torch.float32, device="cuda")
    rows2 = torch.empty((total, dim - 1), dtype=torch.float32, device="cuda")
    cl2 = torch.empty(total, dtype=torch.int16, device="cuda")
    assert mnv.compact_guided_samples(num, guided, clusters, off2, z2, rows2, cl2) == total
    torch.cuda.synchronize()
    assert torch.equal(off2, offsets) and torch.equal(z2, valid[:, 0]) and torch.equal(rows2, valid[:, 1:]) and torch.equal(cl2, valid_clusters)
    # networks + composite
    mlp = mnv.Mlp(desc, params)
    values = torch.zeros((total, tree.host_view().data_dim + 1), dtype=torch.float32, device="cuda")
    mlp.query(valid_clusters, valid[:, 1:], values)
    out = torch.empty((cam.height, cam.width, 4), dtype=torch.float32, device="cuda")
    mnv.render_nerf_results(dv, cam, opt, values, valid[:, 0].contiguous(), offsets, rgba=out)
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(frame))
    # further frames with an unchanged camera (Camera::_update renormalises: a fixed point from the second call on)
    # reuse the network outputs (can_reuse_results) and give the same picture
    r.render()
    f2 = r.download()
    st3 = r.render()
    assert st3["guided_samples"] > 0 and np.array_equal(cases.bits(r.download()), cases.bits(f2))
    assert np.abs(f2 - frame).max() < 1e-4


def test_plain_frames_in_flight_match_the_oracle(mnv, orc, torch_gpu):
    """VolumeRenderer::frames_in_flight: plain frames rotate over slots with their own streams and buffers
    (mnv_renderer_set_frames_in_flight / _last_slot / _download_slot).  Seven poses through three slots, each downloaded
    only after the following frames were issued; every frame equals the oracle's bit for bit, and a renderer with one
    frame in flight gives the same frames."""
    spec = cases.CASES["sh9_d7_aniso"]
    tree = cases.make_tree(mnv, spec["tree"])
    ot = orc.tree_from_view(tree.host_view())
    w, h, fx = 320, 200, 420.0
    poses = [((-3.0 + 0.4 * i, 2.0 - 0.3 * i, 4.0 + 0.2 * i), (-0.5 + 0.05 * i, 0.3, 0.75)) for i in range(7)]

    def run(in_flight):
        r = mnv.Renderer()
        r.resize(w, h)
        r.set(tree, tree.capacity)
        r.set_frames_in_flight(in_flight)
        opt = r.options
        opt.background_brightness = 0.25
        got, pending = {}, []
        for i, (center, back) in enumerate(poses):
            if len(pending) >= in_flight:
                j, slot = pending.pop(0)
                got[j] = r.download_slot(slot, want_rgba8=True)
            r.set_camera(center, back, fx=fx)
            st = r.render()
            assert st["used_accel"]
            pending.append((i, r.last_slot()))
        slots_used = {s for _, s in pending}
        for j, slot in pending:
            got[j] = r.download_slot(slot, want_rgba8=True)
        return got, slots_used, mnv.RenderOptions.from_buffer_copy(opt)   # the struct lives inside the renderer

    got3, slots3, opt = run(3)
    got1, slots1, _ = run(1)
    assert slots3 == {0, 1, 2} and slots1 == {0}
    for i, (center, back) in enumerate(poses):
        cam = mnv.Camera(w, h, fx).set_pose(center, back)
        ref = orc.render(ot, cam.c, opt, want_rgba8=True)
        assert np.array_equal(got3[i][0].view(np.uint32), ref["rgba"].view(np.uint32)), i
        assert np.array_equal(got3[i][1], ref["rgba8"]), i
        assert np.array_equal(got1[i][0].view(np.uint32), got3[i][0].view(np.uint32)), i


def test_plain_frames_after_refinement_wait_for_the_tree_edits(mnv, orc, torch_gpu):
    """The frames-in-flight path launches plain frames on slot streams that nothing orders after slot 0's stream, where a
    refinement frame leaves asynchronous work that rewrites the tree and the packed accel (splits, accel patches): the first
    plain frames after the switches go off must see the finished tree.  Several refinement frames (4096 splits each), then three
    plain frames through three slots: each equals the oracle's render of the refined tree bit for bit."""
    r, tree, desc, params, cam_spec = setup(mnv, "sh9_d7_aniso", 60000, use_splitting=True, max_depth=9, split_batch_size=4096, samples_per_corner=4)
    r.set_frames_in_flight(3)
    added = 0
    for _ in range(4):
        added += r.render()["added"]
    assert added > 0
    r.options.use_splitting = False
    w, h, fx = cam_spec["width"], cam_spec["height"], cam_spec["fx"]
    poses = [(cam_spec.get("center", (-3.55, 0.0, 3.55)), cam_spec.get("back", (-0.7071068, 0.0, 0.7071068))),
             ((-3.2, 0.4, 3.6), (-0.68, 0.1, 0.72)), ((-3.4, -0.3, 3.3), (-0.72, -0.05, 0.69))]
    slots = []
    for center, back in poses:
        r.set_camera(center, back, fx=fx)
        st = r.render()          # no synchronisation between the last refinement frame and these
        assert st["used_accel"] and st["added"] == 0
        slots.append(r.last_slot())
    assert len(set(slots)) == 3
    frames = [r.download_slot(s) for s in slots]
    r.sync_tree()
    ot = orc.tree_from_view(tree.host_view())
    opt = mnv.RenderOptions.from_buffer_copy(r.options)
    for (center, back), got in zip(poses, frames):
        ref = orc.render(ot, mnv.Camera(w, h, fx).set_pose(center, back).c, opt)["rgba"]
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))

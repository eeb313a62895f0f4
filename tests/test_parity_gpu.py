"""GPU parity tests proper: the HIP kernels, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Bar: BIT-EXACT float RGBA (north_star tolerance is 1e-4 per channel; the
kernels reproduce the oracle's arithmetic specification exactly, so the tests ask for equality)."""
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def require_live_reference():
    """oracle/_ref/libmnv_ref_gfx950.so, loaded.  Absent on a machine that never had /root/reference: skip.  Present at collection
    (conftest.py then sets MNV_REQUIRE_LIVE_REF=1) but unusable now: fail -- a skip would pass for a comparison that did not happen."""
    import mnv_ref

    must = os.environ.get("MNV_REQUIRE_LIVE_REF") == "1"
    if not mnv_ref.available():
        if must:
            pytest.fail("MNV_REQUIRE_LIVE_REF=1 but oracle/_ref/libmnv_ref_gfx950.so is missing")
        pytest.skip("oracle/_ref/libmnv_ref_gfx950.so not built (needs /root/reference at build time)")
    try:
        mnv_ref.lib()
    except OSError as e:
        if must:
            pytest.fail(f"oracle/_ref/libmnv_ref_gfx950.so is present but does not load: {e}")
        pytest.skip(f"oracle/_ref/libmnv_ref_gfx950.so does not load here: {e}")
    return mnv_ref



def _render_gpu(mnv, torch, tree, cam, opt, which, tile=None, want_u8=False):
    w, h = (cam.width, cam.height) if tile is None else (tile[2], tile[3])
    rgba = torch.full((h, w, 4), float("nan"), dtype=torch.float32, device="cuda")
    rgba8 = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") if want_u8 else None
    if which.startswith("ref_layout"):
        # reference-layout kernel: "_table" forces the per-launch level-7 lookup table at every size, "_walk" forbids it
        # (per-workgroup level-3 table only); plain = the default switch-over at 65536 rays
        # (the forced forms run the same object code through the test-hook build's library: the shipped one has no such switch)
        mnv.render_voxels(tree.device_view(), cam, opt, tile=tile, rgba=rgba, rgba8=rgba8,
                          table_min_rays={"ref_layout": None, "ref_layout_table": 0, "ref_layout_walk": -1}[which])
    else:
        mnv.render_voxels_accel(tree.accel, cam, opt, tile=tile, rgba=rgba, rgba8=rgba8)
    torch.cuda.synchronize()
    return rgba.cpu().numpy(), (rgba8.cpu().numpy() if want_u8 else None)


@pytest.mark.parametrize("name", list(cases.CASES))
@pytest.mark.parametrize("which", ["ref_layout", "ref_layout_table", "ref_layout_walk", "accel"])
def test_case_bit_exact_vs_oracle(mnv, orc, torch_gpu, name, which):
    spec = cases.CASES[name]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    ref = orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt, want_rgba8=True)
    tree.move_to_device()
    got, got8 = _render_gpu(mnv, torch_gpu, tree, cam, opt, which, want_u8=True)
    assert not np.isnan(got).any(), "kernel left pixels unwritten"
    diff = cases.bits(got) != cases.bits(ref["rgba"])
    assert not diff.any(), f"{name}/{which}: {int(diff.any(axis=-1).sum())} pixels differ, max|d|={np.abs(got - ref['rgba']).max():.3e}"
    assert np.array_equal(got8, ref["rgba8"])


@pytest.mark.parametrize("name", ["sh4_d6", "sh9_d7_aniso", "rgba_d5"])
def test_trackers_with_the_per_launch_table(mnv, orc, torch_gpu, name):
    """Tracker rows of the reference-layout kernel when leaves come out of the per-launch lookup table (chunk, child and depth of a
    leaf are decoded from the table word, its sigma is the table's copy): equal to the oracle's, and to the rows without the table."""
    torch = torch_gpu
    spec = cases.CASES[name]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.max_depth = 5
    opt.max_sample_count = 9
    v = tree.host_view()
    sc = np.full((v.capacity, 8), 8, np.int16)
    sc[::3] = 12
    ref = orc.render(orc.tree_from_view(v, sample_counts=sc), cam.c, opt, want_trackers=True)
    tree.move_to_device(need_sample_counts=True)
    dv = tree.device_view()
    sc_dev = torch.from_numpy(sc).cuda()
    dv.sample_counts = sc_dev.data_ptr()
    h, w = cam.height, cam.width
    for min_rays in (0, -1):
        rgba = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
        split = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
        sample = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
        mnv.render_voxels(dv, cam, opt, rgba=rgba, split_track=split, sample_track=sample, table_min_rays=min_rays)
        torch.cuda.synchronize()
        assert np.array_equal(cases.bits(rgba.cpu().numpy()), cases.bits(ref["rgba"]))
        assert np.array_equal(split.cpu().numpy().reshape(-1, 3), ref["split"].reshape(-1, 3))
        assert np.array_equal(sample.cpu().numpy().reshape(-1, 3), ref["sample"].reshape(-1, 3))


def test_trackers_and_visited_match_oracle(mnv, orc, torch_gpu):
    """Refinement trackers (rt_core.cuh:237-252,308-321) and visit marks (:132-134) of the
    reference-layout kernel."""
    torch = torch_gpu
    spec = cases.CASES["sh4_d6"]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.max_depth = 5
    opt.max_sample_count = 9
    v = tree.host_view()
    sc = np.full((v.capacity, 8), 8, np.int16)
    sc[::3] = 12  # some voxels already over max_sample_count
    visited_ref = np.zeros(v.capacity, np.int32)
    ref = orc.render(orc.tree_from_view(v, sample_counts=sc), cam.c, opt, want_trackers=True, visited=visited_ref, track_visit=True)
    tree.move_to_device(need_sample_counts=True)
    dv = tree.device_view()
    sc_dev = torch.from_numpy(sc).cuda()
    dv.sample_counts = sc_dev.data_ptr()
    h, w = cam.height, cam.width
    rgba = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    split = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
    sample = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
    visited = torch.zeros(v.capacity, dtype=torch.int32, device="cuda")
    mnv.render_voxels(dv, cam, opt, rgba=rgba, split_track=split, sample_track=sample, visited=visited, track_visit=True)
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(rgba.cpu().numpy()), cases.bits(ref["rgba"]))
    assert np.array_equal(split.cpu().numpy(), ref["split"])
    assert np.array_equal(sample.cpu().numpy(), ref["sample"])
    assert np.array_equal(visited.cpu().numpy(), visited_ref)


@pytest.mark.parametrize("name", ["sh4_d6", "shell_d7_sh9", "rgba_d5", "terrain_d7_aniso", "cfg1_sh1_d4", "camera_inside", "ray_miss"])
def test_accel_visit_marks_equal_the_reference_layout_kernels(mnv, orc, torch_gpu, name):
    """mnv_render_voxels_accel_visit: the tuned kernel marks the chunk of every leaf it steps through and a closure pass adds the
    ancestors -- the array the reference's per-level marking leaves (rt_core.cuh:132-134), element for element; trackers and frame
    unchanged.  Checked against the oracle and the reference-layout kernel, also without trackers and in the sample march."""
    torch = torch_gpu
    spec = cases.CASES[name]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.max_depth, opt.max_sample_count = 5, 9
    v = tree.host_view()
    sc = np.full((v.capacity, 8), 8, np.int16)
    sc[::3] = 12
    visited_ref = np.zeros(v.capacity, np.int32)
    ref = orc.render(orc.tree_from_view(v, sample_counts=sc), cam.c, opt, want_trackers=True, visited=visited_ref, track_visit=True)
    tree.move_to_device(need_parent=True, need_sample_counts=True)
    dv = tree.device_view()
    sc_dev = torch.from_numpy(sc).cuda()
    parent = dv.parent
    assert parent
    h, w = cam.height, cam.width
    rgba = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    split = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
    sample = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
    visited = torch.zeros(v.capacity, dtype=torch.int32, device="cuda")
    mnv.render_voxels_accel_visit(tree.accel, cam, opt, visited, parent, rgba=rgba, split_track=split, sample_track=sample, sample_counts=sc_dev)
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(rgba.cpu().numpy()), cases.bits(ref["rgba"]))
    assert np.array_equal(split.cpu().numpy(), ref["split"]) and np.array_equal(sample.cpu().numpy(), ref["sample"])
    assert np.array_equal(visited.cpu().numpy(), visited_ref)
    # marks only (no trackers asked for)
    visited2 = torch.zeros(v.capacity, dtype=torch.int32, device="cuda")
    mnv.render_voxels_accel_visit(tree.accel, cam, opt, visited2, parent, rgba=rgba)
    torch.cuda.synchronize()
    assert torch.equal(visited2, visited) and np.array_equal(cases.bits(rgba.cpu().numpy()), cases.bits(ref["rgba"]))
    # the sample-emitting march leaves the same marks (same steps)
    from test_renderer_refine_gpu import make_grid
    opt.max_guided_samples = 8
    n_px = h * w
    num = torch.zeros(n_px, dtype=torch.int16, device="cuda")
    guided = torch.zeros((n_px, 8, 4), dtype=torch.float32, device="cuda")
    clusters = torch.zeros((n_px, 8), dtype=torch.int16, device="cuda")
    visited3 = torch.zeros(v.capacity, dtype=torch.int32, device="cuda")
    mnv.get_samples_from_voxels_accel_visit(tree.accel, cam, opt, visited3, parent, num, guided, clusters, make_grid(mnv))
    torch.cuda.synchronize()
    assert torch.equal(visited3, visited)


@pytest.mark.parametrize("name,max_depth,with_counts", [("sh4_d6", 5, True), ("shell_d7_sh9", 7, True), ("rgba_d5", 3, False),
                                                        ("terrain_d7_aniso", 6, True), ("cfg1_sh1_d4", 9, True)])
def test_accel_trackers_match_oracle(mnv, orc, torch_gpu, name, max_depth, with_counts):
    """The tuned kernel's tracker mode (mnv_render_voxels_accel_track) writes the rows of
    rt_core.cuh:237-252,308-321 -- voxel indices recovered through the lookup grids."""
    torch = torch_gpu
    spec = cases.CASES[name]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.max_depth = max_depth
    opt.max_sample_count = 9
    v = tree.host_view()
    rng = np.random.default_rng(7)
    sc = rng.integers(0, 14, size=(v.capacity, 8)).astype(np.int16) if with_counts else None
    ref = orc.render(orc.tree_from_view(v, sample_counts=sc), cam.c, opt, want_trackers=True)
    tree.move_to_device()
    sc_dev = torch.from_numpy(sc).cuda() if with_counts else None
    h, w = cam.height, cam.width
    rgba = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    split = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
    sample = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
    mnv.render_voxels_accel_track(tree.accel, cam, opt, rgba=rgba, split_track=split, sample_track=sample, sample_counts=sc_dev)
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(rgba.cpu().numpy()), cases.bits(ref["rgba"]))
    assert np.array_equal(split.cpu().numpy(), ref["split"])
    assert np.array_equal(sample.cpu().numpy(), ref["sample"])
    # a tile of the frame, split tracker only
    tile = (w // 4, h // 4, w // 2 + 3, h // 2 + 1)
    split_t = torch.full((tile[3], tile[2], 3), -1.0, dtype=torch.float32, device="cuda")
    mnv.render_voxels_accel_track(tree.accel, cam, opt, tile=tile, split_track=split_t)
    torch.cuda.synchronize()
    assert np.array_equal(split_t.cpu().numpy(), ref["split"][tile[1]:tile[1] + tile[3], tile[0]:tile[0] + tile[2]])


def test_empty_tree_draws_background(mnv, torch_gpu):
    torch = torch_gpu
    cam = mnv.Camera(64, 48)
    opt = mnv.RenderOptions.defaults()
    opt.background_brightness = 0.75
    v = mnv.TreeView()  # N == 0: "draw nothing" (renderer_kernel.cu:266)
    rgba = torch.zeros((48, 64, 4), dtype=torch.float32, device="cuda")
    mnv.render_voxels(v, cam, opt, rgba=rgba)
    torch.cuda.synchronize()
    out = rgba.cpu().numpy()
    assert np.all(out[..., :3] == 0.75) and np.all(out[..., 3] == 0.0)


def test_ragged_tiles_and_empty_tile(mnv, orc, torch_gpu):
    """Tile rectangles that are not multiples of the 8x8 wave tile, a 1x1 tile and an empty tile."""
    spec = cases.CASES["sh9_d7_aniso"]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    full = orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt)["rgba"]
    tree.move_to_device()
    for which in ("ref_layout", "accel"):
        for tile in [(3, 5, 37, 21), (100, 0, 140, 136), (239, 135, 1, 1), (0, 0, 8, 8)]:
            got, _ = _render_gpu(mnv, torch_gpu, tree, cam, opt, which, tile=tile)
            x0, y0, w, h = tile
            assert np.array_equal(cases.bits(got), cases.bits(full[y0:y0 + h, x0:x0 + w])), (which, tile)
        # empty tile: no launch, no error
        if which == "ref_layout":
            mnv.render_voxels(tree.device_view(), cam, opt, tile=(0, 0, 0, 0), rgba=None)
        else:
            mnv.render_voxels_accel(tree.accel, cam, opt, tile=(0, 0, 0, 0), rgba=None)


def test_invalid_arguments_report_errors(mnv, torch_gpu):
    cam = mnv.Camera(16, 16)
    opt = mnv.RenderOptions.defaults()
    v = mnv.TreeView()
    v.N = 17  # branching factors up to 16 take the general walk (test_general_branching_factor_bit_exact); beyond: refused
    with pytest.raises(mnv.MnvError) as e:
        mnv.render_voxels(v, cam, opt, rgba=None)
    assert e.value.code == mnv.MNV_E_UNSUPPORTED
    v.N = 2  # null arrays
    with pytest.raises(mnv.MnvError) as e:
        mnv.render_voxels(v, cam, opt, rgba=None)
    assert e.value.code == mnv.MNV_E_INVALID
    # the tuned kernel numbers the pixels of a launch with 32 bits: a rectangle beyond that is refused, not wrapped
    import ctypes as C
    tree = cases.make_tree(mnv, cases.CASES["rgba_d5"]["tree"])
    tree.move_to_device()
    rc = mnv.lib().mnv_render_voxels_accel(C.c_void_p(tree.accel), C.byref(cam.c), C.byref(opt), mnv.Rect(0, 0, 70000, 70000), None, None, None)
    assert rc == mnv.MNV_E_UNSUPPORTED and "2^32" in mnv.lib().mnv_last_error().decode()


@pytest.fixture(scope="module")
def cfg2(mnv, torch_gpu):
    tree = cases.make_tree(mnv, cases.CFG2_TREE)
    tree.move_to_device()
    return tree


def test_cfg2_full_size_accel_equals_ref_layout_and_oracle(mnv, orc, torch_gpu, cfg2):
    """BASELINE.json configs[1] at full size (1920x1080, depth 10, SH9): the tuned kernel, the
    reference-layout kernel and the CPU oracle agree bit for bit on every pixel."""
    cam = cases.cfg2_camera(mnv, pose=3)
    opt = mnv.RenderOptions.cli_defaults()
    a, _ = _render_gpu(mnv, torch_gpu, cfg2, cam, opt, "accel")
    b, _ = _render_gpu(mnv, torch_gpu, cfg2, cam, opt, "ref_layout")          # 2 M rays: with the per-launch level-7 table
    assert np.array_equal(cases.bits(a), cases.bits(b))
    b, _ = _render_gpu(mnv, torch_gpu, cfg2, cam, opt, "ref_layout_walk")     # per-workgroup level-3 table only
    assert np.array_equal(cases.bits(a), cases.bits(b))
    ref = orc.render(orc.tree_from_view(cfg2.host_view()), cam.c, opt)
    assert np.array_equal(cases.bits(a), cases.bits(ref["rgba"]))
    c = ref["counters"].as_dict()
    assert c["rays_hit"] > 0.2 * c["rays"]  # the shell is actually in view


def test_cfg2_tile_partition_invariance_and_determinism(mnv, torch_gpu, cfg2):
    """Size-independent properties at full size: rendering the frame as 8 interleaved tile sets
    (the multi-GPU partition) reproduces the single-launch frame bit for bit, and two launches of
    the (work-stealing, order-nondeterministic) kernel give identical images."""
    torch = torch_gpu
    cam = cases.cfg2_camera(mnv, pose=7)
    opt = mnv.RenderOptions.cli_defaults()
    full, _ = _render_gpu(mnv, torch, cfg2, cam, opt, "accel")
    again, _ = _render_gpu(mnv, torch, cfg2, cam, opt, "accel")
    assert np.array_equal(cases.bits(full), cases.bits(again))
    th, tw = 120, 240
    tiles = [(x, y, tw, th) for y in range(0, 1080, th) for x in range(0, 1920, tw)]
    out = np.empty_like(full)
    for rank in range(8):
        for (x, y, w, h) in tiles[rank::8]:
            got, _ = _render_gpu(mnv, torch, cfg2, cam, opt, "accel", tile=(x, y, w, h))
            out[y:y + h, x:x + w] = got
    assert np.array_equal(cases.bits(out), cases.bits(full))
    # alpha is an opacity, rgb is bounded by the compositing weights
    assert full[..., 3].min() >= 0.0 and full[..., 3].max() <= 1.0
    assert np.isfinite(full).all()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_cfg2_interleaved_partition_reassembles_bit_exact(mnv, torch_gpu, cfg2, world):
    """The single-launch multi-GPU partition (mnv_render_voxels_accel_part): every rank's compact
    local-tile-major buffer, un-permuted the way rank 0 does after the RCCL gather, reproduces the
    plain full-frame render bit for bit (float and u8), including ragged tile counts (world = 3)
    and ragged right/bottom macro tiles (tile size that does not divide the frame)."""
    torch = torch_gpu
    cam = cases.cfg2_camera(mnv, pose=5)
    opt = mnv.RenderOptions.cli_defaults()
    full, full8 = _render_gpu(mnv, torch, cfg2, cam, opt, "accel", want_u8=True)
    H, W = cam.height, cam.width
    from mega_nerf_viewer_amd.multigpu import TilePartition
    for (tw, th, M) in [(64, 24, 0), (64, 24, 6), (128, 120, 0), (200, 136, 2)]:
        mx, my = -(-W // tw), -(-H // th)
        part = TilePartition(W, H, world, tw, th, M)
        out = np.full((my * th, mx * tw, 4), np.nan, np.float32)
        out8 = np.zeros((my * th, mx * tw, 4), np.uint8)
        for rank in range(world):
            n_local = mnv.partition_local_tiles((0, 0, W, H), rank, world, tw, th, M)
            assert n_local == part.local_tiles(rank) and (M or n_local == len(range(rank, mx * my, world)))
            buf = torch.full((max(n_local, 1), th, tw, 4), float("nan"), dtype=torch.float32, device="cuda")
            buf8 = torch.zeros((max(n_local, 1), th, tw, 4), dtype=torch.uint8, device="cuda")
            mnv.render_voxels_accel_part(cfg2.accel, cam, opt, rank, world, tw, th, rgba=buf, rgba8=buf8, root_period=M)
            torch.cuda.synchronize()
            b, b8 = buf.cpu().numpy(), buf8.cpu().numpy()
            for j, m in enumerate(part.tiles_of(rank)):
                MX, MY = m % mx, m // mx
                out[MY * th:(MY + 1) * th, MX * tw:(MX + 1) * tw] = b[j]
                out8[MY * th:(MY + 1) * th, MX * tw:(MX + 1) * tw] = b8[j]
        assert np.array_equal(cases.bits(out[:H, :W]), cases.bits(full)), (world, tw, th, M)
        assert np.array_equal(out8[:H, :W], full8)
        if my * th > H:  # pixels outside the frame are never written
            assert np.isnan(out[H:, :]).all()


@pytest.mark.parametrize("name", list(cases.CASES))
def test_hip_kernel_matches_reference_goldens(mnv, torch_gpu, name):
    """The tuned kernel against the committed outputs of the reference's own device code
    (tests/golden/README.md).  Contract 1e-4 per channel; asserted 1e-6 with no outliers."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"ref_{name}.npz"))
    spec = cases.CASES[name]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    tree.move_to_device()
    got, _ = _render_gpu(mnv, torch_gpu, tree, cam, opt, "accel")
    assert np.abs(got.astype(np.float64) - g["rgba"].astype(np.float64)).max() <= 1e-6


def test_cfg2_hip_kernel_vs_live_reference_build(mnv, torch_gpu, cfg2, tmp_path):
    """When oracle/_ref/ travelled to the GPU box: the reference's own render_voxels_trace_ray,
    compiled for gfx950, rendered live at full size against the tuned kernel."""
    mnv_ref = require_live_reference()
    cam = cases.cfg2_camera(mnv, pose=11)
    opt = mnv.RenderOptions.cli_defaults()
    path = str(tmp_path / "cfg2.npz")
    cfg2.save_npz(path)
    ref = mnv_ref.render_npz(path, cam.c, opt)["rgba"]
    got, _ = _render_gpu(mnv, torch_gpu, cfg2, cam, opt, "accel")
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64)).max(axis=-1)
    assert (d > 1e-4).sum() == 0 and d.max() <= 1e-6, f"max|d| {d.max():.3e}, {(d > 1e-6).sum()} px > 1e-6"


def test_cfg2_batched_launch_equals_per_frame_launches(mnv, torch_gpu, cfg2):
    """mnv_render_voxels_accel_batch: several cameras in one launch (wavefronts walk the frames at their
    own pace) give exactly the frames of one launch per camera -- plain and under a partition."""
    torch = torch_gpu
    opt = mnv.RenderOptions.cli_defaults()
    cams = [cases.cfg2_camera(mnv, pose=p, width=960, height=544, fx=800.0) for p in (0, 5, 9, 14, 3)]
    H, W = 544, 960
    batch = torch.full((len(cams), H, W, 4), float("nan"), dtype=torch.float32, device="cuda")
    batch8 = torch.zeros((len(cams), H, W, 4), dtype=torch.uint8, device="cuda")
    mnv.render_voxels_accel_batch(cfg2.accel, cams, opt, rgba=batch, rgba8=batch8)
    torch.cuda.synchronize()
    b, b8 = batch.cpu().numpy(), batch8.cpu().numpy()
    for i, cam in enumerate(cams):
        one, one8 = _render_gpu(mnv, torch, cfg2, cam, opt, "accel", want_u8=True)
        assert np.array_equal(cases.bits(b[i]), cases.bits(one)), i
        assert np.array_equal(b8[i], one8), i
    # partition: rank 2 of 3 owns one macro tile fewer than ranks 0/1 (20 tiles): frames are still
    # j_max = ceil(20 / 3) = 7 local tiles apart
    tw, th, world, rank = 200, 136, 3, 2
    n_local = mnv.partition_local_tiles((0, 0, W, H), rank, world, tw, th)
    j_max = -(-((-(-W // tw)) * (-(-H // th))) // world)
    assert n_local == j_max - 1
    pb = torch.full((len(cams), j_max, th, tw, 4), float("nan"), dtype=torch.float32, device="cuda")
    mnv.render_voxels_accel_batch(cfg2.accel, cams, opt, part=(rank, world, tw, th), rgba=pb)
    torch.cuda.synchronize()
    for i, cam in enumerate(cams):
        single = torch.full((n_local, th, tw, 4), float("nan"), dtype=torch.float32, device="cuda")
        mnv.render_voxels_accel_part(cfg2.accel, cam, opt, rank, world, tw, th, rgba=single)
        torch.cuda.synchronize()
        a, s = pb[i, :n_local].cpu().numpy(), single.cpu().numpy()
        assert np.isnan(pb[i, n_local:].cpu().numpy()).all()
        assert np.array_equal(np.isnan(a), np.isnan(s))
        assert np.array_equal(cases.bits(np.nan_to_num(a)), cases.bits(np.nan_to_num(s))), i
    with pytest.raises(mnv.MnvError):
        mnv.render_voxels_accel_batch(cfg2.accel, [cams[0], cases.cfg2_camera(mnv, 0)], opt, rgba=batch)  # mixed sizes


def test_cfg3_merged_octree_standin_full_size(mnv, orc, torch_gpu):
    """BASELINE.json configs[2] stand-in at full size: anisotropic (invradius3 = 0.5, 0.125, 0.125) 4x2-brick
    terrain tree, 2.7 M chunks, oblique aerial camera, 1920x1080 -- tuned kernel == oracle bit for bit."""
    tree = cases.make_tree(mnv, cases.CFG3_TREE)
    assert tree.capacity > 2_000_000
    cam = cases.cfg3_camera(mnv, pose=2)
    opt = mnv.RenderOptions.cli_defaults()
    ref = orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt)
    assert ref["counters"].rays_hit > 0.4 * ref["counters"].rays
    tree.move_to_device()
    got, _ = _render_gpu(mnv, torch_gpu, tree, cam, opt, "accel")
    assert np.array_equal(cases.bits(got), cases.bits(ref["rgba"]))


_STRESS = {
    # axis-aligned camera: direction components that are exactly 0 -> invdir = 1 / 1e-9
    "axis_aligned": dict(camera=dict(width=96, height=96, fx=300.0, center=(0.0, 0.0, 4.0), back=(0.0, 0.0, 1.0), up=(0.0, 1.0, 0.0)), options=dict()),
    # negative sigma threshold: every leaf, including sigma == 0 ones, is shaded
    "all_leaves_dense": dict(camera=dict(width=96, height=72, fx=250.0), options=dict(sigma_thresh=-1.0)),
    # never stop early / stop at the first sample
    "no_early_stop": dict(camera=dict(width=96, height=72, fx=250.0), options=dict(stop_thresh=0.0)),
    "stop_immediately": dict(camera=dict(width=96, height=72, fx=250.0), options=dict(stop_thresh=1.0)),
    # a small and a huge epsilon step (a step_size below the float spacing of t, e.g. 0 or 1e-7 at t ~ 4,
    # stalls the march on a cell face forever -- in the reference as well -- and is not a valid input)
    "small_step": dict(camera=dict(width=64, height=48, fx=200.0), options=dict(step_size=1e-5)),
    "huge_step": dict(camera=dict(width=96, height=72, fx=250.0), options=dict(step_size=0.25)),
    # degenerate / inverted bounding boxes and a slab thinner than a voxel
    "empty_bbox": dict(camera=dict(width=64, height=48, fx=200.0), options=dict(render_bbox=(0.6, 0.6, 0.6, 0.4, 0.4, 0.4))),
    "thin_slab": dict(camera=dict(width=96, height=72, fx=250.0), options=dict(render_bbox=(0.0, 0.0, 0.4999, 1.0, 1.0, 0.5001))),
    # wide field of view from inside a voxel corner, off-centre principal point, non-square pixels
    "inside_wide": dict(camera=dict(width=120, height=80, fx=40.0, fy=25.0, cx=10.0, cy=70.0, center=(0.01, 0.02, -0.03), back=(0.3, -0.5, 0.81)),
                        options=dict(background_brightness=0.3)),
    # basis window that removes every basis function, and one beyond the basis count
    "no_basis": dict(camera=dict(width=96, height=72, fx=250.0), options=dict(basis_minmax=(5, 2))),
    "basis_window_high": dict(camera=dict(width=96, height=72, fx=250.0), options=dict(basis_minmax=(3, 40))),
    # large rotation of the view directions
    "rot_pi": dict(camera=dict(width=96, height=72, fx=250.0), options=dict(rot_dirs=(0.0, 3.14159274, 0.0))),
}


@pytest.mark.parametrize("name", list(_STRESS))
def test_stress_options_bit_exact(mnv, orc, torch_gpu, name):
    """Corner cases of camera and RenderOptions on an SH9 tree: both kernels == oracle, bit for bit."""
    spec = _STRESS[name]
    tree = cases.make_tree(mnv, dict(kind="random", depth=6, basis_dim=9, refine_prob=0.55, empty_prob=0.5, sigma_max=50.0, coef_sd=1.2, seed=21))
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, dict(spec["options"], base="cli"))
    ref = orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt, want_rgba8=True)
    assert np.isfinite(ref["rgba"]).all()
    tree.move_to_device()
    for which in ("ref_layout", "accel"):
        got, got8 = _render_gpu(mnv, torch_gpu, tree, cam, opt, which, want_u8=True)
        assert np.array_equal(cases.bits(got), cases.bits(ref["rgba"])), (name, which, float(np.abs(got - ref["rgba"]).max()))
        assert np.array_equal(got8, ref["rgba8"]), (name, which)


def test_batch_limits(mnv, torch_gpu):
    torch = torch_gpu
    tree = cases.make_tree(mnv, cases.CASES["sh4_d6"]["tree"])
    tree.move_to_device()
    opt = mnv.RenderOptions.cli_defaults()
    cams = [mnv.orbit_camera(64, 40, 90.0, 3.0, 5.0 * i, 15.0) for i in range(mnv.MAX_BATCH)]
    out = torch.full((mnv.MAX_BATCH, 40, 64, 4), float("nan"), dtype=torch.float32, device="cuda")
    mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba=out)  # the maximum batch
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert np.isfinite(o).all()
    for i in (0, 31, 63):
        one = torch.empty((40, 64, 4), dtype=torch.float32, device="cuda")
        mnv.render_voxels_accel(tree.accel, cams[i], opt, rgba=one)
        torch.cuda.synchronize()
        assert np.array_equal(cases.bits(o[i]), cases.bits(one.cpu().numpy()))
    with pytest.raises(mnv.MnvError):
        mnv.render_voxels_accel_batch(tree.accel, cams + cams[:1], opt, rgba=out)  # 65 cameras
    with pytest.raises(mnv.MnvError):
        mnv.render_voxels_accel_batch(tree.accel, [], opt, rgba=out)


@pytest.mark.parametrize("depth,refine,basis", [(1, 0.0, 9), (2, 0.9, 4), (3, 0.8, 1), (12, 0.33, 9), (14, 0.3, 4)])
def test_tree_depth_extremes_bit_exact(mnv, orc, torch_gpu, depth, refine, basis):
    """Trees shallower than the lookup grids (root chunk only, depth 2-3) and much deeper than them
    (depth 12 / 14: the level-2 grid is capped by its memory budget, several node loads follow)."""
    tree = cases.make_tree(mnv, dict(kind="random", depth=depth, basis_dim=basis, refine_prob=refine, empty_prob=0.4, sigma_max=60.0, seed=100 + depth))
    cam = mnv.Camera(160, 120, 420.0).set_pose((-2.6, 1.3, 1.9), (-0.75, 0.37, 0.55))
    opt = mnv.RenderOptions.cli_defaults()
    ref = orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt)
    assert ref["counters"].hits > 0
    tree.move_to_device()
    for which in ("ref_layout", "accel"):
        got, _ = _render_gpu(mnv, torch_gpu, tree, cam, opt, which)
        assert np.array_equal(cases.bits(got), cases.bits(ref["rgba"])), (depth, which)


@pytest.mark.parametrize("depth,basis,fmt_kw", [(10, 9, {}), (11, 4, {}), (12, 9, {}), (13, 1, {}), (11, -1, dict(fmt=0)), (11, 16, {})])
def test_bricks_below_the_second_grid_bit_exact(mnv, orc, torch_gpu, depth, basis, fmt_kw):
    """Trees with leaves two or more levels below the second lookup grid carry bricks (the two levels below the grid in one 8-byte
    load: AccelView::recs): plain, depth and tracker frames read them and stay bit-identical to the oracle, across a tree edit
    (mnv_accel_refresh patches them) and a rebuild."""
    torch = torch_gpu
    spec = dict(kind="random", depth=depth, basis_dim=basis, refine_prob=0.42, empty_prob=0.93, sigma_max=40.0, seed=300 + depth + basis, **fmt_kw)
    tree = cases.make_tree(mnv, spec)
    cam = mnv.Camera(200, 144, 500.0).set_pose((-2.5, 1.4, 1.8), (-0.74, 0.4, 0.54))
    opt = mnv.RenderOptions.cli_defaults()
    t = orc.tree_from_view(tree.host_view())
    ref = orc.render(t, cam.c, opt)
    assert ref["counters"].hits > 5000 and ref["counters"].levels > 3.5 * ref["counters"].steps   # mostly empty: rays go deep
    tree.move_to_device(need_parent=True)
    info = mnv.accel_info(tree.accel)
    assert info["brick_levels"] == 2 and depth >= info["grid2_level"] + 2, info
    got, _ = _render_gpu(mnv, torch, tree, cam, opt, "accel")
    assert np.array_equal(cases.bits(got), cases.bits(ref["rgba"]))
    if basis != -1:
        opt_d = mnv.RenderOptions.cli_defaults()
        opt_d.render_depth = True
        want_d = orc.render(t, cam.c, opt_d)["rgba"]
        got_d, _ = _render_gpu(mnv, torch, tree, cam, opt_d, "accel")
        assert np.array_equal(cases.bits(got_d), cases.bits(want_d))
    # a negative sigma_thresh makes EMPTY leaves dense samples (0 > thresh: weight 0, the colour row is still read) -- the inline words and
    # records cannot answer that, the launch falls back to the node words
    opt_n = mnv.RenderOptions.cli_defaults()
    opt_n.sigma_thresh = -0.5
    want_n = orc.render(t, cam.c, opt_n)
    assert want_n["counters"].hits > 3 * ref["counters"].hits
    got_n, _ = _render_gpu(mnv, torch, tree, cam, opt_n, "accel")
    assert np.array_equal(cases.bits(got_n), cases.bits(want_n["rgba"]))
    # a sub-rectangle and a batch of two cameras through the same kernel
    cam2 = mnv.Camera(200, 144, 500.0).set_pose((2.2, -1.9, 1.1), (0.7, -0.6, 0.39))
    ref2 = orc.render(t, cam2.c, opt)["rgba"]
    out = torch.empty((2, cam.height, cam.width, 4), dtype=torch.float32, device="cuda")
    mnv.render_voxels_accel_batch(tree.accel, [cam, cam2], opt, rgba=out)
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(out[0].cpu().numpy()), cases.bits(ref["rgba"])) and np.array_equal(cases.bits(out[1].cpu().numpy()), cases.bits(ref2))
    # a tree edit (mnv_accel_refresh) keeps both: the refresh patches the words of the cells and records the edit touches (until round 5 it
    # dropped them until a rebuild or sixteen plain frames in a row).  An edit that changes nothing -- one row rewritten with itself --
    # leaves every frame kind as it was; tests/test_accel_edit_gpu.py holds real edits against the reference-layout kernel.
    dv = tree.device_view()
    changed = torch.tensor([[0, 0]], dtype=torch.int32, device="cuda")
    mnv.accel_refresh(tree.accel, dv, dv.capacity, changed_nodes=changed)
    assert mnv.accel_info(tree.accel)["brick_levels"] == 2
    got, _ = _render_gpu(mnv, torch, tree, cam, opt, "accel")
    assert np.array_equal(cases.bits(got), cases.bits(ref["rgba"]))
    mnv.accel_rebuild(tree.accel, dv)
    assert mnv.accel_info(tree.accel)["brick_levels"] == 2
    got, _ = _render_gpu(mnv, torch, tree, cam, opt, "accel")
    assert np.array_equal(cases.bits(got), cases.bits(ref["rgba"]))
    # tracker frames read the inline words and records too (SH16 rows go through the cooperative colour pass, which has no such variant):
    # pixels, tracker rows (voxel numbers!) against the oracle
    counts = np.full((tree.host_view().capacity, 8), 8, np.int16)
    opt_t = mnv.RenderOptions.cli_defaults()
    opt_t.max_depth, opt_t.max_sample_count = depth - 1, 9
    want_t = orc.render(orc.tree_from_view(tree.host_view(), sample_counts=counts), cam.c, opt_t, want_trackers=True)
    split = torch.full((cam.height, cam.width, 3), -1.0, dtype=torch.float32, device="cuda")
    sample = torch.full((cam.height, cam.width, 3), -1.0, dtype=torch.float32, device="cuda")
    out_t = torch.full((cam.height, cam.width, 4), float("nan"), dtype=torch.float32, device="cuda")
    mnv.render_voxels_accel_track(tree.accel, cam, opt_t, rgba=out_t, split_track=split, sample_track=sample, sample_counts=torch.from_numpy(counts).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(out_t.cpu().numpy()), cases.bits(want_t["rgba"]))
    assert np.array_equal(split.cpu().numpy(), want_t["split"]) and np.array_equal(sample.cpu().numpy(), want_t["sample"])


@pytest.mark.parametrize("step", [1e-4, 1.5e-3, 3e-3, 5e-2])
def test_extreme_opacity_exercises_expf_tails(mnv, orc, torch_gpu, step):
    """sigma up to the binary16 maximum: the opacity exponent -dt * scale * sigma sweeps through the
    denormal (-87 .. -104) and underflow (< -104) branches of expf on the device, bit for bit."""
    tree = cases.make_tree(mnv, dict(kind="random", depth=5, basis_dim=4, refine_prob=0.6, empty_prob=0.3, sigma_max=65000.0, coef_sd=30.0, seed=77))
    cam = mnv.Camera(128, 96, 300.0)
    opt = mnv.RenderOptions.cli_defaults()
    opt.step_size, opt.stop_thresh = step, 0.0  # never stop early: T itself goes denormal / zero
    ref = orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt)
    assert np.isfinite(ref["rgba"]).all()
    tree.move_to_device()
    for which in ("ref_layout", "accel"):
        got, _ = _render_gpu(mnv, torch_gpu, tree, cam, opt, which)
        assert np.array_equal(cases.bits(got), cases.bits(ref["rgba"])), (step, which, float(np.abs(got - ref["rgba"]).max()))


@pytest.mark.parametrize("world,w,h,frames,root_period", [(3, 1920, 1080, 2, 0), (8, 1000, 700, 0, 0), (2, 136, 128, 3, 0), (5, 3840, 2160, 1, 0),
                                                         (8, 1920, 1080, 2, 6), (3, 1000, 700, 0, 2)])
def test_assemble_tiles_kernel_equals_index_permutation(mnv, torch_gpu, world, w, h, frames, root_period):
    """mnv_assemble_tiles (rank 0's un-permute after the gather) against the torch index_select / permute form of
    TilePartition.unpermute, RGBA8 and float RGBA, with and without the frame dimension, ragged tile counts."""
    torch = torch_gpu
    from mega_nerf_viewer_amd.multigpu import TilePartition
    part = TilePartition(w, h, world, 128, 120, root_period)
    lead = (frames,) if frames else ()
    for dt in (torch.uint8, torch.float32):
        g = torch.randint(0, 255, (world,) + lead + (part.j_max, 120, 128, 4), device="cuda").to(dt)
        want = part.unpermute(g)  # torch path (out=None)
        out = torch.zeros(lead + (h, w, 4), dtype=dt, device="cuda")
        got = part.unpermute(g, out=out)
        torch.cuda.synchronize()
        assert got is out and torch.equal(out, want)


@pytest.mark.parametrize("name", sorted(f[len("ref_trackers_"):-4] for f in os.listdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
                                        if f.startswith("ref_trackers_")))
def test_trackers_match_reference_device_code_goldens(mnv, torch_gpu, name):
    """Both march kernels against the tracker rows / visit marks the reference's own device code wrote on gfx950."""
    torch = torch_gpu
    from test_goldens import tracker_setup
    z, tree, cam, opt, counts = tracker_setup(mnv, name)
    tree.move_to_device()
    dv = tree.device_view()
    sc = torch.from_numpy(counts).cuda()
    dv.sample_counts = sc.data_ptr()
    h, w = cam.height, cam.width
    new = lambda: torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")  # noqa: E731
    split, sample, visited = new(), new(), torch.zeros(dv.capacity, dtype=torch.int32, device="cuda")
    mnv.render_voxels(dv, cam, opt, split_track=split, sample_track=sample, visited=visited, track_visit=True)
    split2, sample2 = new(), new()
    mnv.render_voxels_accel_track(tree.accel, cam, opt, split_track=split2, sample_track=sample2, sample_counts=sc)
    torch.cuda.synchronize()
    for got, want in ((split, "split"), (sample, "sample"), (split2, "split"), (sample2, "sample")):
        assert np.array_equal(got.cpu().numpy(), z[want])
    assert np.array_equal(visited.cpu().numpy(), z["visited"])


@pytest.mark.parametrize("name", ["sh4_d6", "sh9_d7_aniso", "shell_d7_sh9", "sh25_d4", "thresholds", "rgba_d5"])
def test_fast_colour_math_keeps_alpha_and_control_flow_exact(mnv, orc, torch_gpu, name):
    """mnv_accel_set_colour_math(accel, 1): hardware exp2 / rcp in the colour sigmoid only.  Alpha (hence every opacity, transmittance and
    early-stop decision) stays bit-identical to the oracle; colours stay within 2e-6 (contract: 1e-4)."""
    torch = torch_gpu
    spec = cases.CASES[name]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    ref = orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt)["rgba"]
    tree.move_to_device()
    out = torch.empty((cam.height, cam.width, 4), dtype=torch.float32, device="cuda")
    mnv.accel_set_colour_math(tree.accel, 1)
    mnv.render_voxels_accel(tree.accel, cam, opt, rgba=out)
    torch.cuda.synchronize()
    mnv.accel_set_colour_math(tree.accel, 0)
    got = out.cpu().numpy()
    assert np.array_equal(cases.bits(got[..., 3]), cases.bits(ref[..., 3]))
    assert np.abs(got[..., :3] - ref[..., :3]).max() <= 2e-6
    if name != "rgba_d5":  # RGBA rows have no sigmoid: nothing changes there
        assert not np.array_equal(cases.bits(got), cases.bits(ref))
    else:
        assert np.array_equal(cases.bits(got), cases.bits(ref))
    mnv.render_voxels_accel(tree.accel, cam, opt, rgba=out)  # back in exact mode: bit-identical again
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(ref))


def test_two_accels_of_one_process_can_differ_in_colour_math(mnv, orc, torch_gpu):
    """mnv_accel_set_colour_math: the choice belongs to the accel (there is no process-wide switch) -- one accel renders with the hardware
    exp2 / rcp colour sigmoid (alpha bit-identical, colours within 2e-6) while another accel of the same process stays exact, and 0 makes the
    first exact again."""
    torch = torch_gpu
    spec = cases.CASES["shell_d7_sh9"]
    cam, opt = cases.make_camera(mnv, spec["camera"]), cases.make_options(mnv, spec["options"])
    ta, tb = cases.make_tree(mnv, spec["tree"]), cases.make_tree(mnv, spec["tree"])
    ref = orc.render(orc.tree_from_view(ta.host_view()), cam.c, opt)["rgba"]
    ta.move_to_device()
    tb.move_to_device()
    out_a = torch.empty((cam.height, cam.width, 4), dtype=torch.float32, device="cuda")
    out_b = torch.empty_like(out_a)
    mnv.accel_set_colour_math(ta.accel, 1)
    mnv.render_voxels_accel(ta.accel, cam, opt, rgba=out_a)
    mnv.render_voxels_accel(tb.accel, cam, opt, rgba=out_b)
    torch.cuda.synchronize()
    a, b = out_a.cpu().numpy(), out_b.cpu().numpy()
    assert np.array_equal(cases.bits(b), cases.bits(ref))
    assert np.array_equal(cases.bits(a[..., 3]), cases.bits(ref[..., 3])) and not np.array_equal(cases.bits(a), cases.bits(ref))
    assert np.abs(a - ref).max() < 2e-6
    mnv.accel_set_colour_math(ta.accel, 0)    # exact again
    mnv.accel_set_colour_math(tb.accel, 1)    # ... and the other one fast
    mnv.render_voxels_accel(ta.accel, cam, opt, rgba=out_a)
    mnv.render_voxels_accel(tb.accel, cam, opt, rgba=out_b)
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(out_a.cpu().numpy()), cases.bits(ref)) and not np.array_equal(cases.bits(out_b.cpu().numpy()), cases.bits(ref))


@pytest.mark.parametrize("name", ["sh4_d6", "sh9_d7_aniso", "rgba_d5", "terrain_d7_aniso"])
def test_reference_binding_is_a_drop_in(mnv, orc, torch_gpu, tmp_path, name):
    """include/mnv_reference_binding.hpp compiled inside a build of the reference (oracle/Makefile.ref): the reference's OWN loader,
    N3Tree (libtorch tensors on the device) and Camera (glm) feed libmnv.so -- mnv_render_voxels with trackers and visit marks,
    and the packed accel.  Frames, trackers and marks equal the oracle's bit for bit: the C ABI is a drop-in for
    viewer::render_voxels (include/cuda/renderer_kernel.hpp:23-34)."""
    mnv_ref = require_live_reference()
    spec = cases.CASES[name]
    tree = cases.make_tree(mnv, spec["tree"])
    cs = dict(dict(center=(-3.55, 0.0, 3.55), back=(-0.7071068, 0.0, 0.7071068)), **spec["camera"])  # Camera ctor defaults if unposed
    cam = cases.make_camera(mnv, cs)
    opt = cases.make_options(mnv, spec["options"])
    opt.max_depth, opt.max_sample_count = 5, 9
    opt.basis_minmax[0], opt.basis_minmax[1] = 0, max(tree.host_view().basis_dim - 1, 0)
    path = str(tmp_path / "t.npz")
    tree.save_npz(path)
    v = tree.host_view()
    counts = np.full((v.capacity, 8), 8, np.int16)
    visited = np.zeros(v.capacity, np.int32)
    want = orc.render(orc.tree_from_view(v, sample_counts=counts), cam.c, opt, want_rgba8=True, want_trackers=True, visited=visited, track_visit=True)
    cam_spec = dict(width=cs["width"], height=cs["height"], fx=cs["fx"], center=cs["center"], back=cs["back"], up=cs.get("up", (0.0, 0.0, 1.0)))
    got = mnv_ref.dropin_render_npz(path, cam_spec, opt, v.capacity, path=0, want_trackers=True)
    assert np.array_equal(cases.bits(got["rgba"]), cases.bits(want["rgba"])) and np.array_equal(got["rgba8"], want["rgba8"])
    assert np.array_equal(got["split"], want["split"]) and np.array_equal(got["sample"], want["sample"])
    assert np.array_equal(got["visited"], visited)
    got = mnv_ref.dropin_render_npz(path, cam_spec, opt, v.capacity, path=1)
    assert np.array_equal(cases.bits(got["rgba"]), cases.bits(want["rgba"])) and np.array_equal(got["rgba8"], want["rgba8"])


def test_reserved_stream_renders_bit_identical_frames(mnv, torch_gpu, cfg2):
    """mnv_stream_create_reserved + mnv_accel_set_cu_budget (the multi-GPU path's CU-masked launch stream): frames, single and batched,
    are the frames of the ordinary stream; invalid reservations and budgets are refused; more launches than launch slots may be queued."""
    torch = torch_gpu
    opt = mnv.RenderOptions.cli_defaults()
    cams = [cases.cfg2_camera(mnv, pose=p) for p in (2, 9)]
    H, W = cams[0].height, cams[0].width
    want = torch.empty((2, H, W, 4), dtype=torch.float32, device="cuda")
    mnv.render_voxels_accel_batch(cfg2.accel, cams, opt, rgba=want)
    torch.cuda.synchronize()
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    for bad in (-1, n_cus):
        with pytest.raises(mnv.MnvError):
            mnv.stream_create_reserved(bad)
    with pytest.raises(mnv.MnvError):
        mnv.accel_set_cu_budget(cfg2.accel, n_cus + 1)
    handle, enabled = mnv.stream_create_reserved(32)
    try:
        assert enabled == n_cus - 32
        mnv.accel_set_cu_budget(cfg2.accel, enabled)
        st = torch.cuda.ExternalStream(handle)
        got = torch.full((2, H, W, 4), float("nan"), dtype=torch.float32, device="cuda")
        one = torch.full((H, W, 4), float("nan"), dtype=torch.float32, device="cuda")
        for _ in range(70):  # more than the 64 launch slots of the accel: the 65th launch waits for the first
            mnv.render_voxels_accel(cfg2.accel, cams[1], opt, rgba=one, stream=handle)
        mnv.render_voxels_accel_batch(cfg2.accel, cams, opt, rgba=got, stream=handle)
        st.synchronize()
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
        assert torch.equal(one.view(torch.int32), want[1].view(torch.int32))
    finally:
        mnv.accel_set_cu_budget(cfg2.accel, 0)
        torch.cuda.synchronize()
        mnv.stream_destroy(handle)


def test_single_rank_partition_layout(mnv, torch_gpu, cfg2):
    """world == 1 with a tile size: the macro-tile-major layout from one rank (what bench.py --force-dist gathers) un-permutes
    to the plain frame."""
    from mega_nerf_viewer_amd.multigpu import TilePartition
    torch = torch_gpu
    opt = mnv.RenderOptions.cli_defaults()
    cam = cases.cfg2_camera(mnv, pose=11)
    H, W = cam.height, cam.width
    full, _ = _render_gpu(mnv, torch, cfg2, cam, opt, "accel")
    part = TilePartition(W, H, 1, 64, 24)
    assert mnv.partition_local_tiles((0, 0, W, H), 0, 1, 64, 24) == part.n_macro == 1350
    buf = torch.full((1, part.j_max, 24, 64, 4), float("nan"), dtype=torch.float32, device="cuda")
    mnv.render_voxels_accel_part(cfg2.accel, cam, opt, 0, 1, 64, 24, rgba=buf[0])
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    part.unpermute(buf, out=out)
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(full))


def test_binding_rejects_output_tensors_that_are_too_small_or_of_the_wrong_kind(mnv, torch_gpu):
    """The library writes through raw pointers; the ctypes harness checks size, element type, contiguity and device of its tensors."""
    torch = torch_gpu
    spec = cases.CASES["sh4_d6"]
    tree = cases.make_tree(mnv, spec["tree"])
    tree.move_to_device()
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    h, w = cam.height, cam.width
    with pytest.raises(mnv.MnvError):
        mnv.render_voxels_accel(tree.accel, cam, opt, rgba=torch.empty((h - 1, w, 4), dtype=torch.float32, device="cuda"))
    with pytest.raises(mnv.MnvError):
        mnv.render_voxels_accel(tree.accel, cam, opt, rgba=torch.empty((h, w, 4), dtype=torch.float16, device="cuda"))
    with pytest.raises(mnv.MnvError):
        mnv.render_voxels_accel(tree.accel, cam, opt, rgba=torch.empty((h, w, 4), dtype=torch.float32))          # host tensor
    with pytest.raises(mnv.MnvError):
        mnv.render_voxels_accel(tree.accel, cam, opt, rgba8=torch.empty((h, w, 8), dtype=torch.uint8, device="cuda")[..., :4])  # strided
    n1 = mnv.partition_local_tiles((0, 0, w, h), 1, 3, 64, 24)
    with pytest.raises(mnv.MnvError):
        mnv.render_voxels_accel_part(tree.accel, cam, opt, 1, 3, 64, 24, rgba=torch.empty((n1 - 1, 24, 64, 4), dtype=torch.float32, device="cuda"))
    j_max = max(mnv.partition_local_tiles((0, 0, w, h), r, 3, 64, 24) for r in range(3))
    with pytest.raises(mnv.MnvError):
        mnv.render_voxels_accel_batch(tree.accel, [cam, cam], opt, part=(1, 3, 64, 24), rgba=torch.empty((2, j_max - 1, 24, 64, 4), dtype=torch.float32, device="cuda"))
    ok = torch.empty((2, j_max, 24, 64, 4), dtype=torch.float32, device="cuda")
    mnv.render_voxels_accel_batch(tree.accel, [cam, cam], opt, part=(1, 3, 64, 24), rgba=ok)
    torch.cuda.synchronize()


def test_tree_cache_of_the_stateless_entry_point(mnv, orc, torch_gpu):
    """mnv_set_tree_cache(1): mnv_render_voxels -- the literal replacement of viewer::render_voxels (renderer_kernel.hpp:23-34), arrays
    handed over with every call -- keeps the packed re-layout of the trees it has seen.  Frames are bit-identical to the stateless
    path and to the oracle; an in-place edit of the arrays shows up after mnv_tree_invalidate (and, by the rule in include/mnv.h, not
    before); frames with trackers run on the re-layout too, frames with visit marks walk the arrays; five trees through four cache entries."""
    torch = torch_gpu
    names = ["sh9_d7_aniso", "rgba_d5", "sh4_d6", "shell_d7_sh9", "cfg1_sh1_d4"]
    trees, views, cams, opts, refs = [], [], [], [], []
    for n in names:
        spec = cases.CASES[n]
        t = cases.make_tree(mnv, spec["tree"])
        refs.append(orc.render(orc.tree_from_view(t.host_view()), cases.make_camera(mnv, spec["camera"]).c, cases.make_options(mnv, spec["options"]))["rgba"])
        t.move_to_device()
        trees.append(t); views.append(t.device_view()); cams.append(cases.make_camera(mnv, spec["camera"])); opts.append(cases.make_options(mnv, spec["options"]))
    try:
        mnv.set_tree_cache(True)
        side = torch.cuda.Stream()
        for rnd in range(3):  # second and third round: cache hits (and re-builds of the evicted fifth tree), also from another stream
            for i in range(len(names)):
                out = torch.full((cams[i].height, cams[i].width, 4), float("nan"), dtype=torch.float32, device="cuda")
                mnv.render_voxels(views[i], cams[i], opts[i], rgba=out, stream=side.cuda_stream if rnd == 2 else 0)
                torch.cuda.synchronize()
                assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(refs[i])), (names[i], rnd)
        out = torch.empty((cams[0].height, cams[0].width, 4), dtype=torch.float32, device="cuda")
        # frames with trackers run on the kept re-layout as well (its tracker instantiation): rows equal to the oracle's; visit marks walk
        split = torch.full((cams[0].height, cams[0].width, 3), -1.0, dtype=torch.float32, device="cuda")
        sample = torch.full((cams[0].height, cams[0].width, 3), -1.0, dtype=torch.float32, device="cuda")
        opts[0].max_depth, opts[0].max_sample_count = 5, 9
        want_t = orc.render(orc.tree_from_view(trees[0].host_view()), cams[0].c, opts[0], want_trackers=True)
        mnv.render_voxels(views[0], cams[0], opts[0], rgba=out, split_track=split, sample_track=sample)
        torch.cuda.synchronize()
        assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(refs[0])) and bool((split[..., 1] >= 0).any())
        assert np.array_equal(split.cpu().numpy(), want_t["split"]) and np.array_equal(sample.cpu().numpy(), want_t["sample"])
        visited = torch.zeros(views[0].capacity, dtype=torch.int32, device="cuda")
        split.fill_(-1.0)
        mnv.render_voxels(views[0], cams[0], opts[0], rgba=out, split_track=split, visited=visited, track_visit=True)
        torch.cuda.synchronize()
        assert np.array_equal(split.cpu().numpy(), want_t["split"]) and int(visited.sum().item()) > 0
        # an in-place edit of the arrays (the caller's own device copies here): every sigma set to zero -> an empty picture, but only
        # once the cache has been told (include/mnv.h: THE RULE)
        h_data, h_child, _ = trees[0].host_arrays()
        d_data, d_child = torch.from_numpy(h_data.copy()).cuda(), torch.from_numpy(h_child.copy()).cuda()
        mine = type(views[0]).from_buffer_copy(views[0])
        mine.data, mine.child, mine.parent, mine.sample_counts = d_data.data_ptr(), d_child.data_ptr(), None, None
        mnv.render_voxels(mine, cams[0], opts[0], rgba=out)
        torch.cuda.synchronize()
        assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(refs[0]))
        d_data[..., -1] = 0
        mnv.render_voxels(mine, cams[0], opts[0], rgba=out)       # stale by the rule: still the old picture
        torch.cuda.synchronize()
        assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(refs[0]))
        mnv.tree_invalidate(d_child)
        mnv.render_voxels(mine, cams[0], opts[0], rgba=out)
        torch.cuda.synchronize()
        empty = out.cpu().numpy()
        assert np.all(empty[..., 3] == 0.0) and np.all(empty[..., :3] == opts[0].background_brightness)
        mnv.set_tree_cache(False)                                   # and the stateless path sees the same arrays
        mnv.render_voxels(mine, cams[0], opts[0], rgba=out)
        torch.cuda.synchronize()
        assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(empty))
    finally:
        mnv.set_tree_cache(False)


def test_tree_cache_takes_the_transform_from_the_call(mnv, orc, torch_gpu):
    """The cached re-layout is keyed by the arrays, not by offset / scale: the same device arrays under another transform (a re-centred
    scene) render with the transform of the view they are called with, bit-identical to the oracle -- not with the one the re-layout
    was built under."""
    torch = torch_gpu
    spec = cases.CASES["sh9_d7_aniso"]
    tree = cases.make_tree(mnv, spec["tree"])
    cam, opt = cases.make_camera(mnv, spec["camera"]), cases.make_options(mnv, spec["options"])
    hv = tree.host_view()
    tree.move_to_device()
    dv = tree.device_view()
    try:
        mnv.set_tree_cache(True)
        for k, (off, sc) in enumerate([(tuple(hv.offset), tuple(hv.scale)), ((0.45, 0.55, 0.5), (0.4, 0.3, 0.15)), ((0.5, 0.5, 0.5), (0.25, 0.25, 0.25))]):
            h2, d2 = type(hv).from_buffer_copy(hv), type(dv).from_buffer_copy(dv)
            for i in range(3):
                h2.offset[i] = d2.offset[i] = off[i]
                h2.scale[i] = d2.scale[i] = sc[i]
            want = orc.render(orc.tree_from_view(h2), cam.c, opt)["rgba"]
            out = torch.full((cam.height, cam.width, 4), float("nan"), dtype=torch.float32, device="cuda")
            mnv.render_voxels(d2, cam, opt, rgba=out)
            torch.cuda.synchronize()
            assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(want)), f"transform {k}"
    finally:
        mnv.set_tree_cache(False)


def test_tree_cache_remembers_a_tree_it_cannot_re_lay_out(mnv, orc, torch_gpu):
    """A tree deeper than the packed layout goes (24 levels): every frame takes the stateless walk, the failed build is not repeated
    per frame (second frame much faster than the first is not asserted -- only that frames stay right and other entries survive)."""
    torch = torch_gpu
    depth = 25
    child = np.zeros((depth, 2, 2, 2), np.int32)
    data = np.zeros((depth, 2, 2, 2, 4), np.float16)
    for c in range(depth - 1):
        child[c, 1, 0, 1] = 1          # a chain of chunks through one corner-ish voxel
    data[..., 3] = 0.0
    data[:, 0, 1, 0, :] = (0.8, 0.3, 0.1, 30.0)
    data[depth - 1, :, :, :, :] = (0.2, 0.9, 0.4, 500.0)
    deep = mnv.N3Tree.from_arrays(data, child, data_format="RGBA")
    spec = cases.CASES["sh4_d6"]
    other = cases.make_tree(mnv, spec["tree"])
    ocam, oopt = cases.make_camera(mnv, spec["camera"]), cases.make_options(mnv, spec["options"])
    cam = mnv.Camera(96, 96, 300.0)
    opt = mnv.RenderOptions.defaults()
    want_deep = orc.render(orc.tree_from_view(deep.host_view()), cam.c, opt)["rgba"]
    want_other = orc.render(orc.tree_from_view(other.host_view()), ocam.c, oopt)["rgba"]
    other.move_to_device()
    # (N3Tree.move_to_device builds the packed layout and would refuse this tree: the arrays go up by hand)
    d_data, d_child = torch.from_numpy(data.view(np.uint16).copy()).cuda(), torch.from_numpy(child.copy()).cuda()
    deep_view = type(deep.host_view()).from_buffer_copy(deep.host_view())
    deep_view.data, deep_view.child, deep_view.parent, deep_view.sample_counts = d_data.data_ptr(), d_child.data_ptr(), None, None
    try:
        mnv.set_tree_cache(True)
        for _ in range(3):
            out = torch.full((ocam.height, ocam.width, 4), float("nan"), dtype=torch.float32, device="cuda")
            mnv.render_voxels(other.device_view(), ocam, oopt, rgba=out)
            torch.cuda.synchronize()
            assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(want_other))
            out = torch.full((96, 96, 4), float("nan"), dtype=torch.float32, device="cuda")
            mnv.render_voxels(deep_view, cam, opt, rgba=out)
            torch.cuda.synchronize()
            assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(want_deep))
    finally:
        mnv.set_tree_cache(False)


def _n_ary_tree(n, depth, data_dim, refine_prob, seed):
    """A random N^3-ary tree in the reference's array layout (n3tree.cpp:92-107,183-188): child[chunk][N^3] relative chunk offsets
    (0 = leaf), data[chunk][N^3][data_dim] binary16 with sigma in the last column.  Breadth-first, like svox writes them."""
    rng = np.random.default_rng(seed)
    n3 = n ** 3
    children = [np.zeros(n3, np.int32)]
    level = [0]
    for d in range(1, depth):
        nxt = []
        for c in level:
            for k in range(n3):
                if rng.random() < refine_prob:
                    children.append(np.zeros(n3, np.int32))
                    children[c][k] = len(children) - 1 - c
                    nxt.append(len(children) - 1)
        level = nxt
    child = np.stack(children)
    cap = child.shape[0]
    data = (rng.standard_normal((cap, n3, data_dim)) * 1.2).astype(np.float16)
    sigma = rng.uniform(0.0, 40.0, (cap, n3)) * (rng.random((cap, n3)) < 0.45)
    data[..., -1] = sigma.astype(np.float16)
    return np.ascontiguousarray(child), np.ascontiguousarray(data.view(np.uint16))


@pytest.mark.parametrize("n,depth,basis,fmt", [(3, 4, 4, 1), (3, 3, -1, 0), (4, 3, 9, 1), (5, 2, 1, 1)])
def test_general_branching_factor_bit_exact(mnv, orc, torch_gpu, n, depth, basis, fmt):
    """rt_core.cuh:137-143 descends with `tree.N`, not 2: mnv_render_voxels takes N^3-ary trees through the general walk of the
    reference-layout kernel (no lookup tables; cube size N^depth by repeated multiplication, as the oracle) -- pixels, RGBA8 and the
    tracker rows bit for bit against the oracle.  (PlenOctree files are N = 2; the reference's loader warns about anything else,
    n3tree.cpp:85-87; the packed accel, the sample march and the refinement kernels stay N == 2.)"""
    torch = torch_gpu
    dd = 3 * basis + 1 if fmt == 1 else 4
    child, data = _n_ary_tree(n, depth, dd, 0.35 if n < 5 else 0.5, seed=n * 10 + depth)
    v = mnv.TreeView()
    v.N, v.data_dim, v.format, v.basis_dim, v.capacity = n, dd, fmt, basis, child.shape[0]
    for i in range(3):
        v.offset[i], v.scale[i] = 0.5, 0.5
    v.data, v.child = data.ctypes.data, child.ctypes.data
    cam = mnv.Camera(160, 120, 260.0).set_pose((-2.2, 1.3, 1.7), (-0.72, 0.42, 0.55))
    opt = mnv.RenderOptions.cli_defaults()
    opt.basis_minmax[0], opt.basis_minmax[1] = 0, max(basis - 1, 0)
    opt.max_depth, opt.max_sample_count = depth, 6
    sc = np.random.default_rng(3).integers(0, 12, size=(child.shape[0], n ** 3)).astype(np.int16)
    ref = orc.render(orc.tree_from_view(v, sample_counts=sc), cam.c, opt, want_rgba8=True, want_trackers=True)
    assert ref["counters"].rays_hit > 0.3 * 160 * 120 and ref["counters"].max_steps > 4
    d_data, d_child, d_sc = torch.from_numpy(data.view(np.int16)).cuda(), torch.from_numpy(child).cuda(), torch.from_numpy(sc).cuda()
    dv = type(v).from_buffer_copy(v)
    dv.data, dv.child, dv.sample_counts = d_data.data_ptr(), d_child.data_ptr(), d_sc.data_ptr()
    out = torch.full((120, 160, 4), float("nan"), dtype=torch.float32, device="cuda")
    out8 = torch.zeros((120, 160, 4), dtype=torch.uint8, device="cuda")
    split = torch.full((120, 160, 3), -1.0, dtype=torch.float32, device="cuda")
    sample = torch.full((120, 160, 3), -1.0, dtype=torch.float32, device="cuda")
    mnv.render_voxels(dv, cam, opt, rgba=out, rgba8=out8, split_track=split, sample_track=sample)
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(ref["rgba"]))
    assert np.array_equal(out8.cpu().numpy(), ref["rgba8"])
    assert np.array_equal(split.cpu().numpy(), ref["split"]) and np.array_equal(sample.cpu().numpy(), ref["sample"])
    assert (ref["split"][..., 1] >= 0).any()
    # the paths that stay N == 2 say so
    with pytest.raises(mnv.MnvError) as e:
        mnv.accel_create(dv)
    assert e.value.code == mnv.MNV_E_UNSUPPORTED

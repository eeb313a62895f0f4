"""BASELINE.json configs[3] and configs[4] at their stated sizes on ONE MI355X (the driver-run suite has no 8-GPU box):

* cfg4 -- the merged-octree stand-in at 3840x2160: the tuned kernel equals the oracle on the full 8.3 M-pixel frame, and the
  eight ranks of the multi-GPU run (interleaved 64x24 macro tiles, root-relieving deal, batched launch per rank, gather table,
  mnv_assemble_tiles) are executed one after the other on this GPU and reassemble the same frames byte for byte;
* cfg5 -- dynamic refinement and guided sampling on the 1.5 M-chunk depth-10 SH9 tree at 1920x1080 through VolumeRenderer
  (mnv_renderer_*): the tree stays a valid tree, the packed accel follows it (tuned kernel == reference-layout kernel on the
  grown tree), and the guided-sampling frame equals the reference's tensor-op formulation evaluated by torch.
"""
import ctypes as C

import numpy as np
import pytest

import cases
import mlp_cases
from test_renderer_refine_gpu import check_tree_links, make_grid

pytestmark = pytest.mark.gpu


def test_cfg4_merged_octree_4k_full_frame_and_world8_reassembly(mnv, orc, torch_gpu):
    torch = torch_gpu
    from mega_nerf_viewer_amd.multigpu import TilePartition

    W, H, world, tw, th, period = 3840, 2160, 8, 64, 24, 8
    tree = cases.make_tree(mnv, cases.CFG3_TREE)
    assert tree.capacity > 2_000_000
    ot = orc.tree_from_view(tree.host_view())
    tree.move_to_device()
    opt = mnv.RenderOptions.cli_defaults()
    cams = [cases.cfg3_camera(mnv, pose, W, H, fx=2800.0) for pose in (5, 11)]
    refs = [orc.render(ot, c.c, opt, want_rgba8=True) for c in cams]
    assert refs[0]["counters"].rays == W * H and refs[0]["counters"].rays_hit > 0.4 * W * H

    # the whole 4K frame on the tuned kernel: float and RGBA8 equal the oracle's
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    out8 = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
    mnv.render_voxels_accel(tree.accel, cams[0], opt, rgba=out, rgba8=out8)
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(refs[0]["rgba"]))
    assert np.array_equal(out8.cpu().numpy(), refs[0]["rgba8"])

    # the eight ranks' launches, one after the other: two frames per launch into each rank's compact buffer, the buffers placed
    # in the table the RCCL gather fills, un-permuted by the root's kernel
    part = TilePartition(W, H, world, tw, th, period)
    n_macro = part.n_macro
    counts = [mnv.partition_local_tiles((0, 0, W, H), r, world, tw, th, period) for r in range(world)]
    assert sum(counts) == n_macro and counts[0] < counts[1] and max(counts) == part.j_max       # rank 0 renders less (root relief)
    table = torch.zeros((world, len(cams), part.j_max, th, tw, 4), dtype=torch.uint8, device="cuda")
    for r in range(world):
        mnv.render_voxels_accel_batch(tree.accel, cams, opt, part=part.part(r), rgba8=table[r])
    frames = torch.zeros((len(cams), H, W, 4), dtype=torch.uint8, device="cuda")
    mnv.assemble_tiles(table, frames, W, H, world, tw, th, n_frames=len(cams), root_period=period)
    torch.cuda.synchronize()
    got = frames.cpu().numpy()
    for f in range(len(cams)):
        assert np.array_equal(got[f], refs[f]["rgba8"]), f
    # size-independent properties of the full frame
    a = refs[0]["rgba"][..., 3]
    assert a.min() >= 0.0 and a.max() <= 1.0 and np.isfinite(refs[0]["rgba"]).all()


def test_bench_workload_cfg3_cfg4_full_tree_against_the_oracle(mnv, orc, torch_gpu):
    """The tree bench.py times for BASELINE.json configs[2] / configs[3] (cases.CFG3_FULL: 7.2 M chunks, depth 11, inside the 5-10 M
    of the config) -- not the 2.7 M-chunk stand-in of the test above: one pose at 1920x1080 and one at 3840x2160 on the tuned kernel,
    float and RGBA8, bit for bit against the oracle; the oracle's integer counters of those poses equal the committed ones
    (tests/golden/cfg3_counters.json: the numerator of the cfg3 / cfg4_n1 rooflines), and the reference-layout kernel agrees at 1080p."""
    import json
    import os

    torch = torch_gpu
    tree = cases.make_tree(mnv, cases.CFG3_FULL)
    assert 5_000_000 < tree.capacity < 10_000_000
    ot = orc.tree_from_view(tree.host_view())
    tree.move_to_device()
    opt = mnv.RenderOptions.cli_defaults()
    committed = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg3_counters.json")))
    assert committed["capacity"] == tree.capacity
    for section, (w, h), pose in (("cfg3", (1920, 1080), 3), ("cfg4", (3840, 2160), 9)):
        cam = cases.cfg3_camera(mnv, pose, w, h, fx=1400.0 * w / 1920)
        ref = orc.render(ot, cam.c, opt, want_rgba8=True)
        c = ref["counters"].as_dict()
        assert all(committed[section]["poses"][str(pose)][k] == x for k, x in c.items()), (section, pose)
        assert c["rays"] == w * h and c["rays_hit"] > 0.4 * w * h
        out = torch.full((h, w, 4), float("nan"), dtype=torch.float32, device="cuda")
        out8 = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        mnv.render_voxels_accel(tree.accel, cam, opt, rgba=out, rgba8=out8)
        torch.cuda.synchronize()
        assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(ref["rgba"])), section
        assert np.array_equal(out8.cpu().numpy(), ref["rgba8"]), section
        if section == "cfg3":
            out.fill_(float("nan"))
            mnv.render_voxels(tree.device_view(), cam, opt, rgba=out)
            torch.cuda.synchronize()
            assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(ref["rgba"]))


def _renderer_on_cfg2(mnv, extra_capacity, w=1920, h=1080, **opt_over):
    tree = cases.make_tree(mnv, cases.CFG2_TREE)
    v = tree.host_view()
    desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, dir_octaves=2, need_viewdir=False, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    params = mlp_cases.make_params(mnv, desc, seed=21)
    r = mnv.Renderer()
    r.resize(w, h)
    r.set(tree, v.capacity + extra_capacity)
    r.set_model(desc, params, make_grid(mnv))
    r.set_seed(7)
    for k, val in opt_over.items():
        setattr(r.options, k, val)
    return r, tree, desc, params


def _pose(mnv, r, pose, w=1920, h=1080):
    cam = cases.cfg2_camera(mnv, pose, w, h, 1600.0)
    m = cam.c2w
    r.set_camera(tuple(m[9:12]), tuple(m[6:9]), fx=1600.0)
    return cam


def test_cfg5_splitting_at_scale_keeps_a_valid_tree_and_the_accel_follows(mnv, torch_gpu):
    torch = torch_gpu
    # room to grow and capacity <= 3/4 max: camera changes do not trigger visit tracking, every frame runs on the tuned kernel
    r, tree, desc, params = _renderer_on_cfg2(mnv, 1_000_000, use_splitting=True, max_depth=12, split_batch_size=4096, samples_per_corner=8)
    cap0 = tree.capacity
    assert cap0 == 1_499_569
    caps, added = [], 0
    for f in range(6):
        _pose(mnv, r, f % 3)
        st = r.render()
        caps.append(st["capacity"])
        added += st["added"]
        assert st["used_accel"] == 1 and st["track_visit"] == 0 and st["pruned"] == 0
        assert st["split_candidates"] > 0 and 0 < st["added"] <= 4096
    assert caps == sorted(caps) and caps[-1] == cap0 + added and added >= 6 * 2048
    frame = r.download()
    assert np.isfinite(frame).all() and frame[..., 3].min() >= 0.0 and frame[..., 3].max() <= 1.0
    r.sync_tree()
    data, child, parent = tree.host_arrays()
    assert child.shape[0] == caps[-1]
    check_tree_links(child, parent, caps[-1])
    # the packed accel was patched six times (mnv_accel_refresh): it still describes the tree the reference-layout kernel reads
    cam = cases.cfg2_camera(mnv, 1)
    opt = mnv.RenderOptions()
    C.memmove(C.byref(opt), C.byref(r.options), C.sizeof(opt))
    a = torch.empty((1080, 1920, 4), dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    mnv.render_voxels(tree.device_view(), cam, opt, rgba=a)
    mnv.render_voxels_accel(tree.accel, cam, opt, rgba=b)
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_cfg5_guided_sampling_frame_at_scale_matches_tensor_ops(mnv, torch_gpu):
    """use_guided_sampling on the 1.5 M-chunk tree at 1920x1080 (about 9 M samples through the networks per frame): the frame
    VolumeRenderer produces equals sample march -> cumsum / boolean-mask compaction (cuda_renderer.cpp:116-121, by torch) ->
    networks -> CSR composite, bit for bit."""
    torch = torch_gpu
    max_g, dim = 64, 4
    r, tree, desc, params = _renderer_on_cfg2(mnv, 1_000_000, use_guided_sampling=True, max_guided_samples=max_g)
    cam = _pose(mnv, r, 2)
    st = r.render()
    frame = r.download()
    assert st["guided_samples"] > 5_000_000 and st["used_accel"] == 1 and st["fused"] == 1   # one kernel: march + networks + composite
    opt = mnv.RenderOptions()
    C.memmove(C.byref(opt), C.byref(r.options), C.sizeof(opt))
    dv = tree.device_view()
    # the renderer's camera went through Camera::_update(); rebuild the same block the way its first render() did
    cam = mnv.Camera(1920, 1080, 1600.0).set_pose(tuple(cam.c2w[9:12]), tuple(cam.c2w[6:9]))
    n_px = 1920 * 1080
    num = torch.zeros(n_px, dtype=torch.int16, device="cuda")
    guided = torch.zeros((n_px, max_g, dim), dtype=torch.float32, device="cuda")
    guided[:, :, 0] = -1
    clusters = torch.zeros((n_px, max_g), dtype=torch.int16, device="cuda")
    mnv.get_samples_from_voxels(dv, cam, opt, num, guided, clusters, make_grid(mnv))
    offsets = torch.cumsum(num, 0)
    flat = guided.view(-1, dim)
    mask = flat[:, 0] >= 0
    valid, valid_clusters = flat[mask], clusters.view(-1)[mask]
    total = valid.shape[0]
    assert total == st["guided_samples"] == int(offsets[-1])
    values = torch.zeros((total, tree.host_view().data_dim + 1), dtype=torch.float32, device="cuda")
    mnv.Mlp(desc, params).query(valid_clusters, valid[:, 1:].contiguous(), values)
    out = torch.empty((1080, 1920, 4), dtype=torch.float32, device="cuda")
    mnv.render_nerf_results(dv, cam, opt, values, valid[:, 0].contiguous(), offsets, rgba=out)
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(frame))
    a = frame[..., 3]
    assert np.isfinite(frame).all() and a.min() >= 0.0 and a.max() <= 1.0 + 1e-6


def test_depth11_tree_in_the_size_class_of_the_references_budget(mnv, orc, torch_gpu):
    """The reference reserves 20 M chunks by default (`-c`, src/opts.cpp:17-32).  The depth-11 version of the merged-octree
    stand-in has 12.7 M chunks (5.7 GB of voxel rows, 102 M voxels): beyond 2^24 voxels and 4 GiB of row bytes, where 32-bit index
    arithmetic and float-encoded tracker rows (rt_core.cuh:237-252, inexact from 2^24) start to matter.  Both kernels == oracle."""
    torch = torch_gpu
    spec = dict(cases.CFG3_TREE, depth=11)
    tree = cases.make_tree(mnv, spec)
    assert tree.capacity > 12_000_000
    cam = cases.cfg3_camera(mnv, 5, 960, 540, 700.0)
    opt = mnv.RenderOptions.cli_defaults()
    opt.basis_minmax[1] = 8
    ref = orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt)["rgba"]
    assert (ref[..., 3] > 0).sum() > 100_000
    tree.move_to_device()
    # the inline cell words name their chunk in 22 bits RELATIVE to the smallest chunk number of that depth: none of this tree's all-leaves chunks
    # one level below the grid is lost to the field, although their numbers run far beyond 2^22
    info = mnv.accel_info(tree.accel, coverage=True)
    assert info["brick_levels"] == 2 and info["inline_cells"] > 100_000 and info["inline_lost_to_chunk_field"] == 0 and info["record_chunks"] == info["nonleaf_cells"], info
    a = torch.empty((540, 960, 4), dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    mnv.render_voxels_accel(tree.accel, cam, opt, rgba=a)
    mnv.render_voxels(tree.device_view(), cam, opt, rgba=b)
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(a.cpu().numpy()), cases.bits(ref))
    assert np.array_equal(cases.bits(b.cpu().numpy()), cases.bits(ref))


def test_cfg5_refinement_and_guided_sampling_together_at_scale(mnv, torch_gpu):
    """BASELINE.json configs[4] as stated -- dynamic refinement AND guided sampling on -- on the 1.5 M-chunk tree at 1920x1080: every
    frame is ONE fused kernel (march + networks + composite + trackers) followed by the split step; the tree stays valid and the
    packed accel follows it."""
    torch = torch_gpu
    r, tree, desc, params = _renderer_on_cfg2(mnv, 1_000_000, use_splitting=True, use_guided_sampling=True, max_depth=12, split_batch_size=4096,
                                              samples_per_corner=8, max_guided_samples=32)
    cap0 = tree.capacity
    added = 0
    for f in range(4):
        _pose(mnv, r, f)
        st = r.render()
        added += st["added"]
        assert st["fused"] == 1 and st["used_accel"] == 1 and st["guided_samples"] > 5_000_000
        assert st["split_candidates"] > 0 and 0 < st["added"] <= 4096 and st["pruned"] == 0
    frame = r.download()
    assert np.isfinite(frame).all()
    r.sync_tree()
    data, child, parent = tree.host_arrays()
    assert child.shape[0] == cap0 + added
    check_tree_links(child, parent, cap0 + added)
    cam = cases.cfg2_camera(mnv, 2)
    opt = mnv.RenderOptions()
    C.memmove(C.byref(opt), C.byref(r.options), C.sizeof(opt))
    a = torch.empty((1080, 1920, 4), dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    mnv.render_voxels(tree.device_view(), cam, opt, rgba=a)
    mnv.render_voxels_accel(tree.accel, cam, opt, rgba=b)
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_cfg5_on_the_merged_octree_at_the_survey_parameters(mnv, orc, torch_gpu):
    """BASELINE.json configs[4] as SURVEY.md 8(d) defines it -- cfg3's tree (cases.CFG3_FULL, 7.2 M chunks, depth 11) + the tiny MLP,
    max_guided_samples 128, samples_per_corner 8 -- at 1920x1080: the fused guided frame equals the four kernels it replaces bit for bit (long
    oblique rays, a quota of 128: where the producers' rings fill), no spin-wait is abandoned, a sub-rectangle equals the crop of the frame, the
    same pose at a quarter of the size equals the oracle's CPU chain (rt_core.cuh:418-576 -> compaction -> network -> rt_core.cuh:334-416), and
    the renderer runs both switches on it: one fused kernel per frame, a valid tree, the packed accel follows."""
    torch = torch_gpu
    from test_guided_fused_gpu import four_step_frame, oracle_frames

    tree = cases.make_tree(mnv, cases.CFG3_FULL)
    v = tree.host_view()
    assert 5_000_000 < v.capacity < 10_000_000
    tree.move_to_device()
    opt = mnv.RenderOptions.cli_defaults()
    opt.basis_minmax[1] = 8
    opt.max_guided_samples = 128
    desc = mnv.mlp_desc(n_clusters=8, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    params = mlp_cases.make_params(mnv, desc, seed=4)
    mlp = mnv.Mlp(desc, params)
    grid = cases.cfg3_cluster_grid(mnv)
    W, H = 1920, 1080
    cam = cases.cfg3_camera(mnv, 5, W, H, fx=1400.0)
    ref, ref8, total = four_step_frame(mnv, torch, tree, cam, opt, mlp, grid, 128, 4)
    assert total > 5_000_000
    for version in (2, 1):   # producer / consumer wavefronts (the default), then the one-role kernel
        mnv.accel_set_fused_kernel(tree.accel, version)
        out = torch.full((H, W, 4), float("nan"), dtype=torch.float32, device="cuda")
        counter = torch.zeros(1, dtype=torch.int64, device="cuda")
        mnv.render_guided_fused(tree.accel, cam, opt, mlp, grid, rgba=out, sample_counter=counter)
        torch.cuda.synchronize()
        assert int(counter.item()) == total, version
        assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(ref)), version
    mnv.accel_set_fused_kernel(tree.accel, -1)
    assert mnv.accel_fused_faults(tree.accel) == 0
    # a sub-rectangle that is not aligned to the 8x8 ray tiles
    x0, y0, w, h = 611, 203, 333, 251
    sub = torch.full((h, w, 4), float("nan"), dtype=torch.float32, device="cuda")
    mnv.render_guided_fused(tree.accel, cam, opt, mlp, grid, tile=(x0, y0, w, h), rgba=sub)
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(sub.cpu().numpy()), cases.bits(np.ascontiguousarray(ref[y0:y0 + h, x0:x0 + w])))
    # the oracle's CPU chain on the same pose at 480x270 (1.4 M-sample class): march, sample order, delta z, clusters, composite bit for bit
    small = cases.cfg3_camera(mnv, 5, 480, 270, fx=350.0)
    all_cpu, hybrid, n_small = oracle_frames(mnv, orc, torch, tree, small, opt, mlp, desc, params, grid, 128, 4)
    out = torch.full((270, 480, 4), float("nan"), dtype=torch.float32, device="cuda")
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    mnv.render_guided_fused(tree.accel, small, opt, mlp, grid, rgba=out, sample_counter=counter)
    torch.cuda.synchronize()
    assert int(counter.item()) == n_small and n_small > 300_000
    assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(hybrid))
    assert float(np.abs(out.cpu().numpy() - all_cpu).max()) < 2e-2
    assert mnv.accel_fused_faults(tree.accel) == 0
    del mlp
    # both switches through the host renderer at the survey's parameters
    cap0 = v.capacity
    r = mnv.Renderer()
    r.resize(W, H)
    r.set(tree, cap0 + 200_000)
    r.set_model(desc, params, grid)
    r.set_seed(7)
    o = r.options
    o.use_splitting, o.use_guided_sampling, o.max_depth, o.split_batch_size, o.samples_per_corner, o.max_guided_samples = True, True, 13, 4096, 8, 128
    added = 0
    for f in range(3):
        c = cases.cfg3_camera(mnv, 3 + f, W, H, fx=1400.0)
        m = c.c2w
        r.set_camera(tuple(m[9:12]), tuple(m[6:9]), up=(1.0, 0.0, 0.0), fx=1400.0)
        st = r.render()
        added += st["added"]
        assert st["fused"] == 1 and st["used_accel"] == 1 and st["guided_samples"] > 5_000_000 and st["pruned"] == 0
        assert 0 < st["added"] <= 4096
    frame = r.download()
    assert np.isfinite(frame).all()
    r.sync_tree()
    data, child, parent = tree.host_arrays()
    assert child.shape[0] == cap0 + added
    check_tree_links(child, parent, cap0 + added)
    ropt = mnv.RenderOptions()
    C.memmove(C.byref(ropt), C.byref(r.options), C.sizeof(ropt))
    a = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    mnv.render_voxels(tree.device_view(), cam, ropt, rgba=a)
    mnv.render_voxels_accel(tree.accel, cam, ropt, rgba=b)
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def _cfg5_rank(rank, world, lib, idq, resq):
    """One rank of test_cfg5_on_several_ranks_at_scale (the ranks share cuda:0; the transport is tests/shim/fake_rccl.cpp)."""
    import hashlib
    import os
    import sys
    import traceback

    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        if world > 1:
            os.environ["MNV_RCCL_LIBRARY"] = lib
            os.environ["MNV_LIB_PATH"] = os.path.join(root, "mega-nerf-viewer_amd", "testhooks", "libmnv.so")  # the build that honours it (tests/hooks.py)
        for p in (root, os.path.join(root, "tests"), os.path.join(root, "oracle")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch

        import mega_nerf_viewer_amd as mnv
        import test_scale_gpu as me

        torch.cuda.set_device(0)
        r, tree, desc, params = me._renderer_on_cfg2(mnv, 1_000_000, use_splitting=True, use_guided_sampling=True, max_depth=12, split_batch_size=4096,
                                                     samples_per_corner=8, max_guided_samples=32)
        if world > 1:
            if rank == 0:
                uid = mnv.comm_get_unique_id()
                for _ in range(world - 1):
                    idq.put(uid)
            else:
                uid = idq.get(timeout=300)
            r.set_ranks(mnv.Comm(uid, world, rank), 64, 24)
        log, digest = [], hashlib.sha256()
        for f in range(3):
            me._pose(mnv, r, f)
            st = r.render()
            log.append((st["split_candidates"], st["added"], st["capacity"], st["fused"]))
            if rank == 0:
                digest.update(r.download().tobytes())
        r.sync_tree()
        for a in tree.host_arrays():
            digest.update(np.ascontiguousarray(a).tobytes())
        if world > 1:
            r.set_ranks(None)
        resq.put(("ok", rank, log, digest.hexdigest()))
    except Exception:  # noqa: BLE001
        resq.put(("error", rank, traceback.format_exc(), ""))


@pytest.mark.parametrize("world", [2, 8])
def test_cfg5_on_several_ranks_at_scale(mnv, torch_gpu, fake_rccl, world):
    """BASELINE.json configs[4] -- refinement and guided sampling both on -- at its stated size on several ranks (processes sharing this
    GPU over the transport stand-in): 1.5 M-chunk tree, 1920x1080, three frames of 4096 splits each through Renderer.set_ranks.  Rank 0's
    frames and refined tree hash to what one rank produces; the other ranks (their frames stay partial) take the same decisions."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")

    def run(world):
        idq, resq = ctx.Queue(), ctx.Queue()
        procs = [ctx.Process(target=_cfg5_rank, args=(r, world, fake_rccl, idq, resq)) for r in range(world)]
        for p in procs:
            p.start()
        out = {}
        try:
            for _ in range(world):
                kind, rank, log, dig = resq.get(timeout=900)
                assert kind == "ok", log
                out[rank] = (log, dig)
        finally:
            for p in procs:
                p.join(timeout=120)
                if p.is_alive():
                    p.kill()   # exactly the processes started above
        return out

    one, many = run(1), run(world)
    assert all(c > 0 and 0 < a <= 4096 and fused == 1 for c, a, _, fused in one[0][0])
    assert all(many[r][0] == one[0][0] for r in range(world))
    assert many[0][1] == one[0][1]


def test_8k_frame_and_odd_sizes_beyond_the_configs(mnv, orc, torch_gpu):
    """Maximum sizes: one 7680x4320 frame (33 M rays, 518 k tiles: four times configs[3]'s) of the cfg2 tree -- too large for the CPU
    oracle in a test, so the two kernels (different layouts, different traversals) check each other bit for bit, 64 x 64 windows of
    the frame are checked against the oracle, and a tile of it equals the same rectangle rendered alone; then frame sizes that are no multiple of
    anything (1 x 1, 1 x 977, 8191 x 3) against the oracle."""
    torch = torch_gpu
    tree = cases.make_tree(mnv, cases.CFG2_TREE)
    ot = orc.tree_from_view(tree.host_view())
    tree.move_to_device()
    opt = mnv.RenderOptions.cli_defaults()
    W, H = 7680, 4320
    cam = cases.cfg2_camera(mnv, 3, W, H, 6400.0)
    a = torch.full((H, W, 4), float("nan"), dtype=torch.float32, device="cuda")
    b = torch.full((H, W, 4), float("nan"), dtype=torch.float32, device="cuda")
    mnv.render_voxels_accel(tree.accel, cam, opt, rgba=a)
    mnv.render_voxels(tree.device_view(), cam, opt, rgba=b)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(a).any()) and torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert 0.3 < float((a[..., 3] > 0).float().mean()) < 0.9          # the shell fills a good part of the frame
    for x0, y0 in ((0, 0), (W - 64, H - 64), (W // 2 - 32, H // 2 - 32), (2441, 1013), (5207, 3301)):
        tile = (x0, y0, 64, 64)
        ref = orc.render(ot, cam.c, opt, tile=tile)
        assert np.array_equal(cases.bits(a[y0:y0 + 64, x0:x0 + 64].cpu().numpy()), cases.bits(ref["rgba"])), tile
        t = torch.empty((64, 64, 4), dtype=torch.float32, device="cuda")
        mnv.render_voxels_accel(tree.accel, cam, opt, tile=tile, rgba=t)
        torch.cuda.synchronize()
        assert torch.equal(t.view(torch.int32), a[y0:y0 + 64, x0:x0 + 64].contiguous().view(torch.int32)), tile
    del a, b
    for w, h in ((1, 1), (1, 977), (8191, 3)):
        cam = cases.cfg2_camera(mnv, 7, w, h, 900.0)
        ref = orc.render(ot, cam.c, opt, want_rgba8=True)
        for which in ("accel", "ref_layout"):
            out = torch.full((h, w, 4), float("nan"), dtype=torch.float32, device="cuda")
            out8 = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
            if which == "accel":
                mnv.render_voxels_accel(tree.accel, cam, opt, rgba=out, rgba8=out8)
            else:
                mnv.render_voxels(tree.device_view(), cam, opt, rgba=out, rgba8=out8)
            torch.cuda.synchronize()
            assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(ref["rgba"])), (w, h, which)
            assert np.array_equal(out8.cpu().numpy(), ref["rgba8"]), (w, h, which)


def test_fog_workload_full_size_against_the_oracle(mnv, orc, torch_gpu):
    """bench.py's second distribution for the headline kernel (cases.FOG_TREE: 3.8 M chunks, 33 dense samples in 44 steps per ray -- long
    dense runs through inline cell words, sigma read with the colour row): two poses at 1920x1080 on the tuned kernel, float and RGBA8, bit
    for bit against the oracle; the oracle's counters of those poses equal the committed ones (tests/golden/fog_counters.json: the
    numerator of the fog roofline); the reference-layout kernel agrees; and a frame with a per-pixel depth limit and an image under it
    (the reference's offscreen == false call shape) as well."""
    import json
    import os

    torch = torch_gpu
    tree = cases.make_tree(mnv, cases.FOG_TREE)
    assert 3_000_000 < tree.capacity < 5_000_000
    ot = orc.tree_from_view(tree.host_view())
    tree.move_to_device()
    assert mnv.accel_info(tree.accel)["brick_levels"] >= 1
    opt = mnv.RenderOptions.cli_defaults()
    committed = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fog_counters.json")))
    assert committed["capacity"] == tree.capacity and len(committed["poses"]) == 16
    w, h = 1920, 1080
    for pose in (3, 12):
        cam = cases.cfg2_camera(mnv, pose)
        ref = orc.render(ot, cam.c, opt, want_rgba8=True)
        c = ref["counters"].as_dict()
        assert all(committed["poses"][str(pose)][k] == x for k, x in c.items()), pose
        assert c["hits"] > 25 * c["rays"] and c["steps"] > 35 * c["rays"]
        out = torch.full((h, w, 4), float("nan"), dtype=torch.float32, device="cuda")
        out8 = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        mnv.render_voxels_accel(tree.accel, cam, opt, rgba=out, rgba8=out8)
        torch.cuda.synchronize()
        assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(ref["rgba"])), pose
        assert np.array_equal(out8.cpu().numpy(), ref["rgba8"])
        if pose == 3:
            mnv.render_voxels(tree.device_view(), cam, opt, rgba=out)
            torch.cuda.synchronize()
            assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(ref["rgba"]))
            rng = np.random.default_rng(5)
            tmax = (2.6 * rng.uniform(0.8, 1.2, size=(h, w))).astype(np.float32)
            image = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
            want = orc.render(ot, cam.c, opt, tmax_px=tmax, rgba8_init=image)["rgba"]
            mnv.render_voxels_accel(tree.accel, cam, opt, rgba=out, tmax_px=torch.from_numpy(tmax).cuda(), rgba8_init=torch.from_numpy(image).cuda())
            torch.cuda.synchronize()
            assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(want))

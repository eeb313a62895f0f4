"""Guided-sampling kernel pair (BASELINE config 5) against the oracle:
bit-exact samples, cluster ids, counts, trackers and composited pixels."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _grid(mnv):
    g = mnv.ClusterGrid()
    g.grid_dim[0], g.grid_dim[1] = 3, 2
    for i, (lo, rng) in enumerate([(-1.0, 2.0), (-1.1, 2.2), (-0.9, 1.8)]):
        g.min_position[i], g.range[i] = lo, rng
    return g


@pytest.mark.parametrize("need_viewdir,embedding", [(False, -1), (True, -1), (False, 5), (True, 7)])
def test_get_samples_matches_oracle(mnv, orc, torch_gpu, need_viewdir, embedding):
    torch = torch_gpu
    spec = cases.CASES["sh4_d6"]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.need_viewdir, opt.appearance_embedding, opt.max_guided_samples = need_viewdir, embedding, 12
    opt.max_depth, opt.max_sample_count = 5, 9
    opt.rot_dirs[0], opt.rot_dirs[1] = 0.2, -0.1
    dim = 4 + (3 if need_viewdir else 0) + (1 if embedding != -1 else 0)
    v = tree.host_view()
    sc = np.full((v.capacity, 8), 8, np.int16)
    sc[::4] = 11
    visited_ref = np.zeros(v.capacity, np.int32)
    ref = orc.get_samples(orc.tree_from_view(v, sample_counts=sc), cam.c, opt, _grid(mnv), dim, visited=visited_ref, track_visit=True)
    tree.move_to_device(need_sample_counts=True)
    dv = tree.device_view()
    sc_dev = torch.from_numpy(sc).cuda()
    dv.sample_counts = sc_dev.data_ptr()
    n = cam.width * cam.height
    num = torch.zeros(n, dtype=torch.int16, device="cuda")
    samples = torch.full((n, 12, dim), -1.0, dtype=torch.float32, device="cuda")
    clusters = torch.full((n, 12), -1, dtype=torch.int16, device="cuda")
    split = torch.full((n, 3), -1.0, dtype=torch.float32, device="cuda")
    sample = torch.full((n, 3), -1.0, dtype=torch.float32, device="cuda")
    visited = torch.zeros(v.capacity, dtype=torch.int32, device="cuda")
    mnv.get_samples_from_voxels(dv, cam, opt, num, samples, clusters, _grid(mnv), split_track=split, sample_track=sample,
                                visited=visited, track_visit=True)
    torch.cuda.synchronize()
    assert np.array_equal(num.cpu().numpy(), ref["num_samples"]) and ref["num_samples"].max() == 12 and ref["num_samples"].min() == 0
    assert np.array_equal(cases.bits(samples.cpu().numpy()), cases.bits(ref["samples"]))
    assert np.array_equal(clusters.cpu().numpy(), ref["cluster_indices"])
    assert np.array_equal(split.cpu().numpy(), ref["split"]) and np.array_equal(sample.cpu().numpy(), ref["sample"])
    assert np.array_equal(visited.cpu().numpy(), visited_ref)


@pytest.mark.parametrize("case,depth_mode", [("sh4_d6", False), ("sh9_d7_aniso", False), ("rgba_d5", False), ("cfg1_sh1_d4", False), ("sh4_d6", True)])
def test_render_nerf_results_matches_oracle(mnv, orc, torch_gpu, case, depth_mode):
    torch = torch_gpu
    spec = cases.CASES[case]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.render_depth = depth_mode
    v = tree.host_view()
    n = cam.width * cam.height
    rng = np.random.default_rng(11)
    counts = rng.integers(0, 6, n).astype(np.int64)
    counts[:5] = [0, 1, 0, 2, 1]  # empty first ray, single-sample rays
    offsets = np.cumsum(counts)
    total = int(offsets[-1])
    stride = v.data_dim + 1  # nerf_result_buffer rows, cuda_renderer.cpp:489-491
    values = rng.normal(0, 1.0, (total, stride)).astype(np.float32)
    values[:, 3] = np.abs(values[:, 3]) * 20
    z = np.concatenate([np.sort(rng.uniform(0.5, 6.0, c)) for c in counts]).astype(np.float32) if total else np.zeros(0, np.float32)
    ref = orc.render_nerf_results(orc.tree_from_view(v), cam.c, opt, values, z, offsets, want_rgba8=True)
    rgba = torch.full((cam.height, cam.width, 4), float("nan"), dtype=torch.float32, device="cuda")
    rgba8 = torch.zeros((cam.height, cam.width, 4), dtype=torch.uint8, device="cuda")
    mnv.render_nerf_results(v, cam, opt, torch.from_numpy(values).cuda(), torch.from_numpy(z).cuda(), torch.from_numpy(offsets).cuda(),
                            rgba=rgba, rgba8=rgba8)
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(rgba.cpu().numpy()), cases.bits(ref["rgba"]))
    assert np.array_equal(rgba8.cpu().numpy(), ref["rgba8"])
    assert np.all(ref["rgba"][..., 3] == 1.0)  # out[3] = 1 before the composite (renderer_kernel.cu:316)


def test_guided_pair_round_trip_reproduces_voxel_render_structure(mnv, orc, torch_gpu):
    """Feeding the emitted samples' own voxel data back through render_nerf_results is not the voxel
    render (different quadrature), but the per-ray sample counts must equal the dense-step counts of the
    march up to max_guided_samples -- a size-independent consistency property of the pair."""
    spec = cases.CASES["shell_d7_sh9"]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.max_guided_samples = 128
    v = tree.host_view()
    ot = orc.tree_from_view(v)
    ref = orc.get_samples(ot, cam.c, opt, _grid(mnv), 4)
    march = orc.render(ot, cam.c, opt)
    assert int(ref["num_samples"].astype(np.int64).sum()) == march["counters"].hits
    z = ref["samples"][..., 0]
    for ray in np.flatnonzero(ref["num_samples"] > 1)[:200]:
        k = ref["num_samples"][ray]
        assert np.all(np.diff(z[ray, :k]) > 0)  # samples are emitted front to back


@pytest.mark.parametrize("case,need_viewdir,embedding", [("sh4_d6", False, -1), ("sh4_d6", True, 7), ("rgba_d5", True, -1), ("terrain_d7_aniso", False, 3),
                                                         ("sh25_d4", True, 2)])
def test_get_samples_on_the_packed_accel_matches_oracle(mnv, orc, torch_gpu, case, need_viewdir, embedding):
    """mnv_get_samples_from_voxels_accel (the tuned kernel's sample-emitting mode) against the oracle, every row format."""
    torch = torch_gpu
    spec = cases.CASES[case]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.need_viewdir, opt.appearance_embedding, opt.max_guided_samples = need_viewdir, embedding, 6
    opt.max_depth, opt.max_sample_count = 5, 9
    opt.rot_dirs[0], opt.rot_dirs[1] = 0.2, -0.1
    dim = 4 + (3 if need_viewdir else 0) + (1 if embedding != -1 else 0)
    v = tree.host_view()
    sc = np.random.default_rng(4).integers(0, 14, size=(v.capacity, 8)).astype(np.int16)
    ref = orc.get_samples(orc.tree_from_view(v, sample_counts=sc), cam.c, opt, _grid(mnv), dim)
    tree.move_to_device()
    sc_dev = torch.from_numpy(sc).cuda()
    n = cam.width * cam.height
    num = torch.zeros(n, dtype=torch.int16, device="cuda")
    samples = torch.full((n, 6, dim), -1.0, dtype=torch.float32, device="cuda")
    clusters = torch.full((n, 6), -1, dtype=torch.int16, device="cuda")
    split = torch.full((n, 3), -1.0, dtype=torch.float32, device="cuda")
    sample = torch.full((n, 3), -1.0, dtype=torch.float32, device="cuda")
    mnv.get_samples_from_voxels_accel(tree.accel, cam, opt, num, samples, clusters, _grid(mnv), split_track=split, sample_track=sample,
                                      sample_counts=sc_dev)
    torch.cuda.synchronize()
    assert np.array_equal(num.cpu().numpy(), ref["num_samples"]) and ref["num_samples"].max() >= 3
    k = np.arange(6)[None, :] < ref["num_samples"][:, None]  # emitted rows; the rest keep the caller's fill on both sides
    got_s, got_c = samples.cpu().numpy(), clusters.cpu().numpy()
    assert np.array_equal(cases.bits(got_s[k]), cases.bits(ref["samples"][k])) and np.all(got_s[~k] == -1.0)
    assert np.array_equal(got_c[k], ref["cluster_indices"][k])
    assert np.array_equal(split.cpu().numpy(), ref["split"]) and np.array_equal(sample.cpu().numpy(), ref["sample"])
    with pytest.raises(mnv.MnvError):
        mnv.get_samples_from_voxels_accel(tree.accel, cam, opt, num, samples[..., :dim - 1].contiguous(), clusters, _grid(mnv))

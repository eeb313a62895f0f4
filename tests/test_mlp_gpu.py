"""The build's own per-sample sub-module MLP (stands in for query_submodules,
cuda_renderer.cpp:165-203) on the matrix cores against its CPU restatement.  Parity with the reference is
unpinned by construction (its networks are TorchScript files outside the repository); what is checked here is
the HIP kernel against the build's own definition.  Tolerance: the encoded inputs, weights and activations are
bit-identical binary16 values on both sides; the only difference is the order of the fp32 accumulation inside
v_mfma_f32_16x16x32_f16 versus the sequential CPU sum, which can flip a binary16 rounding of a hidden
activation (relative 2^-11) -- bounded below by 4e-3 * (1 + |value|) on O(1) outputs."""
import numpy as np
import pytest

import mlp_cases

pytestmark = pytest.mark.gpu

CONFIGS = {
    "w64_l2_rgba": dict(n_clusters=3, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=5, center=(0.1, -0.2, 0.3), inv_extent=(0.5, 0.6, 0.7)),
    "w64_l1_min": dict(n_clusters=1, pos_octaves=0, hidden_width=64, hidden_layers=1, out_dim=1),
    "w64_l3_dir_emb_sh9": dict(n_clusters=8, pos_octaves=10, dir_octaves=4, need_viewdir=True, n_embeddings=12, embedding_dim=16,
                               hidden_width=64, hidden_layers=3, out_dim=29, inv_extent=(0.25, 0.25, 0.25)),
    "w128_l3_dir_sh16": dict(n_clusters=5, pos_octaves=12, dir_octaves=4, need_viewdir=True, hidden_width=128, hidden_layers=3, out_dim=50),
    "w128_l4_dir_sh9": dict(n_clusters=3, pos_octaves=10, dir_octaves=4, need_viewdir=True, hidden_width=128, hidden_layers=4, out_dim=29),
    "w128_l2_out128": dict(n_clusters=2, pos_octaves=2, hidden_width=128, hidden_layers=2, out_dim=128),
    # the K-slot layout of the first layer (every block starts at a multiple of 16 slots) at its corners: a direction block right behind three
    # position features; blocks of 16 octaves (99 features: seven half tiles, the last one masked); embeddings that push the first layer past
    # four K tiles (the kernels without a compile-time tile count); two and one K tiles of a 128-wide network
    "w64_l2_dir_only_coords": dict(n_clusters=2, pos_octaves=0, dir_octaves=0, need_viewdir=True, hidden_width=64, hidden_layers=2, out_dim=4),
    "w64_l2_oct16": dict(n_clusters=2, pos_octaves=16, dir_octaves=16, need_viewdir=True, hidden_width=64, hidden_layers=2, out_dim=7),
    "w128_l2_oct16_emb": dict(n_clusters=2, pos_octaves=16, dir_octaves=3, need_viewdir=True, n_embeddings=5, embedding_dim=33, hidden_width=128, hidden_layers=2,
                              out_dim=9),
    "w64_l2_emb64": dict(n_clusters=3, pos_octaves=10, dir_octaves=4, need_viewdir=True, n_embeddings=7, embedding_dim=64, hidden_width=64, hidden_layers=2, out_dim=5),
    "w128_l3_two_tiles": dict(n_clusters=2, pos_octaves=5, dir_octaves=1, need_viewdir=True, hidden_width=128, hidden_layers=3, out_dim=16),
    "w128_l1_one_tile": dict(n_clusters=2, pos_octaves=1, dir_octaves=1, need_viewdir=True, hidden_width=128, hidden_layers=1, out_dim=3),
}


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_mlp_matches_cpu_restatement(mnv, orc, torch_gpu, name):
    torch = torch_gpu
    desc = mnv.mlp_desc(**CONFIGS[name])
    params = mlp_cases.make_params(mnv, desc, seed=3)
    n = 5000 if desc.hidden_width == 64 else 3000
    x, cluster = mlp_cases.make_samples(desc, n, seed=4)
    mlp = mnv.Mlp(desc, params)
    pad = 2  # extra leading / trailing columns: strides larger than the widths
    d_x = torch.zeros((n, x.shape[1] + pad), dtype=torch.float32, device="cuda")
    d_x[:, :x.shape[1]] = torch.from_numpy(x).cuda()
    d_out = torch.full((n, desc.out_dim + pad), 7.0, dtype=torch.float32, device="cuda")
    mlp.query(torch.from_numpy(cluster).cuda(), d_x, d_out)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    want = orc.mlp_forward(desc, params, cluster, x)
    assert np.all(got[:, desc.out_dim:] == 7.0)  # columns beyond out_dim untouched
    got = got[:, :desc.out_dim]
    valid = (cluster >= 0) & (cluster < desc.n_clusters)
    assert np.all(got[~valid] == 0.0) and np.all(want[~valid] == 0.0)
    err = np.abs(got - want) / (1.0 + np.abs(want))
    assert np.isfinite(got).all() and err.max() < 4e-3, f"{name}: max rel err {err.max():.3e}"
    assert np.abs(want[valid]).mean() > 0.05  # the comparison is not vacuous
    # a second call on the same handle (scratch reuse) with a different n gives the same rows
    m = n // 3
    d_out2 = torch.zeros((m, desc.out_dim), dtype=torch.float32, device="cuda")
    mlp.query(torch.from_numpy(cluster[:m]).cuda(), d_x[:m], d_out2)
    torch.cuda.synchronize()
    assert np.array_equal(d_out2.cpu().numpy(), got[:m])


def test_mlp_rejects_bad_descriptions(mnv, torch_gpu):
    big = dict(pos_octaves=12, dir_octaves=4, need_viewdir=True, hidden_width=128, hidden_layers=5, out_dim=50)  # 194 KB of weights
    for bad in (dict(hidden_width=96), dict(out_dim=65), dict(n_clusters=0), dict(hidden_layers=0), dict(n_embeddings=4, embedding_dim=0)):
        desc = mnv.mlp_desc(**bad)
        assert mnv.Mlp.param_count(desc) == 0
        with pytest.raises(mnv.MnvError):
            mnv.Mlp(desc, np.zeros(16, np.float16))
    desc = mnv.mlp_desc()
    with pytest.raises(mnv.MnvError):
        mnv.Mlp(desc, np.zeros(mnv.Mlp.param_count(desc) + 1, np.float16))  # wrong blob size
    desc = mnv.mlp_desc(**big)
    with pytest.raises(mnv.MnvError, match="LDS"):
        mnv.Mlp(desc, np.zeros(mnv.Mlp.param_count(desc), np.float16))  # does not fit the CU's LDS


@pytest.mark.parametrize("width", [64, 128])
def test_mlp_ragged_batches_and_many_clusters(mnv, orc, torch_gpu, width):
    """Batch sizes around the kernel's units (one row; one short of / one past a 512-row tile and 8192 rows (the round-5 kernel's workgroup); a sort chunk of 2048
    rows) on a network with 1024 sub-modules -- most of them with no row at all, the rest with a handful: every workgroup is a partial pass."""
    torch = torch_gpu
    desc = mnv.mlp_desc(n_clusters=1024, pos_octaves=3, dir_octaves=1, need_viewdir=True, hidden_width=width, hidden_layers=2, out_dim=4)
    params = mlp_cases.make_params(mnv, desc, seed=8)
    mlp = mnv.Mlp(desc, params)
    for n in (1, 2, 63, 511, 513, 2047, 2049, 8191, 8193):
        x, cluster = mlp_cases.make_samples(desc, n, seed=100 + n)
        if n >= 2047:
            cluster[: n // 2] = 7  # one sub-module with a few thousand rows among a thousand with one or none
        d_out = torch.full((n, desc.out_dim), 7.0, dtype=torch.float32, device="cuda")
        mlp.query(torch.from_numpy(cluster).cuda(), torch.from_numpy(x).cuda(), d_out)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        want = orc.mlp_forward(desc, params, cluster, x)
        err = np.abs(got - want) / (1.0 + np.abs(want))
        assert np.isfinite(got).all() and err.max() < 4e-3, f"n = {n}: max rel err {err.max():.3e}"
        valid = (cluster >= 0) & (cluster < desc.n_clusters)
        assert np.all(got[~valid] == 0.0)


@pytest.mark.parametrize("width", [64, 128])
def test_mlp_tile_ranges_of_the_persistent_workgroups(mnv, orc, torch_gpu, width):
    """mlp_forward_kernel runs as many workgroups as the device holds and gives each a contiguous range of tiles (a tile = one pass of rows of
    one sub-module): batches of several tiles per workgroup whose ranges cross sub-module boundaries -- one sub-module with most of the rows,
    one with none in the middle, one with a single row -- against the CPU restatement; and a row's result must not depend on the batch it came
    in (which tile, which workgroup, which neighbours): slices of the batch evaluated on their own give the same bits."""
    torch = torch_gpu
    desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, dir_octaves=2, need_viewdir=True, hidden_width=width, hidden_layers=2, out_dim=7)
    params = mlp_cases.make_params(mnv, desc, seed=21)
    mlp = mnv.Mlp(desc, params)
    n = 400_000  # 782 tiles of 512 rows / 1563 of 256: three per workgroup on 256 compute units
    x, cluster = mlp_cases.make_samples(desc, n, seed=22)
    rng = np.random.default_rng(23)
    cluster[:] = rng.choice(np.array([0, 1, 3, 4, 5], np.int16), size=n, p=[0.1, 0.05, 0.7, 0.1, 0.05])  # sub-module 2 has no row
    cluster[cluster == 5] = 4
    cluster[n // 2] = 5  # ... and sub-module 5 exactly one
    cluster[7] = -3  # an invalid index: the row's result is zero
    d_x, d_c = torch.from_numpy(x).cuda(), torch.from_numpy(cluster).cuda()
    d_out = torch.full((n, desc.out_dim), 7.0, dtype=torch.float32, device="cuda")
    mlp.query(d_c, d_x, d_out)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    check = np.concatenate([np.arange(0, 6000), np.arange(n // 2 - 3000, n // 2 + 3000), np.arange(n - 6000, n)])  # the CPU side takes a sample of the rows
    want = orc.mlp_forward(desc, params, cluster[check], x[check])
    err = np.abs(got[check] - want) / (1.0 + np.abs(want))
    assert np.isfinite(got).all() and err.max() < 4e-3, f"max rel err {err.max():.3e}"
    assert np.all(got[7] == 0.0)
    for a, b in ((0, 1), (0, 70_001), (123_457, 300_000), (n - 513, n)):
        part = torch.zeros((b - a, desc.out_dim), dtype=torch.float32, device="cuda")
        mlp.query(d_c[a:b], d_x[a:b], part)
        torch.cuda.synchronize()
        assert np.array_equal(part.cpu().numpy(), got[a:b]), f"rows {a}:{b} depend on the batch they are evaluated in"

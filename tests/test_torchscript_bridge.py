"""the reference's networks are TorchScript containers it loads with torch::jit::load and evaluates per cluster under fp16
autocast (src/renderer/cuda_renderer.cpp:518-543, 165-203); what is INSIDE a container is not part of the reference.  tools/torchscript_container.py
puts the build's own network family into exactly that protocol (a container the reference could load) and converts such a container back into the
build's parameter blob / .npz.  Here: the container carries every attribute load_model reads; the way back is lossless; the reference's evaluation
procedure (query_submodules restated with torch ops) applied to the container agrees with the oracle's restatement of the family (CPU, fp32) and --
under autocast on the GPU -- with mnv_query_submodules; and the renderer renders the same guided frame from the exported .npz as from the blob.
Tolerance: binary16 weights and activations on both sides; what differs is the order of the fp32 sums inside a layer (a flipped binary16
rounding of a hidden activation, relative 2^-11) and, under autocast, the binary16 rounding of the OUTPUT layer that torch adds: 4e-3 (1 + |v|)."""
import os
import sys

import numpy as np
import pytest

import mlp_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

CONFIGS = {
    "w64_l2_sh9": dict(n_clusters=6, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=29),
    "w64_l3_dir_emb": dict(n_clusters=4, pos_octaves=6, dir_octaves=2, need_viewdir=True, n_embeddings=5, embedding_dim=8, hidden_width=64, hidden_layers=3, out_dim=5,
                           center=(0.1, -0.2, 0.3), inv_extent=(0.5, 0.6, 0.7)),
    "w128_l4_dir": dict(n_clusters=3, pos_octaves=10, dir_octaves=4, need_viewdir=True, hidden_width=128, hidden_layers=4, out_dim=29),
}


def _grid(mnv):
    g = mnv.ClusterGrid()
    g.grid_dim[0], g.grid_dim[1] = 3, 2
    for i, (lo, rng) in enumerate([(-1.0, 2.0), (-1.1, 2.2), (-0.9, 1.8)]):
        g.min_position[i], g.range[i] = lo, rng
    return g


def _saved_and_loaded(tsc, torch, mnv, desc, params, tmp_path):
    c = tsc.build_container(desc, params, _grid(mnv))
    path = str(tmp_path / "model.pt")
    torch.jit.save(c, path)
    return torch.jit.load(path, map_location="cpu")


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_container_protocol_round_trip_and_the_oracle(mnv, orc, tmp_path, name):
    import torch

    import torchscript_container as tsc

    desc = mnv.mlp_desc(**CONFIGS[name])
    params = mlp_cases.make_params(mnv, desc, seed=5)
    c = _saved_and_loaded(tsc, torch, mnv, desc, params, tmp_path)
    # what load_model reads (cuda_renderer.cpp:524-539)
    assert tuple(c.grid_dim.tolist()) == (3, 2) and c.min_position.shape == (3,) and c.max_position.shape == (3,)
    assert c.centroids.shape == (desc.n_clusters, 3)
    assert isinstance(c.need_viewdir, bool) and c.need_viewdir == bool(desc.need_viewdir)
    assert isinstance(c.need_appearance_embedding, bool) and c.need_appearance_embedding == (desc.n_embeddings > 0)
    subs = [getattr(c, f"sub_module_{i}") for i in range(desc.n_clusters)]
    x, cluster = mlp_cases.make_samples(desc, 700, seed=6)
    y = subs[0].forward(torch.from_numpy(x), False)             # nerfs[i].forward({input, false}), :190-191
    assert tuple(y.shape) == (700, desc.out_dim) and y.dtype == torch.float32
    # the way back is lossless
    fields, blob, grid = tsc.container_to_mnv(c)
    assert np.array_equal(blob, np.ascontiguousarray(params).view(np.uint16))
    for k in ("n_clusters", "pos_octaves", "dir_octaves", "need_viewdir", "n_embeddings", "embedding_dim", "hidden_width", "hidden_layers", "out_dim"):
        assert fields[k] == getattr(desc, k), k
    assert np.allclose(fields["center"], list(desc.center)) and np.allclose(fields["inv_extent"], list(desc.inv_extent))
    assert grid["grid_dim"] == [3, 2] and np.allclose(grid["max_position"], [1.0, 1.1, 0.9])
    npz = str(tmp_path / "model.npz")
    tsc.export_npz(c, npz)
    z = np.load(npz)
    assert z["mlp_desc"].tolist() == [desc.n_clusters, desc.pos_octaves, desc.dir_octaves, desc.need_viewdir, desc.n_embeddings, desc.embedding_dim,
                                      desc.hidden_width, desc.hidden_layers, desc.out_dim]
    assert np.array_equal(z["mlp_params"].view(np.uint16), blob)
    # the reference's evaluation procedure on the container (fp32: autocast is a CUDA matter) against the oracle's restatement of the family
    got = tsc.reference_query_submodules(c, torch.from_numpy(cluster.astype(np.int64)), torch.from_numpy(x), desc.out_dim, nerf_batch_size=128, autocast=False).numpy()
    want = orc.mlp_forward(desc, params, cluster, x)
    valid = (cluster >= 0) & (cluster < desc.n_clusters)
    err = np.abs(got - want) / (1.0 + np.abs(want))
    assert np.isfinite(got).all() and err[valid].max() < 4e-3, float(err[valid].max())
    assert np.abs(want[valid]).mean() > 0.05 and np.all(got[~valid] == 0.0)
    # a container of another architecture is refused, not misread
    class Other(torch.nn.Module):
        def forward(self, x: torch.Tensor, flag: bool) -> torch.Tensor:
            return x

    holder = torch.nn.Module()
    holder.register_buffer("centroids", torch.zeros((1, 3)))
    holder.add_module("sub_module_0", Other())
    with pytest.raises(ValueError):
        tsc.container_to_mnv(holder)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_query_submodules_against_the_references_procedure_under_autocast(mnv, torch_gpu, tmp_path, name):
    """mnv_query_submodules (matrix cores) == query_submodules (cuda_renderer.cpp:165-203: per-cluster batches through the TorchScript
    sub-modules under fp16 autocast) on a container of the build's family."""
    torch = torch_gpu
    import torchscript_container as tsc

    desc = mnv.mlp_desc(**CONFIGS[name])
    params = mlp_cases.make_params(mnv, desc, seed=5)
    c = _saved_and_loaded(tsc, torch, mnv, desc, params, tmp_path).to("cuda")
    n = 20000
    x, cluster = mlp_cases.make_samples(desc, n, seed=7, invalid_frac=0.0)
    d_x, d_c = torch.from_numpy(x).cuda(), torch.from_numpy(cluster).cuda()
    want = tsc.reference_query_submodules(c, d_c, d_x, desc.out_dim, nerf_batch_size=4096, batch_mult=1, autocast=True).cpu().numpy()
    got = torch.zeros((n, desc.out_dim), dtype=torch.float32, device="cuda")
    mnv.Mlp(desc, params).query(d_c, d_x, got)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    err = np.abs(got - want) / (1.0 + np.abs(want))
    print(f"{name}: max rel err vs the TorchScript container under autocast {err.max():.3e}, mean {err.mean():.3e}")
    assert np.isfinite(got).all() and err.max() < 4e-3 and err.mean() < 2e-4, (float(err.max()), float(err.mean()))
    assert np.abs(want).mean() > 0.05


@pytest.mark.gpu
def test_renderer_loads_the_exported_container(mnv, torch_gpu, tmp_path):
    """container -> export_npz -> mnv_renderer_load_model: the guided-sampling frame equals the frame of the same parameters handed over as a blob."""
    torch = torch_gpu
    import cases
    import torchscript_container as tsc

    spec = cases.CASES["sh9_d7_aniso"]
    frames = []
    for how in ("blob", "container"):
        tree = cases.make_tree(mnv, spec["tree"])
        v = tree.host_view()
        desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
        params = mlp_cases.make_params(mnv, desc, seed=21)
        r = mnv.Renderer()
        cam = spec["camera"]
        r.resize(cam["width"], cam["height"])
        r.set(tree, v.capacity)
        if how == "blob":
            r.set_model(desc, params, _grid(mnv))
        else:
            c = tsc.build_container(desc, params, _grid(mnv))
            pt, npz = str(tmp_path / "m.pt"), str(tmp_path / "m.npz")
            torch.jit.save(c, pt)
            tsc.export_npz(torch.jit.load(pt, map_location="cpu"), npz)
            r.load_model(npz)
        r.set_camera(cam.get("center", (-3.55, 0.0, 3.55)), cam.get("back", (-0.7071068, 0.0, 0.7071068)), fx=cam["fx"])
        r.options.use_guided_sampling, r.options.max_guided_samples = True, 32
        st = r.render()
        assert st["guided_samples"] > 1000
        frames.append(r.download())
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))

"""TorchScript model containers in the protocol the reference's load_model reads (src/renderer/cuda_renderer.cpp:518-543), for the build's own
network family -- and the way back into the build's parameter blob.

The reference loads `--model_path` with torch::jit::load and reads from the container
    grid_dim [2] int, min_position [3], max_position [3], centroids [n, 3]      tensors          (:524-527, :529)
    sub_module_<i>                                                              modules, one per centroid, called as forward({x, false})  (:530-533, :190-191)
    need_viewdir, need_appearance_embedding                                     bools            (:536-539)
and evaluates the sub-modules per cluster under fp16 autocast (query_submodules, :165-203).  The architecture INSIDE the sub-modules is whatever
cmusatyalab/mega-nerf scripted; it is not part of the reference repository, so this build defines its own family
(csrc/mnv_mlp.h: triangle-wave position / direction encoding, binary16 weights and activations, ReLU, fp32 accumulation).  This file is the bridge:

    build_container(desc, params, grid)     the family as a scripted container with exactly those attributes: the reference could load it
    container_to_mnv(container)             -> (desc fields, params blob, grid): what mnv_mlp_create / VolumeRenderer::load_model take
    export_npz(container, path)             the .npz `mnv_render --model_path` / mnv_renderer_load_model read (host/volume_renderer.cpp)
    reference_query_submodules(...)         query_submodules restated with torch ops (sort by cluster, batches, autocast, scatter): the checker

tests/test_torchscript_bridge.py holds mnv_query_submodules and the oracle's orc_mlp_forward against that procedure.
usage: python tools/torchscript_container.py export model.pt model.npz        (a container of THIS family; others are refused)"""
import sys
from typing import List

import numpy as np
import torch


def _tri(t: torch.Tensor) -> torch.Tensor:
    return 4.0 * torch.abs(t - torch.floor(t + 0.5)) - 1.0          # tri_wave, csrc/mnv_mlp.h


def _round16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float16).to(torch.float32)                     # activations of this family are binary16 values


class SubModule(torch.nn.Module):
    """One sub-module of the build's family.  forward(x, flag): x [n, 3 (+3 view direction) (+1 embedding index)] -- the columns the reference
    hands over (`valid_samples.slice(1, 1)`: the sample rows without z) -> [n, out_dim]."""

    def __init__(self, pos_octaves: int, dir_octaves: int, need_viewdir: bool, n_embeddings: int, embedding_dim: int, center, inv_extent,
                 weights: List[np.ndarray], biases: List[np.ndarray], embeddings):
        super().__init__()
        self.pos_octaves, self.dir_octaves, self.need_viewdir = int(pos_octaves), int(dir_octaves), bool(need_viewdir)
        self.n_embeddings, self.embedding_dim = int(n_embeddings), int(embedding_dim)
        self.register_buffer("center", torch.tensor(list(center), dtype=torch.float32))
        self.register_buffer("inv_extent", torch.tensor(list(inv_extent), dtype=torch.float32))
        layers = []
        for w, b in zip(weights, biases):
            lin = torch.nn.Linear(w.shape[1], w.shape[0])
            with torch.no_grad():
                lin.weight.copy_(torch.from_numpy(w.astype(np.float32)))
                lin.bias.copy_(torch.from_numpy(b.astype(np.float32)))
            layers.append(lin)
        self.layers = torch.nn.ModuleList(layers)
        emb = np.zeros((1, 1), np.float32) if embeddings is None else embeddings.astype(np.float32)
        self.register_buffer("embedding_table", torch.from_numpy(emb))

    def encode(self, v: torch.Tensor, octaves: int) -> torch.Tensor:
        feats = [v]
        s = 1.0
        for _ in range(octaves):
            feats.append(_tri(v * s))
            feats.append(_tri(v * s + 0.25))
            s = s * 2.0
        return torch.cat(feats, dim=1)

    def forward(self, x: torch.Tensor, flag: bool) -> torch.Tensor:
        x = x.to(torch.float32)
        p = (x[:, 0:3] - self.center) * self.inv_extent
        feats = [self.encode(p, self.pos_octaves)]
        col = 3
        if self.need_viewdir:
            feats.append(self.encode(x[:, 3:6], self.dir_octaves))
            col = 6
        if self.n_embeddings > 0:
            idx = torch.clamp(x[:, col].to(torch.int64), 0, self.n_embeddings - 1)
            feats.append(self.embedding_table[idx])
        h = _round16(torch.cat(feats, dim=1))
        n_layers = len(self.layers)
        i = 0
        for layer in self.layers:
            h = layer(h)
            if i + 1 < n_layers:
                h = _round16(torch.relu(h.to(torch.float32)))
            i += 1
        return h.to(torch.float32)


class Container(torch.nn.Module):
    def __init__(self, subs: List[SubModule], grid_dim, min_position, max_position, centroids, need_viewdir: bool, need_appearance_embedding: bool):
        super().__init__()
        self.need_viewdir: bool = bool(need_viewdir)
        self.need_appearance_embedding: bool = bool(need_appearance_embedding)
        self.register_buffer("grid_dim", torch.tensor(list(grid_dim), dtype=torch.int32))
        self.register_buffer("min_position", torch.tensor(list(min_position), dtype=torch.float32))
        self.register_buffer("max_position", torch.tensor(list(max_position), dtype=torch.float32))
        self.register_buffer("centroids", torch.tensor(np.asarray(centroids, np.float32)))
        for i, m in enumerate(subs):
            self.add_module(f"sub_module_{i}", m)

    def forward(self, x: torch.Tensor) -> torch.Tensor:   # (the reference never calls the container itself)
        return x


def _split_params(desc, params):
    """The blob of include/mnv.h (cluster-major; per cluster: W0 [W][in], b0, hidden layers, Wout [out][W], bout, embeddings) -> per-cluster arrays."""
    n_pos = 3 + 6 * desc.pos_octaves
    n_dir = (3 + 6 * desc.dir_octaves) if desc.need_viewdir else 0
    emb = desc.embedding_dim if desc.n_embeddings > 0 else 0
    in_dim, w = n_pos + n_dir + emb, desc.hidden_width
    p = np.ascontiguousarray(params).view(np.float16).reshape(-1)
    at, out = 0, []
    for _ in range(desc.n_clusters):
        dims = [(w, in_dim)] + [(w, w)] * (desc.hidden_layers - 1) + [(desc.out_dim, w)]
        ws, bs = [], []
        for o, i in dims:
            ws.append(p[at:at + o * i].reshape(o, i))
            at += o * i
            bs.append(p[at:at + o])
            at += o
        table = None
        if emb:
            table = p[at:at + desc.n_embeddings * emb].reshape(desc.n_embeddings, emb)
            at += desc.n_embeddings * emb
        out.append((ws, bs, table))
    assert at == p.size, "the parameter blob does not match the description"
    return out


def build_container(desc, params, grid, centroids=None) -> torch.jit.ScriptModule:
    """desc: mnv.MlpDesc; params: the blob; grid: mnv.ClusterGrid.  -> scripted container with the attributes of cuda_renderer.cpp:518-543."""
    subs = [SubModule(desc.pos_octaves, desc.dir_octaves, bool(desc.need_viewdir), desc.n_embeddings, desc.embedding_dim, list(desc.center), list(desc.inv_extent),
                      ws, bs, table) for ws, bs, table in _split_params(desc, params)]
    gd = [int(grid.grid_dim[0]), int(grid.grid_dim[1])]
    lo = [float(v) for v in grid.min_position]
    hi = [lo[i] + float(grid.range[i]) for i in range(3)]
    if centroids is None:   # cell centres of the cluster grid over world y, z (rt_core.cuh:541-549), x at the middle of the range
        centroids = [[(lo[0] + hi[0]) / 2, lo[1] + (a + 0.5) * (hi[1] - lo[1]) / gd[0], lo[2] + (b + 0.5) * (hi[2] - lo[2]) / gd[1]]
                     for a in range(gd[0]) for b in range(gd[1])][:desc.n_clusters]
        while len(centroids) < desc.n_clusters:
            centroids.append(centroids[-1])
    c = Container(subs, gd, lo, hi, centroids, bool(desc.need_viewdir), desc.n_embeddings > 0)
    c.eval()
    return torch.jit.script(c)


def container_to_mnv(container):
    """A container of THIS family -> (dict of mnv_mlp_desc fields, params blob uint16, dict(grid_dim, min_position, max_position)).
    Raises ValueError for any other architecture: the sub-modules must carry this family's attributes and Linear layers."""
    n = int(container.centroids.shape[0])
    fields, blobs = None, []
    for i in range(n):
        m = getattr(container, f"sub_module_{i}")
        try:
            f = dict(pos_octaves=int(m.pos_octaves), dir_octaves=int(m.dir_octaves), need_viewdir=int(bool(m.need_viewdir)), n_embeddings=int(m.n_embeddings),
                     embedding_dim=int(m.embedding_dim), center=[float(v) for v in m.center], inv_extent=[float(v) for v in m.inv_extent])
            layers = [getattr(m.layers, str(k)) for k in range(len(list(m.layers.children())))]
        except AttributeError as e:
            raise ValueError(f"sub_module_{i} is not of the build's network family ({e}): its architecture cannot be mapped onto mnv_mlp_desc") from None
        f.update(hidden_width=int(layers[0].weight.shape[0]), hidden_layers=len(layers) - 1, out_dim=int(layers[-1].weight.shape[0]))
        if fields is None:
            fields = f
        elif f != fields:
            raise ValueError("the sub-modules differ in shape: mnv_mlp_desc describes one shape for all clusters")
        for lin in layers:
            w, b = lin.weight.detach().cpu().numpy(), lin.bias.detach().cpu().numpy()
            if not (np.array_equal(w.astype(np.float16).astype(np.float32), w) and np.array_equal(b.astype(np.float16).astype(np.float32), b)):
                raise ValueError("weights are not binary16 values: quantise the model before exporting it (the family's weights are binary16)")
            blobs.append(w.astype(np.float16).reshape(-1))
            blobs.append(b.astype(np.float16))
        if fields["n_embeddings"] > 0:
            blobs.append(m.embedding_table.detach().cpu().numpy().astype(np.float16).reshape(-1))
    fields["n_clusters"] = n
    grid = dict(grid_dim=[int(v) for v in container.grid_dim], min_position=[float(v) for v in container.min_position],
                max_position=[float(v) for v in container.max_position])
    return fields, np.concatenate(blobs).view(np.uint16), grid


def export_npz(container, path):
    """The container as the .npz mnv_renderer_load_model / `mnv_render --model_path` read (host/volume_renderer.cpp: load_model)."""
    f, params, grid = container_to_mnv(container)
    np.savez(path, mlp_desc=np.array([f["n_clusters"], f["pos_octaves"], f["dir_octaves"], f["need_viewdir"], f["n_embeddings"], f["embedding_dim"],
                                      f["hidden_width"], f["hidden_layers"], f["out_dim"]], np.int32),
             mlp_center=np.array(f["center"], np.float32), mlp_inv_extent=np.array(f["inv_extent"], np.float32), mlp_params=params.view(np.float16),
             grid_dim=np.array(grid["grid_dim"], np.int64), min_position=np.array(grid["min_position"], np.float32),
             max_position=np.array(grid["max_position"], np.float32))


def reference_query_submodules(container, cluster_indices: torch.Tensor, samples: torch.Tensor, out_dim: int, nerf_batch_size: int = 4096,
                               batch_mult: int = 1, autocast: bool = True) -> torch.Tensor:
    """VolumeRenderer::Impl::query_submodules (cuda_renderer.cpp:165-203) with torch ops, statement for statement: sort the samples by cluster,
    per cluster and batch of nerf_batch_size * batch_mult rows gather the rows, run nerfs[cluster].forward({input, false}) under autocast, scatter
    the results back.  (Rows whose cluster names no sub-module stay zero here; the reference would index out of range.)"""
    result = torch.zeros((samples.shape[0], out_dim), dtype=torch.float32, device=samples.device)
    n_modules = int(container.centroids.shape[0])
    nerfs = [getattr(container, f"sub_module_{i}") for i in range(n_modules)]
    sorted_idx, inverse = torch.sort(cluster_indices.to(torch.int64), stable=True)
    uniq, counts = torch.unique_consecutive(sorted_idx, return_counts=True)
    offset, batch = 0, nerf_batch_size * batch_mult
    with torch.inference_mode():
        for c, cnt in zip(uniq.tolist(), counts.tolist()):
            for chunk in range(0, cnt, batch):
                q = inverse[offset + chunk:min(offset + chunk + batch, offset + cnt)]
                if 0 <= c < n_modules:
                    x = samples.index_select(0, q)
                    if autocast:
                        with torch.autocast(device_type=samples.device.type, dtype=torch.float16):
                            r = nerfs[c].forward(x, False)
                    else:
                        r = nerfs[c].forward(x, False)
                    result.index_copy_(0, q, r.to(torch.float32))
            offset += cnt
    return result


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "export":
        export_npz(torch.jit.load(sys.argv[2], map_location="cpu"), sys.argv[3])
        print("wrote", sys.argv[3])
    else:
        print(__doc__)
        sys.exit(2)

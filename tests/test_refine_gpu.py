"""Refinement kernels (reference src/cuda/renderer_kernel.cu:63-213) against the oracle.
These three kernels sit in renderer_kernel.cu, which cannot be built here, so the oracle for them is a
restatement only (parity unpinned)."""
import numpy as np
import pytest

import cases
from guided_cases import cluster_grid

pytestmark = pytest.mark.gpu


def _setup(mnv, need_viewdir, embedding):
    tree = cases.make_tree(mnv, cases.CASES["sh9_d7_aniso"]["tree"])
    v = tree.host_view()
    _, child, parent = tree.host_arrays()
    opt = mnv.RenderOptions.cli_defaults()
    opt.samples_per_corner, opt.need_viewdir, opt.appearance_embedding = 5, need_viewdir, embedding
    dim = 3 + (3 if need_viewdir else 0) + (1 if embedding != -1 else 0)
    return tree, v, child.copy(), parent.copy(), opt, dim


@pytest.mark.parametrize("need_viewdir,embedding", [(False, -1), (True, 3)])
def test_add_children_and_generate_samples(mnv, orc, torch_gpu, need_viewdir, embedding):
    torch = torch_gpu
    tree, v, child, parent, opt, dim = _setup(mnv, need_viewdir, embedding)
    cap, n_new = v.capacity, 37
    max_cap = cap + n_new
    rng = np.random.default_rng(5)
    leaves = np.argwhere(child == 0)
    parent_nodes = leaves[rng.choice(len(leaves), n_new, replace=False)].astype(np.int32)
    child_big = np.zeros((max_cap, 8), np.int32)
    child_big[:cap] = child
    child_big[cap:] = 12345  # garbage the kernel must clear
    parent_big = np.full(max_cap, -7, np.int32)
    parent_big[:cap] = parent
    visited = np.zeros(max_cap, np.int32)
    visited[:cap:2] = 1
    samples = rng.uniform(0, 1, (n_new * 8, 5, dim)).astype(np.float32)
    clusters = np.full((n_new * 8, 5), -1, np.int16)
    grid = cluster_grid(mnv.ClusterGrid)
    d = {k: torch.from_numpy(a.copy()).cuda() for k, a in dict(child=child_big, parent=parent_big, visited=visited, samples=samples,
                                                               clusters=clusters, nodes=parent_nodes).items()}
    edit = mnv.tree_edit(d["child"], d["parent"], list(v.offset), list(v.scale), cap)
    mnv.add_children_and_generate_samples(edit, opt, d["nodes"], d["samples"], d["clusters"], d["visited"], grid)
    torch.cuda.synchronize()
    orc.add_children_and_generate_samples(child_big, parent_big, list(v.offset), list(v.scale), cap, opt, parent_nodes, samples, clusters, visited, grid)
    assert np.array_equal(d["child"].cpu().numpy(), child_big) and np.array_equal(d["parent"].cpu().numpy(), parent_big)
    assert np.array_equal(d["visited"].cpu().numpy(), visited)
    assert np.array_equal(cases.bits(d["samples"].cpu().numpy()), cases.bits(samples))
    assert np.array_equal(d["clusters"].cpu().numpy(), clusters)
    # the new voxels' sample points lie inside their parent voxel's world-space cube
    new_pts = samples[..., :3].reshape(n_new, 8 * 5, 3)
    assert np.all(np.ptp(new_pts, axis=1) < 2.0 / np.float32(list(v.scale)))

    # generate_samples for existing voxels (renderer_kernel.cu:200-213) on the grown tree
    nodes = np.argwhere(child_big == 0)[::17][:64].astype(np.int32)
    s2 = rng.uniform(0, 1, (len(nodes), 5, dim)).astype(np.float32)
    c2 = np.full((len(nodes), 5), -1, np.int16)
    ds, dc, dn = torch.from_numpy(s2.copy()).cuda(), torch.from_numpy(c2.copy()).cuda(), torch.from_numpy(nodes).cuda()
    edit = mnv.tree_edit(d["child"], d["parent"], list(v.offset), list(v.scale), max_cap)
    mnv.generate_samples(edit, opt, dn, ds, dc, grid)
    torch.cuda.synchronize()
    orc.generate_samples(parent_big, list(v.offset), list(v.scale), opt, nodes, s2, c2, grid)
    assert np.array_equal(cases.bits(ds.cpu().numpy()), cases.bits(s2)) and np.array_equal(dc.cpu().numpy(), c2)


def test_adjust_parents_and_children(mnv, orc, torch_gpu):
    torch = torch_gpu
    tree, v, child, parent, opt, dim = _setup(mnv, False, -1)
    cap = v.capacity
    rng = np.random.default_rng(9)
    # delete a set of leaf chunks (chunks without children), as prune_tree does (cuda_renderer.cpp:343-381)
    is_leaf_chunk = (child != 0).sum(axis=1) == 0
    cand = np.flatnonzero(is_leaf_chunk & (np.arange(cap) > 0))
    to_delete = np.zeros(cap, np.uint8)
    to_delete[rng.choice(cand, len(cand) // 3, replace=False)] = 1
    index_shifts = np.cumsum(to_delete).astype(np.int32)
    first = int(np.flatnonzero(to_delete)[0])
    dchild, dparent = torch.from_numpy(child.copy()).cuda(), torch.from_numpy(parent.copy()).cuda()
    edit = mnv.tree_edit(dchild, dparent, list(v.offset), list(v.scale), cap)
    mnv.adjust_parents_and_children(edit, first, torch.from_numpy(to_delete).cuda(), torch.from_numpy(index_shifts).cuda())
    torch.cuda.synchronize()
    orc.adjust_parents_and_children(child, parent, cap, first, to_delete, index_shifts)
    assert np.array_equal(dchild.cpu().numpy(), child) and np.array_equal(dparent.cpu().numpy(), parent)
    # after compaction (dropping deleted chunks) the links are consistent again
    keep = to_delete == 0
    c2, p2 = child[keep], parent[keep]
    tgt = (np.arange(len(c2))[:, None] + c2)[c2 != 0]
    src = np.argwhere(c2 != 0)
    assert np.array_equal(p2[tgt], src[:, 0] * 8 + src[:, 1])

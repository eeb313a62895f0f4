"""Refinement kernels (reference src/cuda/renderer_kernel.cu:63-213) against the oracle.
These three kernels sit in renderer_kernel.cu, which as a whole cannot be built for gfx950; the block that holds them is
cut out verbatim by oracle/Makefile.ref, and tests/golden/ref_refine_kernels.npz holds what the reference's own kernels
produced (test_refinement_kernels_match_reference_code_goldens)."""
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
    parent_nodes = np.ascontiguousarray(leaves[rng.choice(len(leaves), n_new, replace=False)], dtype=np.int32)
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
    nodes = np.ascontiguousarray(np.argwhere(child_big == 0)[::17][:64], dtype=np.int32)  # argwhere slices are not C-contiguous
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


# ------------------------------------------------------------------ tracker-consuming host logic (csrc/mnv_refine.hip)
import os  # noqa: E402

import refine_oracle as ro  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_SEL = np.load(os.path.join(GOLD, "refine_selection.npz"))
_SEL_NAMES = sorted({k.split("/")[0] for k in _SEL.files})


@pytest.mark.parametrize("name", _SEL_NAMES)
def test_selection_matches_torch_goldens(mnv, torch_gpu, name):
    """expand_voxels' vote / get_more_samples' selection (cuda_renderer.cpp:205-227,281-296) against the vectors
    produced by the reference's libtorch expressions."""
    torch = torch_gpu
    track, k = _SEL[f"{name}/track"], int(_SEL[f"{name}/k"])
    d_track = torch.from_numpy(track).cuda()
    nodes = torch.full((k, 2), -9, dtype=torch.int32, device="cuda")
    n_out, n_cand = mnv.select_split_candidates(d_track, k, nodes)
    want = _SEL[f"{name}/split_nodes"]
    assert n_cand == int(_SEL[f"{name}/split_n"]) and n_out == want.shape[0]
    assert np.array_equal(nodes.cpu().numpy()[:n_out], want) and np.all(nodes.cpu().numpy()[n_out:] == -9)
    nodes.fill_(-9)
    n_out, n_cand = mnv.select_sample_candidates(d_track, k, nodes)
    want = _SEL[f"{name}/sample_nodes"]
    assert n_cand == int(_SEL[f"{name}/sample_n"]) and n_out == want.shape[0]
    assert np.array_equal(nodes.cpu().numpy()[:n_out], want)


def test_selection_full_frame_against_oracle(mnv, torch_gpu):
    """A 1920x1080 tracker (2,073,600 rows) with heavy repetition, against the numpy restatement."""
    torch = torch_gpu
    rng = np.random.default_rng(11)
    n = 1920 * 1080
    chunk = (rng.zipf(1.3, n) % 1_400_000).astype(np.int64)
    track = np.stack([(1 + chunk % 10).astype(np.float32), chunk.astype(np.float32), (chunk * 5 % 8).astype(np.float32)], 1)
    track[rng.random(n) < 0.6] = (11.0, -1.0, -1.0)
    k = 100_000
    d_track = torch.from_numpy(track).cuda()
    nodes = torch.empty((k, 2), dtype=torch.int32, device="cuda")
    n_out, n_cand = mnv.select_split_candidates(d_track, k, nodes)
    want, want_n = ro.select_split_candidates(track, k)
    assert (n_out, n_cand) == (want.shape[0], want_n) and np.array_equal(nodes.cpu().numpy()[:n_out], want)
    n_out, n_cand = mnv.select_sample_candidates(d_track, k, nodes)
    want, want_n = ro.select_sample_candidates(track, k)
    assert (n_out, n_cand) == (want.shape[0], want_n) and np.array_equal(nodes.cpu().numpy()[:n_out], want)
    # max_out == 0 only counts
    assert mnv.select_split_candidates(d_track, 0, None) == (0, ro.select_split_candidates(track, 0)[1])


@pytest.mark.parametrize("what", ["large_chunk_index", "large_priority"])
def test_selection_rows_beyond_the_compact_key_take_the_wide_layout(mnv, torch_gpu, what):
    """The vote sorts 27 + 5 (sample selection: 27 + 9) bit keys -- what the march writes: voxel indices below 2^27, depths / sample counts as
    priorities -- and repeats with the 52-bit layout of the contract in include/mnv.h when a row does not fit: chunk indices above 2^24
    (trees beyond 16.7 M chunks) or priorities above 30 / 510 must give the numpy restatement's answer all the same."""
    torch = torch_gpu
    rng = np.random.default_rng(5)
    n = 300_000
    chunk = (rng.zipf(1.3, n) % 50_000).astype(np.int64)
    prio = 1 + chunk % 9
    if what == "large_chunk_index":
        chunk = chunk * 4 + (1 << 24)          # exactly representable in binary32 (multiples of 4 above 2^24)
    else:
        prio = prio + 600                       # beyond both compact priority ranges
    track = np.stack([prio.astype(np.float32), chunk.astype(np.float32), (chunk * 5 % 8).astype(np.float32)], 1)
    track[rng.random(n) < 0.5] = (-1.0, -1.0, -1.0)
    k = 20_000
    d_track = torch.from_numpy(track).cuda()
    nodes = torch.empty((k, 2), dtype=torch.int32, device="cuda")
    n_out, n_cand = mnv.select_split_candidates(d_track, k, nodes)
    want, want_n = ro.select_split_candidates(track, k)
    assert want_n > 1000 and (n_out, n_cand) == (want.shape[0], want_n) and np.array_equal(nodes.cpu().numpy()[:n_out], want)
    n_out, n_cand = mnv.select_sample_candidates(d_track, k, nodes)
    want, want_n = ro.select_sample_candidates(track, k)
    assert (n_out, n_cand) == (want.shape[0], want_n) and np.array_equal(nodes.cpu().numpy()[:n_out], want)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_selection_does_not_depend_on_the_order_of_the_tracker_rows(mnv, torch_gpu, seed):
    """What refinement on several ranks rests on (VolumeRenderer::set_ranks): every rank votes on the all-gathered tracker rows, which
    arrive rank by rank in compact tile order with -1 rows where a ragged partition left tiles out -- another order and another length
    than the one-GPU frame.  The vote counts candidates (sort, run-length encode, stable sort by count with ties in key order), so any
    permutation of the rows, with any number of empty rows added, selects the same nodes in the same order."""
    torch = torch_gpu
    rng = np.random.default_rng(100 + seed)
    n = 200_000
    chunk = (rng.zipf(1.25, n) % 30_000).astype(np.int64)
    track = np.stack([(1 + chunk % 7).astype(np.float32), chunk.astype(np.float32), (chunk * 3 % 8).astype(np.float32)], 1)
    track[rng.random(n) < 0.5] = (-1.0, -1.0, -1.0)
    k = 5000
    outs = []
    for variant in range(3):
        t = track if variant == 0 else track[rng.permutation(n)]
        if variant == 2:
            pad = np.full((12_345, 3), -1.0, np.float32)
            t = np.concatenate([t[: n // 3], pad, t[n // 3:]])
        d = torch.from_numpy(np.ascontiguousarray(t)).cuda()
        nodes = torch.full((k, 2), -9, dtype=torch.int32, device="cuda")
        res = [mnv.select_split_candidates(d, k, nodes), nodes.cpu().numpy().copy()]
        nodes.fill_(-9)
        res += [mnv.select_sample_candidates(d, k, nodes), nodes.cpu().numpy().copy()]
        outs.append(res)
    assert outs[0][0][0] > 100 and outs[0][2][0] > 100
    for o in outs[1:]:
        assert o[0] == outs[0][0] and o[2] == outs[0][2]
        assert np.array_equal(o[1], outs[0][1]) and np.array_equal(o[3], outs[0][3])


def test_apply_split_and_sample_results(mnv, torch_gpu):
    torch = torch_gpu
    z = np.load(os.path.join(GOLD, "refine_split_mean.npz"))
    results = z["results"]
    n_children, spc, dd = results.shape[0], 8, 28
    cap = 5
    rng = np.random.default_rng(3)
    data = (rng.standard_normal((cap + n_children // 8 + 2, 8, dd))).astype(np.float16)
    counts = rng.integers(0, 30, (data.shape[0], 8)).astype(np.int16)
    d_data, d_counts, d_res = torch.from_numpy(data.copy()).cuda(), torch.from_numpy(counts.copy()).cuda(), torch.from_numpy(results).cuda()
    mnv.apply_split_results(d_data, d_counts, cap, n_children // 8, d_res, spc, dd)
    torch.cuda.synchronize()
    ro.apply_split_results(data, counts, cap, results, spc)
    got = d_data.cpu().numpy()
    assert np.array_equal(got.view(np.uint16), data.view(np.uint16)) and np.array_equal(d_counts.cpu().numpy(), counts)
    # torch.mean(out=half) golden: within one binary16 step (reduction order unspecified there)
    g = got.reshape(-1, dd)[cap * 8: cap * 8 + n_children].view(np.uint16).astype(np.int32)
    w = z["rows"].astype(np.int32)
    mono = lambda u: np.where(u & 0x8000, -(u & 0x7fff), u)  # noqa: E731
    assert np.abs(mono(g) - mono(w)).max() <= 1

    # running average of existing voxels (cuda_renderer.cpp:307-332)
    nodes = np.stack([rng.permutation(data.shape[0])[:40], rng.integers(0, 8, 40)], 1).astype(np.int32)
    res2 = (rng.standard_normal((40, spc, dd + 1)) * 2).astype(np.float32)
    d_nodes, d_res2 = torch.from_numpy(nodes).cuda(), torch.from_numpy(res2).cuda()
    mnv.apply_sample_results(d_data, d_counts, d_nodes, d_res2, spc, dd)
    torch.cuda.synchronize()
    ro.apply_sample_results(data, counts, nodes, res2, spc)
    assert np.array_equal(d_data.cpu().numpy().view(np.uint16), data.view(np.uint16)) and np.array_equal(d_counts.cpu().numpy(), counts)


def test_prune_matches_torch_goldens(mnv, torch_gpu):
    torch = torch_gpu
    z = np.load(os.path.join(GOLD, "refine_prune.npz"))
    cap, max_cap = int(z["capacity"]), z["visited"].shape[0]
    pad = lambda a: np.concatenate([a, np.zeros((max_cap - cap,) + a.shape[1:], a.dtype)])  # noqa: E731
    d = {k: torch.from_numpy(pad(z[k]) if k != "visited" else z[k].copy()).cuda() for k in ("data", "child", "parent", "visited")}
    d["data"] = d["data"].view(torch.int16)
    edit = mnv.tree_edit(d["child"], d["parent"], [0.5] * 3, [0.5] * 3, cap)
    new_cap, n_del = mnv.prune_tree(edit, d["data"], z["data"].shape[2], None, d["visited"], max_cap)
    assert new_cap == int(z["new_capacity"]) and n_del == cap - new_cap
    assert np.array_equal(d["data"].cpu().numpy()[:new_cap].view(np.uint16), z["out_data"])
    assert np.array_equal(d["child"].cpu().numpy()[:new_cap], z["out_child"])
    assert np.array_equal(d["parent"].cpu().numpy()[:new_cap], z["out_parent"])
    v = d["visited"].cpu().numpy()
    assert v[0] == 1 and not v[1:].any()
    # nothing to prune: marks are cleared, arrays untouched
    d["visited"][:new_cap] = 1
    before = d["child"].clone()
    edit = mnv.tree_edit(d["child"], d["parent"], [0.5] * 3, [0.5] * 3, new_cap)
    assert mnv.prune_tree(edit, d["data"], z["data"].shape[2], None, d["visited"], max_cap) == (new_cap, 0)
    assert torch.equal(before, d["child"]) and not d["visited"].cpu().numpy()[1:].any()
    # unmarked root is refused
    with pytest.raises(mnv.MnvError):
        mnv.prune_tree(edit, d["data"], z["data"].shape[2], None, d["visited"] * 0, max_cap)


def test_prune_large_tree_keeps_the_image(mnv, orc, torch_gpu):
    """Visit marks from a real track_visit frame on a 300k-chunk tree (several compaction segments): the HIP
    prune equals the oracle's, and the pruned tree renders the same frame bit for bit from that camera."""
    torch = torch_gpu
    tree = cases.make_tree(mnv, dict(kind="shell", depth=9, basis_dim=4, radius=0.35, half_thickness=1.5 / 512, seed=3))
    cam = mnv.orbit_camera(640, 360, 520.0, 2.6, 30.0, 20.0)
    opt = mnv.RenderOptions.cli_defaults()
    v = tree.host_view()
    cap, dd = v.capacity, v.data_dim
    assert cap > (1 << 18)
    data, child, parent = (a.copy() for a in tree.host_arrays())
    max_cap = cap + 100
    tree.move_to_device(max_capacity=max_cap, need_parent=True, need_sample_counts=True)
    dv = tree.device_view()
    h, w = cam.height, cam.width
    rgba = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    visited = torch.zeros(max_cap, dtype=torch.int32, device="cuda")
    mnv.render_voxels(dv, cam, opt, rgba=rgba, visited=visited, track_visit=True)
    torch.cuda.synchronize()
    visited_h = visited.cpu().numpy()
    counts = (np.arange(cap * 8) % 251).astype(np.int16).reshape(cap, 8)
    d_counts = torch.from_numpy(np.concatenate([counts, np.zeros((100, 8), np.int16)])).cuda()
    d_child = torch.from_numpy(np.concatenate([child, np.zeros((100, 8), np.int32)])).cuda()
    d_parent = torch.from_numpy(np.concatenate([parent, np.zeros(100, np.int32)])).cuda()
    d_data = torch.from_numpy(np.concatenate([data, np.zeros((100, 8, dd), np.uint16)]).view(np.int16)).cuda()
    edit = mnv.tree_edit(d_child, d_parent, list(v.offset), list(v.scale), cap)
    new_cap, n_del = mnv.prune_tree(edit, d_data, dd, d_counts, visited, max_cap)
    want_cap, want_del = ro.prune_tree(orc, child, parent, data, counts, visited_h, cap, max_cap)
    assert (new_cap, n_del) == (want_cap, want_del) and 0 < new_cap < cap
    assert np.array_equal(d_child.cpu().numpy()[:new_cap], child[:new_cap]) and np.array_equal(d_parent.cpu().numpy()[:new_cap], parent[:new_cap])
    assert np.array_equal(d_data.cpu().numpy()[:new_cap].view(np.uint16), data[:new_cap])
    assert np.array_equal(d_counts.cpu().numpy()[:new_cap], counts[:new_cap])
    # render the compacted arrays
    pv = mnv.TreeView()
    for f in ("offset", "scale", "N", "data_dim", "format", "basis_dim"):
        setattr(pv, f, getattr(dv, f))
    pv.data, pv.child, pv.parent, pv.capacity = d_data.data_ptr(), d_child.data_ptr(), d_parent.data_ptr(), new_cap
    rgba2 = torch.empty_like(rgba)
    mnv.render_voxels(pv, cam, opt, rgba=rgba2)
    torch.cuda.synchronize()
    assert torch.equal(rgba.view(torch.int32), rgba2.view(torch.int32))


def test_refinement_kernels_match_reference_code_goldens(mnv, orc, torch_gpu):
    """The three HIP refinement kernels against the outputs of the reference's own kernels (renderer_kernel.cu:63-213, cut out
    verbatim by oracle/Makefile.ref and run on gfx950; tests/golden/ref_refine_kernels.npz)."""
    torch = torch_gpu
    import refine_kernel_cases as rk
    z = np.load(os.path.join(GOLD, "ref_refine_kernels.npz"))
    g = rk.grid(mnv)
    for variant in rk.VARIANTS:
        tree, opt, dim, parent_nodes, visited, samples = rk.add_children_inputs(mnv, variant)
        v = tree.host_view()
        _, child, parent = tree.host_arrays()
        cap = tree.capacity
        child_big = np.zeros((cap + rk.N_NEW, 8), np.int32)
        child_big[:cap] = child
        parent_big = np.zeros(cap + rk.N_NEW, np.int32)
        parent_big[:cap] = parent
        d = {k: torch.from_numpy(a.copy()).cuda() for k, a in dict(child=child_big, parent=parent_big, visited=visited, samples=samples,
                                                                   clusters=np.full(samples.shape[:2], -1, np.int16), nodes=parent_nodes).items()}
        edit = mnv.tree_edit(d["child"], d["parent"], list(v.offset), list(v.scale), cap)
        mnv.add_children_and_generate_samples(edit, opt, d["nodes"], d["samples"], d["clusters"], d["visited"], g)
        torch.cuda.synchronize()
        pre = f"add_children/{variant}/"
        for k in ("child", "parent", "visited", "clusters"):
            assert np.array_equal(d[k].cpu().numpy(), z[pre + k]), k
        assert np.array_equal(cases.bits(d["samples"].cpu().numpy()), cases.bits(z[pre + "samples"]))

        tree, opt, dim, nodes, samples = rk.generate_samples_inputs(mnv, variant)
        v = tree.host_view()
        _, child, parent = tree.host_arrays()
        ds, dc, dn = torch.from_numpy(samples.copy()).cuda(), torch.full(samples.shape[:2], -1, dtype=torch.int16, device="cuda"), torch.from_numpy(nodes).cuda()
        dparent = torch.from_numpy(parent.copy()).cuda()
        dchild = torch.from_numpy(child.copy()).cuda()
        edit = mnv.tree_edit(dchild, dparent, list(v.offset), list(v.scale), tree.capacity)
        mnv.generate_samples(edit, opt, dn, ds, dc, g)
        torch.cuda.synchronize()
        pre = f"generate_samples/{variant}/"
        assert np.array_equal(cases.bits(ds.cpu().numpy()), cases.bits(z[pre + "samples"])) and np.array_equal(dc.cpu().numpy(), z[pre + "clusters"])
    tree, to_delete, shifts = rk.adjust_inputs(mnv, orc)
    v = tree.host_view()
    _, child, parent = tree.host_arrays()
    dchild, dparent = torch.from_numpy(child.copy()).cuda(), torch.from_numpy(parent.copy()).cuda()
    edit = mnv.tree_edit(dchild, dparent, list(v.offset), list(v.scale), tree.capacity)
    mnv.adjust_parents_and_children(edit, 1, torch.from_numpy(to_delete).cuda(), torch.from_numpy(shifts).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(dchild.cpu().numpy(), z["adjust_parents/child"])
    keep = to_delete == 0
    assert np.array_equal(dparent.cpu().numpy()[keep], z["adjust_parents/parent"][keep])


@pytest.mark.parametrize("tree_spec,cam_args", [
    (dict(kind="shell", depth=9, basis_dim=4, radius=0.35, half_thickness=1.5 / 512, seed=3), (640, 360, 520.0, 2.6, 30.0, 20.0)),
    (dict(kind="random", depth=6, basis_dim=9, refine_prob=0.55, empty_prob=0.6, sigma_max=40.0, coef_sd=1.0, seed=11), (320, 240, 260.0, 2.4, 200.0, 35.0)),
])
def test_accel_follows_a_prune_in_place(mnv, orc, torch_gpu, tree_spec, cam_args):
    """mnv_prune_tree_accel: the packed accel is patched while the tree is pruned (chunks renumbered in node words and lookup grids,
    sub-trees that went replaced by their parent leaf, colour rows compacted).  Afterwards the patched accel, a freshly built accel of
    the pruned tree and the reference-layout kernel give the same frames, tracker rows (voxel numbers!) and visit marks -- from the
    pruning camera and from others that look at the pruned parts."""
    torch = torch_gpu
    tree = cases.make_tree(mnv, tree_spec)
    v = tree.host_view()
    cap, dd = v.capacity, v.data_dim
    max_cap = cap + 64
    tree.move_to_device(max_capacity=max_cap, need_parent=True, need_sample_counts=True)
    dv = tree.device_view()
    cam = mnv.orbit_camera(*cam_args)
    opt = mnv.RenderOptions.cli_defaults()
    opt.basis_minmax[1] = max(v.basis_dim - 1, 0)
    opt.max_depth, opt.max_sample_count = 7, 9
    h, w = cam.height, cam.width
    visited = torch.zeros(max_cap, dtype=torch.int32, device="cuda")
    before = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    mnv.render_voxels_accel_visit(tree.accel, cam, opt, visited, dv.parent, rgba=before)   # the marks of this view
    torch.cuda.synchronize()
    edit = mnv.TreeEdit()
    edit.child, edit.parent, edit.N, edit.capacity = dv.child, dv.parent, 2, cap
    for i in range(3):
        edit.offset[i], edit.scale[i] = v.offset[i], v.scale[i]
    new_cap, n_del = mnv.prune_tree(edit, dv.data, dd, dv.sample_counts, visited, max_cap, accel=tree.accel)
    assert 0 < new_cap < cap and n_del == cap - new_cap
    pv = mnv.TreeView()
    for f in ("data", "child", "parent", "sample_counts", "offset", "scale", "N", "data_dim", "format", "basis_dim"):
        setattr(pv, f, getattr(dv, f))
    pv.capacity = new_cap
    fresh = mnv.accel_create(pv, max_capacity=max_cap)
    try:
        sc = torch.full((max_cap, 8), 8, dtype=torch.int16, device="cuda")
        sc[::3] = 12
        for pose, other in enumerate([cam, mnv.orbit_camera(cam_args[0], cam_args[1], cam_args[2], cam_args[3], cam_args[4] + 140.0, -cam_args[5])]):
            outs = []
            for which in ("patched", "fresh", "ref"):
                rgba = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
                split = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
                sample = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
                marks = torch.zeros(max_cap, dtype=torch.int32, device="cuda")
                if which == "ref":
                    pv.sample_counts = sc.data_ptr()
                    mnv.render_voxels(pv, other, opt, rgba=rgba, split_track=split, sample_track=sample, visited=marks, track_visit=True)
                else:
                    mnv.render_voxels_accel_visit(tree.accel if which == "patched" else fresh, other, opt, marks, pv.parent, rgba=rgba,
                                                  split_track=split, sample_track=sample, sample_counts=sc)
                torch.cuda.synchronize()
                outs.append((rgba, split, sample, marks))
            for k in range(4):
                assert torch.equal(outs[0][k].view(torch.int32), outs[2][k].view(torch.int32)), (pose, k, "patched vs reference layout")
                assert torch.equal(outs[1][k].view(torch.int32), outs[2][k].view(torch.int32)), (pose, k, "fresh vs reference layout")
            if pose == 0:
                assert torch.equal(outs[0][0].view(torch.int32), before.view(torch.int32))   # the pruning view is unchanged by the prune
    finally:
        mnv.accel_destroy(fresh)


def _vote_tracker(kind, rng):
    """Tracker frames that put the threshold count of the vote's selection in different places."""
    if kind == "heavy_tail":        # a frame's shape: most voxels once or twice, a few thousands of times
        n = 1920 * 1080
        chunk = (rng.zipf(1.3, n) % 1_400_000).astype(np.int64)
        empty = 0.6
    elif kind == "mostly_ties":     # nearly every voted voxel has exactly two votes: the rows AT the threshold fill the batch, in key order
        n = 600_000
        chunk = np.repeat(rng.permutation(1_000_000)[: n // 2], 2).astype(np.int64)
        chunk[:3000] = chunk[0]
        chunk = rng.permutation(chunk)
        empty = 0.1
    elif kind == "few_voted":       # fewer voted voxels than the batch holds
        n = 200_000
        chunk = rng.permutation(3_000_000)[:n].astype(np.int64)
        chunk[:900] = np.repeat(chunk[:300], 3)
        chunk[900:1000] = chunk[0]
        empty = 0.3
    else:                           # "popular": more voxels with >= 2047 votes than the batch holds -> the sort of all counts
        n = 40 * 2100 + 50_000
        chunk = np.concatenate([np.repeat(np.arange(40) * 977 + 5, 2100), rng.permutation(500_000)[:50_000]]).astype(np.int64)
        chunk = rng.permutation(chunk)
        empty = 0.0
    track = np.stack([(1 + chunk % 10).astype(np.float32), chunk.astype(np.float32), (chunk * 5 % 8).astype(np.float32)], 1)
    if empty:
        track[rng.random(n) < empty] = (11.0, -1.0, -1.0)
    return track


@pytest.mark.parametrize("kind,ks", [("heavy_tail", (1, 7, 4096, 4192, 8192)), ("mostly_ties", (64, 4096)), ("few_voted", (4096,)), ("popular", (8, 64))])
def test_selection_of_a_batch_without_sorting_all_counts(mnv, torch_gpu, kind, ks):
    """Batches up to 8192 rows (split_batch_size: 4096 on the command line, 4192 in render_options.hpp:49) are SELECTED: histogram of the
    counts -> the count of the last row that fits -> the runs above it sorted in one workgroup, the runs at it taken in key order
    (csrc/mnv_refine.hip, select_candidates_compact).  Same rows in the same order as the numpy restatement's full sort, wherever the
    threshold falls: in the tail, at two votes with tens of thousands of ties, nowhere (fewer voted voxels than the batch), or in the
    histogram's last bin (falls back to the sort of all counts)."""
    torch = torch_gpu
    track = _vote_tracker(kind, np.random.default_rng(31))
    d_track = torch.from_numpy(track).cuda()
    for k in ks:
        nodes = torch.full((k, 2), -9, dtype=torch.int32, device="cuda")
        n_out, n_cand = mnv.select_split_candidates(d_track, k, nodes)
        want, want_n = ro.select_split_candidates(track, k)
        assert (n_out, n_cand) == (want.shape[0], want_n), (kind, k)
        got = nodes.cpu().numpy()
        assert np.array_equal(got[:n_out], want) and np.all(got[n_out:] == -9), (kind, k)
        nodes.fill_(-9)
        n_out, n_cand = mnv.select_sample_candidates(d_track, k, nodes)
        want, want_n = ro.select_sample_candidates(track, k)
        assert (n_out, n_cand) == (want.shape[0], want_n) and np.array_equal(nodes.cpu().numpy()[:n_out], want), (kind, k)


def test_selection_paths_agree(mnv, torch_gpu, tmp_path):
    """MNV_VOTE_FULL_SORT=1 (test-hook build of the library, read once per process) forces the sort of all counts for small batches too: a child process votes on the same
    tracker with it and must write the same rows."""
    import subprocess
    import sys

    torch = torch_gpu
    track = _vote_tracker("heavy_tail", np.random.default_rng(32))
    np.save(tmp_path / "track.npy", track)
    k = 4096
    nodes = torch.full((k, 2), -9, dtype=torch.int32, device="cuda")
    n_out, n_cand = mnv.select_split_candidates(torch.from_numpy(track).cuda(), k, nodes)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"import sys; sys.path.insert(0, {root!r}); import numpy as np, torch, mega_nerf_viewer_amd as mnv\n"
            f"t = torch.from_numpy(np.load({str(tmp_path / 'track.npy')!r})).cuda(); nodes = torch.full(({k}, 2), -9, dtype=torch.int32, device='cuda')\n"
            f"r = mnv.select_split_candidates(t, {k}, nodes); np.save({str(tmp_path / 'nodes.npy')!r}, nodes.cpu().numpy()); print(r[0], r[1])\n")
    import hooks

    env = hooks.hooks_env(MNV_VOTE_FULL_SORT="1")  # the knob exists in the test-hook build only (csrc/mnv_knobs.h)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split()[-2:] == [str(n_out), str(n_cand)]
    assert np.array_equal(np.load(tmp_path / "nodes.npy"), nodes.cpu().numpy())


@pytest.mark.parametrize("seed", range(40))
def test_selection_random_trackers_against_the_restatement(mnv, torch_gpu, seed):
    """Random tracker frames -- row count, share of empty rows, how votes concentrate, priorities as a function of the voxel or not (rows of
    one voxel with different priorities are different candidates, cuda_renderer.cpp:208) -- and random batch sizes on both sides of the
    selection path's limits (8192 rows, 2047 votes): the numpy restatement's rows in its order, from either path."""
    torch = torch_gpu
    rng = np.random.default_rng(9000 + seed)
    n = int(10 ** rng.uniform(1.0, 5.8))
    n_vox = max(1, int(n * 10 ** rng.uniform(-3.0, 0.3)))
    kind = seed % 4
    if kind == 0:
        vox = rng.integers(0, n_vox, n)                                   # uniform: counts cluster around n / n_vox
    elif kind == 1:
        vox = rng.zipf(rng.uniform(1.1, 2.0), n) % n_vox                  # heavy tail
    elif kind == 2:
        vox = np.repeat(rng.integers(0, 1 << 24, (n + 1) // 2), 2)[:n]    # pairs: the threshold sits at two votes, everything ties
    else:
        vox = np.where(rng.random(n) < 0.5, rng.integers(0, 8, n), rng.integers(0, n_vox, n))  # a handful of voxels with half of the votes
    vox = rng.permutation(vox.astype(np.int64))
    prio = 1 + vox % 11 if seed % 3 else rng.integers(0, 12, n)
    track = np.stack([prio.astype(np.float32), (vox >> 3).astype(np.float32), (vox & 7).astype(np.float32)], 1)
    track[rng.random(n) < rng.uniform(0.0, 0.95)] = (13.0, -1.0, -1.0)
    d_track = torch.from_numpy(np.ascontiguousarray(track)).cuda()
    for k in sorted({1, int(rng.integers(2, 200)), int(rng.integers(200, 8193)), int(rng.integers(8193, 30000))}):
        nodes = torch.full((k, 2), -9, dtype=torch.int32, device="cuda")
        n_out, n_cand = mnv.select_split_candidates(d_track, k, nodes)
        want, want_n = ro.select_split_candidates(track, k)
        assert (n_out, n_cand) == (want.shape[0], want_n), (seed, k)
        got = nodes.cpu().numpy()
        assert np.array_equal(got[:n_out], want) and np.all(got[n_out:] == -9), (seed, k)
        nodes.fill_(-9)
        n_out, n_cand = mnv.select_sample_candidates(d_track, k, nodes)
        want, want_n = ro.select_sample_candidates(track, k)
        assert (n_out, n_cand) == (want.shape[0], want_n) and np.array_equal(nodes.cpu().numpy()[:n_out], want), (seed, k)

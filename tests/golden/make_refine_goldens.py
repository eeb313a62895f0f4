"""Golden vectors for the refinement host logic (tests/golden/refine_*.npz).

The reference implements expand_voxels / get_more_samples / prune_tree as libtorch tensor expressions
(src/renderer/cuda_renderer.cpp:205-381).  This script runs THOSE expressions, operator for operator, through
the same ATen operators from Python (torch CPU, this image's torch 2.10) on seeded inputs and stores inputs +
outputs.  Nothing of the reference's text is stored: the fixtures are arrays.

Run here (no GPU needed):   python tests/golden/make_refine_goldens.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402
import mnv_oracle as orc  # noqa: E402


def torch_split_selection(split_tracker, split_batch_size):
    """cuda_renderer.cpp:205-227."""
    split_candidates = split_tracker[split_tracker[:, 1] >= 0]
    to_split, to_split_counts = torch.unique(split_candidates, dim=0, sorted=False, return_inverse=False, return_counts=True)
    to_split_counts = to_split_counts.to(torch.int32).unsqueeze_(-1)
    to_split = torch.cat([-to_split_counts, to_split], -1)
    to_split = to_split[to_split[:, 0] < -1]
    to_split = torch.unique(to_split, dim=0)
    n = to_split.size(0)
    nodes = to_split[0:split_batch_size][:, 2:].to(torch.int32)
    return nodes.numpy(), n


def torch_sample_selection(sample_tracker, split_batch_size):
    """cuda_renderer.cpp:281-296."""
    sample_candidates = sample_tracker[sample_tracker[:, 1] >= 0]
    if sample_candidates.size(0) == 0:
        return np.zeros((0, 2), np.int32), 0
    sample_candidates = torch.unique(sample_candidates, dim=0)
    to_sample = sample_candidates[0:split_batch_size][:, 1:].to(torch.int32)
    return to_sample.numpy(), sample_candidates.size(0)


def random_tracker(rng, n, n_chunks, prio_lo, prio_hi, invalid_frac, chunk_base=0):
    t = np.full((n, 3), -1.0, np.float32)
    t[:, 0] = prio_hi + 1
    valid = rng.random(n) >= invalid_frac
    # a skewed distribution so that many rows repeat
    chunk = (rng.integers(0, n_chunks, n) * rng.integers(0, 2, n) + rng.integers(0, max(n_chunks // 50, 1), n)) % n_chunks + chunk_base
    t[valid, 0] = rng.integers(prio_lo, prio_hi + 1, n)[valid]
    t[valid, 1] = chunk[valid].astype(np.float32)
    t[valid, 2] = rng.integers(0, 8, n)[valid]
    # priority is a function of the voxel in real trackers (its depth / sample count): make most rows consistent
    consistent = rng.random(n) < 0.8
    t[valid & consistent, 0] = (prio_lo + (t[valid & consistent, 1].astype(np.int64) * 7 + t[valid & consistent, 2].astype(np.int64)) % (prio_hi - prio_lo + 1))
    return t


def selection_cases():
    rng = np.random.default_rng(20260101)
    out = {}
    out["random_small"] = (random_tracker(rng, 4000, 300, 1, 6, 0.3), 64)
    out["random_large"] = (random_tracker(rng, 120000, 40000, 1, 10, 0.2), 4096)
    out["no_votes"] = (np.stack([np.full(500, 3.0, np.float32), np.arange(500, dtype=np.float32), np.zeros(500, np.float32)], 1), 32)
    out["all_invalid"] = (np.tile(np.array([[11.0, -1.0, -1.0]], np.float32), (256, 1)), 32)
    # chunk ids beyond 2^24: the tracker's float rows cannot hold them exactly; rows hold the rounded floats
    big = random_tracker(rng, 6000, 2000, 1, 8, 0.25, chunk_base=19_000_000)
    out["chunks_above_2p24"] = (big, 128)
    # negative priorities: sample_counts is left uninitialised on the device by the reference loader (n3tree.cpp:235-241)
    neg = random_tracker(rng, 5000, 400, -40, 7, 0.3)
    out["negative_priority"] = (neg, 100)
    # trackers of a real march (CPU oracle), sh4_d6
    spec = cases.CASES["sh4_d6"]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.max_depth = 6
    opt.max_sample_count = 9
    v = tree.host_view()
    sc = np.random.default_rng(7).integers(0, 14, size=(v.capacity, 8)).astype(np.int16)
    ref = orc.render(orc.tree_from_view(v, sample_counts=sc), cam.c, opt, want_trackers=True)
    out["march_sh4_d6_split"] = (ref["split"].reshape(-1, 3), 256)
    out["march_sh4_d6_sample"] = (ref["sample"].reshape(-1, 3), 256)
    return out


def main():
    sel = {}
    for name, (track, k) in selection_cases().items():
        t = torch.from_numpy(track)
        s_nodes, s_n = torch_split_selection(t, k)
        a_nodes, a_n = torch_sample_selection(t, k)
        sel[f"{name}/track"] = track
        sel[f"{name}/k"] = np.int32(k)
        sel[f"{name}/split_nodes"] = s_nodes
        sel[f"{name}/split_n"] = np.int32(s_n)
        sel[f"{name}/sample_nodes"] = a_nodes
        sel[f"{name}/sample_n"] = np.int32(a_n)
        print(f"{name}: rows {track.shape[0]}  split {s_n} -> {s_nodes.shape[0]}  sample {a_n} -> {a_nodes.shape[0]}")
    np.savez_compressed(os.path.join(HERE, "refine_selection.npz"), **sel)

    # mean over samples into binary16 rows (cuda_renderer.cpp:262-266): torch.mean(out=half)
    rng = np.random.default_rng(5)
    results = (rng.standard_normal((48 * 8, 8, 29)) * 3).astype(np.float32)
    out = torch.empty((48 * 8, 28), dtype=torch.half)
    torch.mean(torch.from_numpy(results)[:, :, 0:28], 1, out=out)
    np.savez_compressed(os.path.join(HERE, "refine_split_mean.npz"), results=results, rows=out.numpy().view(np.uint16))

    # prune (cuda_renderer.cpp:335-381): visit marks from a real march, torch for to_delete / cumsum / argmin / row copies,
    # the oracle's serial restatement for adjust_parents_and_children_kernel (renderer_kernel.cu cannot be built here)
    spec = cases.CASES["sh4_d6"]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, dict(spec["camera"], width=40, height=32, fx=140.0))  # few rays: part of the tree stays unvisited
    opt = cases.make_options(mnv, spec["options"])
    v = tree.host_view()
    cap = v.capacity
    data, child, parent = (a.copy() for a in tree.host_arrays())
    visited = np.zeros(cap + 16, np.int32)
    orc.render(orc.tree_from_view(v), cam.c, opt, visited=visited, track_visit=True)
    fix = dict(data=data.copy(), child=child.copy(), parent=parent.copy(), visited=visited.copy(), capacity=np.int32(cap))
    tv = torch.from_numpy(visited)
    to_delete = tv[0:cap] == 0
    num_to_delete = int(to_delete.sum().item())
    index_shifts = torch.cumsum(to_delete, 0, dtype=torch.int32)
    first_shift_index = int(index_shifts.argmin().item())
    orc.adjust_parents_and_children(child, parent, cap, first_shift_index, to_delete.numpy().astype(np.uint8), index_shifts.numpy())
    to_delete_shifted = to_delete[first_shift_index:cap]
    copy_indices = torch.arange(first_shift_index, cap)[to_delete_shifted == False]  # noqa: E712
    td, tc, tp = torch.from_numpy(data.view(np.int16)), torch.from_numpy(child), torch.from_numpy(parent)
    n_keep = copy_indices.size(0)
    td[first_shift_index:first_shift_index + n_keep] = td[copy_indices].clone()
    tc[first_shift_index:first_shift_index + n_keep] = tc[copy_indices].clone()
    tp[first_shift_index:first_shift_index + n_keep] = tp[copy_indices].clone()
    new_cap = cap - num_to_delete
    print(f"prune: capacity {cap} -> {new_cap}, first_shift_index {first_shift_index}")
    fix.update(out_data=data[:new_cap], out_child=child[:new_cap], out_parent=parent[:new_cap], new_capacity=np.int32(new_cap),
               first_shift_index=np.int32(first_shift_index))
    np.savez_compressed(os.path.join(HERE, "refine_prune.npz"), **fix)


if __name__ == "__main__":
    main()

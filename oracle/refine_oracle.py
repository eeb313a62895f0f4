"""CPU restatement (numpy) of the reference's refinement host logic.

TEST INFRASTRUCTURE ONLY -- imported by tests/ as the checker, never by the product package.

Restates, op for op, the libtorch expressions of /root/reference/src/renderer/cuda_renderer.cpp:
  select_split_candidates   :205-227   expand_voxels' vote (unique_dim + counts, count >= 2, unique_dim again)
  select_sample_candidates  :281-296   get_more_samples' selection (unique_dim, first K)
  apply_split_results       :262-270   mean over samples into new binary16 rows, sample_counts fill
  apply_sample_results      :307-332   running average (see note) and sample_counts bump
  prune_tree                :335-381   visit-mark compaction (with orc_adjust_parents_and_children)

Pinning: the selection and prune functions are integer / ordering logic whose arithmetic lives in libtorch
(a dependency of the reference, no version pinned by its CMakeLists; torch 2.10 is what this image has).
tests/golden/make_refine_goldens.py runs the reference's own expressions through the same ATen operators
(torch.unique(dim=0), torch.cat, torch.cumsum, ...) and commits inputs + outputs as
tests/golden/refine_*.npz; tests/test_refine_host_oracle.py checks this file against them bit for bit.
apply_split_results is pinned the same way to 1 binary16 ulp (torch's reduction order is unspecified).
apply_sample_results is PARITY UNPINNED at its last step: the reference expression ends in
index_add_(binary16 self, fp32 source), which libtorch rejects, so the final rounding is this build's choice.
"""
import numpy as np


def _unique_rows(rows, return_counts=False):
    """torch.unique(rows, dim=0, sorted=True): lexicographically sorted unique rows."""
    if rows.shape[0] == 0:
        empty = rows.reshape(0, rows.shape[1])
        return (empty, np.zeros(0, np.int64)) if return_counts else empty
    return np.unique(rows, axis=0, return_counts=return_counts)


def select_split_candidates(split_track, max_out):
    """-> (nodes int32 [n][2] = (chunk, child), n_candidates)."""
    track = np.asarray(split_track, np.float32).reshape(-1, 3)
    cand = track[track[:, 1] >= 0]                                   # :206-207
    uniq, counts = _unique_rows(cand, return_counts=True)            # :209-211
    rows = np.concatenate([-counts.astype(np.int32)[:, None].astype(np.float32), uniq], axis=1)  # :214
    rows = rows[rows[:, 0] < -1]                                     # :215
    rows = _unique_rows(rows)                                        # :216-217
    n_candidates = rows.shape[0]                                     # :219
    return rows[:max_out, 2:].astype(np.int32), n_candidates         # :226


def select_sample_candidates(sample_track, max_out):
    track = np.asarray(sample_track, np.float32).reshape(-1, 3)
    cand = track[track[:, 1] >= 0]                                   # :282-283
    rows = _unique_rows(cand)                                        # :289-290
    return rows[:max_out, 1:].astype(np.int32), rows.shape[0]        # :294-295


def apply_split_results(data, sample_counts, capacity, results, samples_per_corner):
    """data: float16 [max_cap][8][data_dim] (edited in place); results float32 [n_children][spc][stride]."""
    data_dim = data.shape[2]
    n_children = results.shape[0]
    acc = np.zeros((n_children, data_dim), np.float32)
    for j in range(samples_per_corner):  # fp32 sum in sample order
        acc = (acc + results[:, j, :data_dim]).astype(np.float32)
    mean = (acc / np.float32(samples_per_corner)).astype(np.float32)
    flat = data.reshape(-1, data_dim)
    flat[capacity * 8: capacity * 8 + n_children] = mean.astype(np.float16)   # :262-266
    if sample_counts is not None:
        sample_counts[capacity: capacity + n_children // 8] = samples_per_corner  # :268-269


def apply_sample_results(data, sample_counts, nodes, results, samples_per_corner):
    data_dim = data.shape[2]
    dest = nodes[:, 0].astype(np.int64) * 8 + nodes[:, 1]            # :307-308
    flat = data.reshape(-1, data_dim)
    counts = sample_counts.reshape(-1)
    new_counts = (counts[dest] + np.int16(samples_per_corner)).astype(np.int16)   # :310-311
    acc = np.zeros((nodes.shape[0], data_dim), np.float32)
    for j in range(samples_per_corner):
        acc = (acc + results[:, j, :data_dim]).astype(np.float32)    # :313
    old = flat[dest]
    scaled_old = (np.float32(samples_per_corner) * old.astype(np.float32)).astype(np.float16)  # Scalar * Half tensor -> Half
    update = ((acc - scaled_old.astype(np.float32)) / new_counts.astype(np.float32)[:, None]).astype(np.float32)  # :316-320
    flat[dest] = (old.astype(np.float32) + update).astype(np.float16)   # intent of :322 (see module note)
    counts[dest] = new_counts                                        # :324-330


def prune_tree(orc, child, parent, data, sample_counts, visited, capacity, max_capacity):
    """Edits the arrays in place; returns (new_capacity, num_deleted).  `orc` is oracle.mnv_oracle (for the
    serial restatement of adjust_parents_and_children_kernel)."""
    to_delete = (visited[:capacity] == 0)                            # :337
    num = int(to_delete.sum())                                       # :339
    if num == 0:
        visited[1:max_capacity] = 0                                  # :343
        return capacity, 0
    index_shifts = np.cumsum(to_delete, dtype=np.int32)              # :348
    first_shift_index = int(np.argmin(index_shifts))                 # :350 (first minimum of a non-decreasing sequence: 0)
    orc.adjust_parents_and_children(child, parent, capacity, first_shift_index, to_delete.astype(np.uint8), index_shifts)
    keep = np.arange(first_shift_index, capacity)[~to_delete[first_shift_index:capacity]]   # :353-356
    n_keep = keep.shape[0]
    data[first_shift_index: first_shift_index + n_keep] = data[keep].copy()       # :357-369 (chunked there)
    child[first_shift_index: first_shift_index + n_keep] = child[keep].copy()
    parent[first_shift_index: first_shift_index + n_keep] = parent[keep].copy()
    if sample_counts is not None:  # not in the reference (it leaves sample_counts uncompacted)
        sample_counts[first_shift_index: first_shift_index + n_keep] = sample_counts[keep].copy()
    visited[1:max_capacity] = 0                                      # :378
    return capacity - num, num

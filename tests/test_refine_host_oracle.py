"""The numpy restatement of the refinement host logic (oracle/refine_oracle.py) against the golden vectors
made by running the reference's libtorch expressions through the same ATen operators
(tests/golden/make_refine_goldens.py; reference src/renderer/cuda_renderer.cpp:205-381)."""
import os

import numpy as np
import pytest

import refine_oracle as ro

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def selection_cases():
    z = np.load(os.path.join(GOLD, "refine_selection.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    return z, names


_Z, _NAMES = selection_cases()


@pytest.mark.parametrize("name", _NAMES)
def test_selection_matches_torch_goldens(name):
    track, k = _Z[f"{name}/track"], int(_Z[f"{name}/k"])
    nodes, n = ro.select_split_candidates(track, k)
    assert n == int(_Z[f"{name}/split_n"]) and np.array_equal(nodes, _Z[f"{name}/split_nodes"])
    nodes, n = ro.select_sample_candidates(track, k)
    assert n == int(_Z[f"{name}/sample_n"]) and np.array_equal(nodes.reshape(-1, 2), _Z[f"{name}/sample_nodes"])


def half_ulp_distance(a_bits, b_bits):
    """Distance in binary16 representable steps (sign-magnitude bits -> monotone integers)."""
    def mono(u):
        u = u.astype(np.int32)
        return np.where(u & 0x8000, -(u & 0x7fff), u)
    return np.abs(mono(a_bits) - mono(b_bits))


def test_split_mean_matches_torch_to_one_half_ulp():
    z = np.load(os.path.join(GOLD, "refine_split_mean.npz"))
    results, want = z["results"], z["rows"]
    n_children = results.shape[0]
    data = np.zeros((n_children // 8 + 3, 8, 28), np.float16)
    counts = np.zeros((n_children // 8 + 3, 8), np.int16)
    ro.apply_split_results(data, counts, 3, results, 8)
    got = data.reshape(-1, 28)[24:].view(np.uint16)
    d = half_ulp_distance(got, want)
    assert d.max() <= 1 and (d == 0).mean() > 0.99
    assert np.all(counts[3:] == 8) and np.all(counts[:3] == 0) and np.all(data[:3] == 0)


def test_prune_matches_torch_goldens(orc):
    z = np.load(os.path.join(GOLD, "refine_prune.npz"))
    cap = int(z["capacity"])
    data, child, parent, visited = z["data"].copy(), z["child"].copy(), z["parent"].copy(), z["visited"].copy()
    new_cap, n_del = ro.prune_tree(orc, child, parent, data, None, visited, cap, visited.shape[0])
    assert new_cap == int(z["new_capacity"]) and n_del == cap - new_cap
    assert np.array_equal(data[:new_cap], z["out_data"]) and np.array_equal(child[:new_cap], z["out_child"])
    assert np.array_equal(parent[:new_cap], z["out_parent"])
    assert visited[0] == 1 and not visited[1:].any()
    # the compacted tree is a valid octree: every non-zero child offset lands inside it, parents point back
    tgt = np.arange(new_cap)[:, None] + child[:new_cap]
    nz = child[:new_cap] != 0
    assert np.all((tgt[nz] > 0) & (tgt[nz] < new_cap))
    rows, slots = np.nonzero(nz)
    assert np.array_equal(parent[tgt[nz]], rows * 8 + slots)


def test_prune_nothing_to_delete(orc):
    z = np.load(os.path.join(GOLD, "refine_prune.npz"))
    cap = 64
    child, parent, data = z["child"][:cap].copy(), z["parent"][:cap].copy(), z["data"][:cap].copy()
    visited = np.ones(cap + 8, np.int32)
    new_cap, n_del = ro.prune_tree(orc, child, parent, data, None, visited, cap, cap + 8)
    assert (new_cap, n_del) == (cap, 0) and visited[0] == 1 and not visited[1:].any()
    assert np.array_equal(child, z["child"][:cap])

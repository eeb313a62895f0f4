"""Golden outputs of the reference's three refinement kernels (src/cuda/renderer_kernel.cu:63-213), run as the reference wrote
them (oracle/_ref: the block is cut out of the reference file verbatim by oracle/Makefile.ref) on an MI355X.

Run on the GPU box:   python tests/golden/make_refine_kernel_goldens.py gpurun_out/goldens
then copy gpurun_out/goldens/ref_refine_kernels.npz and ref_refine_kernel_stats.json into tests/golden/ and commit them.
Inputs are regenerated from tests/refine_kernel_cases.py; only outputs are stored."""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402,F401

import mega_nerf_viewer_amd as mnv  # noqa: E402
import mnv_oracle as orc  # noqa: E402
import mnv_ref  # noqa: E402
import refine_kernel_cases as rk  # noqa: E402


def bits_equal(a, b):
    return bool(np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32)))


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    tmp = tempfile.mkdtemp()
    out, stats = {}, {}
    g = rk.grid(mnv)
    for variant in rk.VARIANTS:
        tree, opt, dim, parent_nodes, visited, samples = rk.add_children_inputs(mnv, variant)
        path = os.path.join(tmp, "t.npz")
        tree.save_npz(path)
        cap = tree.capacity
        ref = mnv_ref.add_children_npz(path, opt, cap + rk.N_NEW, parent_nodes, samples, visited, g)
        # the oracle's restatement on the same inputs
        v = tree.host_view()
        _, child, parent = tree.host_arrays()
        child_big = np.zeros((cap + rk.N_NEW, 8), np.int32)
        child_big[:cap] = child
        parent_big = np.zeros(cap + rk.N_NEW, np.int32)
        parent_big[:cap] = parent
        s2, c2, v2 = samples.copy(), np.full(samples.shape[:2], -1, np.int16), visited.copy()
        orc.add_children_and_generate_samples(child_big, parent_big, list(v.offset), list(v.scale), cap, opt, parent_nodes, s2, c2, v2, g)
        stats[f"add_children/{variant}"] = {
            "child_equal": bool(np.array_equal(child_big, ref["child"])), "parent_equal": bool(np.array_equal(parent_big[:cap + rk.N_NEW], ref["parent"])),
            "visited_equal": bool(np.array_equal(v2, ref["visited"])), "samples_bit_equal": bits_equal(s2, ref["samples"]),
            "samples_max_abs": float(np.abs(s2 - ref["samples"]).max()), "clusters_equal": bool(np.array_equal(c2, ref["clusters"]))}
        for k in ("samples", "clusters", "visited", "child", "parent"):
            out[f"add_children/{variant}/{k}"] = ref[k]

        tree, opt, dim, nodes, samples = rk.generate_samples_inputs(mnv, variant)
        tree.save_npz(path)
        ref = mnv_ref.generate_samples_npz(path, opt, nodes, samples, g)
        v = tree.host_view()
        _, _, parent = tree.host_arrays()
        s2, c2 = samples.copy(), np.full(samples.shape[:2], -1, np.int16)
        orc.generate_samples(parent, list(v.offset), list(v.scale), opt, nodes, s2, c2, g)
        stats[f"generate_samples/{variant}"] = {"samples_bit_equal": bits_equal(s2, ref["samples"]), "samples_max_abs": float(np.abs(s2 - ref["samples"]).max()),
                                                "clusters_equal": bool(np.array_equal(c2, ref["clusters"]))}
        out[f"generate_samples/{variant}/samples"], out[f"generate_samples/{variant}/clusters"] = ref["samples"], ref["clusters"]

    tree, to_delete, shifts = rk.adjust_inputs(mnv, orc)
    path = os.path.join(tmp, "t.npz")
    tree.save_npz(path)
    ref = mnv_ref.adjust_parents_npz(path, tree.capacity, 1, to_delete, shifts)
    _, child, parent = (a.copy() for a in tree.host_arrays())
    orc.adjust_parents_and_children(child, parent, tree.capacity, 1, to_delete, shifts)
    keep = to_delete == 0  # rows of deleted chunks are dropped by the compaction that follows; only survivors are compared
    stats["adjust_parents"] = {"child_equal_on_survivors": bool(np.array_equal(child[keep], ref["child"][keep])),
                               "parent_equal_on_survivors": bool(np.array_equal(parent[keep], ref["parent"][keep])),
                               "child_equal_everywhere": bool(np.array_equal(child, ref["child"])), "deleted": int(to_delete.sum()), "capacity": int(tree.capacity)}
    out["adjust_parents/child"], out["adjust_parents/parent"] = ref["child"], ref["parent"]
    for k, s in stats.items():
        print(k, json.dumps(s), flush=True)
    np.savez_compressed(os.path.join(outdir, "ref_refine_kernels.npz"), **out)
    with open(os.path.join(outdir, "ref_refine_kernel_stats.json"), "w") as f:
        json.dump(stats, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "goldens"))

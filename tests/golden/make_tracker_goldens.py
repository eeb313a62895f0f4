"""Golden tracker rows and visit marks from the REFERENCE's own device code (oracle/_ref, built by oracle/Makefile.ref from
/root/reference: render_voxels_trace_ray, rt_core.cuh:132-134,179-180,237-252,308-321) run on an MI355X.

Run on the GPU box:   python tests/golden/make_tracker_goldens.py gpurun_out/goldens
then copy gpurun_out/goldens/ref_trackers_*.npz and ref_tracker_stats.json into tests/golden/ and commit them.

Stored per case: the reference's split / sample tracker rows (float32 [h][w][3]) and visit marks (int32 [capacity]) for a seeded
sample_counts array and given max_depth / max_sample_count; the tree, camera and options are regenerated from tests/cases.py."""
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

import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402
import mnv_oracle as orc  # noqa: E402
import mnv_ref  # noqa: E402

# case -> (max_depth, max_sample_count, seeded sample counts?)
TRACKER_CASES = {"sh4_d6": (5, 9, True), "rgba_d5": (3, 9, False), "terrain_d7_aniso": (6, 9, True), "cfg1_sh1_d4": (9, 9, True),
                 "camera_inside": (4, 12, True)}


def sample_counts_for(name, capacity):
    return np.random.default_rng(sum(map(ord, name))).integers(0, 14, size=(capacity, 8)).astype(np.int16)


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    tmp = tempfile.mkdtemp()
    stats = {}
    for name, (max_depth, max_sc, with_counts) in TRACKER_CASES.items():
        spec = cases.CASES[name]
        tree = cases.make_tree(mnv, spec["tree"])
        cam = cases.make_camera(mnv, spec["camera"])
        opt = cases.make_options(mnv, spec["options"])
        opt.max_depth, opt.max_sample_count = max_depth, max_sc
        v = tree.host_view()
        sc = sample_counts_for(name, v.capacity) if with_counts else None
        path = os.path.join(tmp, name + ".npz")
        tree.save_npz(path)
        ref = mnv_ref.render_track_npz(path, cam.c, opt, v.capacity, sample_counts=sc, track_visit=True)
        visited = np.zeros(v.capacity, np.int32)
        sc_oracle = sc if sc is not None else np.full((v.capacity, 8), 8, np.int16)  # named: the oracle view only borrows the pointer
        o = orc.render(orc.tree_from_view(v, sample_counts=sc_oracle), cam.c, opt, want_trackers=True, visited=visited, track_visit=True)
        st = {"split_equal": bool(np.array_equal(o["split"], ref["split"])), "sample_equal": bool(np.array_equal(o["sample"], ref["sample"])),
              "visited_equal": bool(np.array_equal(visited, ref["visited"])),
              "split_rows_differing": int((o["split"] != ref["split"]).any(axis=-1).sum()),
              "sample_rows_differing": int((o["sample"] != ref["sample"]).any(axis=-1).sum()),
              "rays_with_split_candidate": int((ref["split"][..., 1] >= 0).sum()), "rays_with_sample_candidate": int((ref["sample"][..., 1] >= 0).sum()),
              "chunks_visited": int(ref["visited"].sum()), "capacity": int(v.capacity),
              "rgba_max_abs_oracle_vs_ref": float(np.abs(o["rgba"] - ref["rgba"]).max())}
        stats[name] = st
        print(name, json.dumps(st), flush=True)
        np.savez_compressed(os.path.join(outdir, f"ref_trackers_{name}.npz"), split=ref["split"], sample=ref["sample"], visited=ref["visited"],
                            max_depth=np.int32(max_depth), max_sample_count=np.int32(max_sc), with_counts=np.int32(with_counts))
        os.remove(path)
    with open(os.path.join(outdir, "ref_tracker_stats.json"), "w") as f:
        json.dump(stats, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "goldens"))

"""Generates the golden vectors under tests/golden/ by running the REFERENCE's own device code
(oracle/_ref/libmnv_ref_gfx950.so, built by oracle/Makefile.ref from /root/reference) on an MI355X.

Run on the GPU box:   python tests/golden/make_goldens.py gpurun_out/goldens
then copy gpurun_out/goldens/*.npz and ref_stats.json into tests/golden/ and commit them.

For every case of tests/cases.py the deterministic synthetic tree is written as an svox .npz with
the build's writer, opened by the reference's N3Tree::open (3rdparty/cnpy), moved to the device by
the reference's move_to_device and rendered by the reference's render_voxels_trace_ray.  Stored per
case: the float RGBA frame (float32), loader probes (first elements of data/child/parent as the
reference loader produced them).  ref_stats.json records, measured in the same run, how the CPU
oracle and the HIP kernels compare with these vectors (max |d|, pixels > 1e-6 / 1e-5 / 1e-4), also for
the headline cfg2 frame at 1920x1080 and for a default-contraction build of the reference."""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402
import mnv_oracle as orc  # noqa: E402
import mnv_ref  # noqa: E402


def cmp(a, b):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max(axis=-1)
    return {"max_abs": float(d.max()), "px_gt_1e-6": int((d > 1e-6).sum()), "px_gt_1e-5": int((d > 1e-5).sum()),
            "px_gt_1e-4": int((d > 1e-4).sum()), "px_not_bit_identical": int((a.view(np.uint32) != b.view(np.uint32)).any(axis=-1).sum())}


def hip_render(tree, cam, opt):
    out = torch.empty((cam.height, cam.width, 4), dtype=torch.float32, device="cuda")
    mnv.render_voxels_accel(tree.accel, cam, opt, rgba=out)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    stats = {}
    tmp = tempfile.mkdtemp()
    todo = [(name, spec["tree"], cases.make_camera(mnv, spec["camera"]), cases.make_options(mnv, spec["options"]), True)
            for name, spec in cases.CASES.items()]
    todo.append(("cfg2_pose3_1920x1080", cases.CFG2_TREE, cases.cfg2_camera(mnv, 3), mnv.RenderOptions.cli_defaults(), False))
    for name, tspec, cam, opt, store_full in todo:
        tree = cases.make_tree(mnv, tspec)
        path = os.path.join(tmp, name + ".npz")
        tree.save_npz(path)
        ref = mnv_ref.render_npz(path, cam.c, opt)
        d, c, p = tree.host_arrays()
        n = len(ref["data_probe"])
        loader_ok = (np.array_equal(ref["data_probe"][: min(n, d.size)], d.reshape(-1)[:n]) and
                     np.array_equal(ref["child_probe"][: min(n, c.size)], c.reshape(-1)[:n]) and
                     np.array_equal(ref["parent_probe"][: min(n, p.size)], p.reshape(-1)[:n]))
        v = tree.host_view()
        meta_ok = ref["meta"] == [v.N, v.data_dim, v.format, v.basis_dim, v.capacity]
        o = orc.render(orc.tree_from_view(v), cam.c, opt)
        tree.move_to_device()
        hip = hip_render(tree, cam, opt)
        st = {"oracle_vs_ref": cmp(o["rgba"], ref["rgba"]), "hip_vs_ref": cmp(hip, ref["rgba"]), "hip_vs_oracle": cmp(hip, o["rgba"]),
              "reference_loader_matches_build_loader": bool(loader_ok and meta_ok), "shape": list(ref["rgba"].shape),
              "counters": o["counters"].as_dict()}
        if mnv_ref.available(contract=True):
            refc = mnv_ref.render_npz(path, cam.c, opt, contract=True)
            st["ref_contract_fast_vs_ref_contract_off"] = cmp(refc["rgba"], ref["rgba"])
        stats[name] = st
        print(name, json.dumps(st), flush=True)
        if store_full:
            np.savez_compressed(os.path.join(outdir, f"ref_{name}.npz"), rgba=ref["rgba"], data_probe=ref["data_probe"],
                                child_probe=ref["child_probe"], parent_probe=ref["parent_probe"], meta=np.int32(ref["meta"]))
        else:
            rng = np.random.default_rng(12345)
            idx = np.sort(rng.choice(cam.width * cam.height, 16384, replace=False))
            flat = ref["rgba"].reshape(-1, 4)
            np.savez_compressed(os.path.join(outdir, f"ref_{name}.npz"), idx=idx.astype(np.int32), rgba_at_idx=flat[idx],
                                sum=flat.astype(np.float64).sum(axis=0), sumsq=(flat.astype(np.float64) ** 2).sum(axis=0),
                                meta=np.int32(ref["meta"]))
        os.remove(path)
        del tree
    # ---- guided-sampling pair (rt_core.cuh:334-576) against the reference's own device functions
    import guided_cases
    tree, cam, opt, dim = guided_cases.get_samples_setup(mnv)
    path = os.path.join(tmp, "guided.npz")
    tree.save_npz(path)
    grid = guided_cases.cluster_grid(mnv.ClusterGrid)
    ref = mnv_ref.get_samples_npz(path, cam.c, opt, grid, dim)
    o = orc.get_samples(orc.tree_from_view(tree.host_view()), cam.c, opt, grid, dim)
    k = np.arange(opt.max_guided_samples)[None, :] < ref["num_samples"][:, None]   # emitted rows only
    stats["guided_get_samples"] = {
        "num_samples_equal": bool(np.array_equal(ref["num_samples"], o["num_samples"])),
        "clusters_equal": bool(np.array_equal(ref["cluster_indices"][k], o["cluster_indices"][k])),
        "samples_max_abs": float(np.abs(ref["samples"][k].astype(np.float64) - o["samples"][k].astype(np.float64)).max()),
        "samples_not_bit_identical": int((ref["samples"][k].view(np.uint32) != o["samples"][k].view(np.uint32)).sum()),
        "total_samples": int(ref["num_samples"].astype(np.int64).sum())}
    print("guided_get_samples", stats["guided_get_samples"], flush=True)
    np.savez_compressed(os.path.join(outdir, "ref_guided_get_samples.npz"), num_samples=ref["num_samples"],
                        samples=np.where(k[..., None], ref["samples"], np.float32(-1)), cluster_indices=np.where(k, ref["cluster_indices"], -1).astype(np.int16))
    for case in ("sh4_d6", "rgba_d5"):
        tree, cam, opt, values, z, offsets = guided_cases.nerf_results_setup(mnv, case)
        path = os.path.join(tmp, f"guided_{case}.npz")
        tree.save_npz(path)
        refimg = mnv_ref.render_nerf_results_npz(path, cam.c, opt, values, z, offsets)
        oimg = orc.render_nerf_results(orc.tree_from_view(tree.host_view()), cam.c, opt, values, z, offsets)["rgba"]
        stats[f"guided_nerf_results_{case}"] = {"oracle_vs_ref": cmp(oimg, refimg)}
        print(case, stats[f"guided_nerf_results_{case}"], flush=True)
        np.savez_compressed(os.path.join(outdir, f"ref_guided_nerf_results_{case}.npz"), rgba=refimg)
    with open(os.path.join(outdir, "ref_stats.json"), "w") as f:
        json.dump(stats, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "goldens"))

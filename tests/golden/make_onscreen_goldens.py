"""Golden vectors of the reference's LIVE call shape (render_voxels with offscreen == false, src/renderer/cuda_renderer.cpp:141-142):
the reference's own march (rt_core.cuh, compiled for gfx950 by oracle/Makefile.ref) run with a per-pixel t_max from a depth image
(renderer_kernel.cu:277-280) and composited over an image (renderer_kernel.cu:230-234) -- the per-pixel wrapper is restated in
oracle/ref_driver.hip because the reference reads both through surface objects, which gfx950 does not have.

Run on the GPU box:   python tests/golden/make_onscreen_goldens.py gpurun_out/goldens
then copy gpurun_out/goldens/ref_onscreen_*.npz (incl. ref_onscreen_trackers_both.npz), ref_guided_get_samples_onscreen.npz and ref_onscreen_stats.json into tests/golden/ and commit them.
Inputs are tests/cases.py::onscreen_inputs (seeded); only the reference's float RGBA frames are stored."""
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
from make_goldens import cmp  # noqa: E402


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    tmp = tempfile.mkdtemp()
    stats = {}
    for name, (base, which, _) in cases.ONSCREEN.items():
        spec = cases.CASES[base]
        tree = cases.make_tree(mnv, spec["tree"])
        cam = cases.make_camera(mnv, spec["camera"])
        opt = cases.make_options(mnv, spec["options"])
        tmax, image = cases.onscreen_inputs(name, cam)
        path = os.path.join(tmp, name + ".npz")
        tree.save_npz(path)
        ref = mnv_ref.render_onscreen_npz(path, cam.c, opt, tmax_px=tmax, rgba8_init=image)
        plain = mnv_ref.render_npz(path, cam.c, opt, n_probe=0)["rgba"]
        o = orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt, tmax_px=tmax, rgba8_init=image)["rgba"]
        tree.move_to_device()
        out = torch.empty((cam.height, cam.width, 4), dtype=torch.float32, device="cuda")
        mnv.render_voxels_accel(tree.accel, cam, opt, rgba=out, tmax_px=None if tmax is None else torch.from_numpy(tmax).cuda(),
                                rgba8_init=None if image is None else torch.from_numpy(image).cuda())
        torch.cuda.synchronize()
        hip = out.cpu().numpy()
        stats[name] = {"base_case": base, "inputs": list(which), "oracle_vs_ref": cmp(o, ref), "hip_vs_ref": cmp(hip, ref), "hip_vs_oracle": cmp(hip, o),
                       "pixels_changed_by_the_inputs": int((ref != plain).any(axis=-1).sum()), "shape": list(ref.shape)}
        print(name, json.dumps(stats[name]), flush=True)
        np.savez_compressed(os.path.join(outdir, f"ref_{name}.npz"), rgba=ref)
        os.remove(path)
    # ---- get_samples_from_voxels with offscreen == false: the depth attachment limits every ray (renderer_kernel.cu:354-357)
    import guided_cases
    tree, cam, opt, dim = guided_cases.get_samples_setup(mnv)
    opt.max_depth, opt.max_sample_count = 5, 9
    tmax = guided_cases.onscreen_tmax(cam)
    path = os.path.join(tmp, "guided_onscreen.npz")
    tree.save_npz(path)
    grid = guided_cases.cluster_grid(mnv.ClusterGrid)
    ref = mnv_ref.get_samples_npz(path, cam.c, opt, grid, dim, tmax_px=tmax)
    plain = mnv_ref.get_samples_npz(path, cam.c, opt, grid, dim)
    drop = mnv_ref.get_samples_npz(path, cam.c, opt, grid, dim, tmax_px=tmax, dropin=True)
    counts = np.full((tree.host_view().capacity, 8), 8, np.int16)   # what the driver gives the reference's tree (ref_driver.hip: sample_counts.fill_(8))
    o = orc.get_samples(orc.tree_from_view(tree.host_view(), sample_counts=counts), cam.c, opt, grid, dim, tmax_px=tmax)
    k = np.arange(opt.max_guided_samples)[None, :] < ref["num_samples"][:, None]   # emitted rows only

    def same(a):
        return bool(np.array_equal(ref["num_samples"], a["num_samples"]) and np.array_equal(ref["cluster_indices"][k], a["cluster_indices"][k]) and
                    np.array_equal(ref["samples"][k].view(np.uint32), a["samples"][k].view(np.uint32)) and
                    np.array_equal(ref["split"].view(np.uint32), a["split"].view(np.uint32)) and np.array_equal(ref["sample"].view(np.uint32), a["sample"].view(np.uint32)))

    stats["guided_get_samples_onscreen"] = {
        "oracle_equals_ref": same(o), "binding_dropin_equals_ref": same(drop),
        "rays_changed_by_the_depth_image": int((ref["num_samples"] != plain["num_samples"]).sum()),
        "total_samples": int(ref["num_samples"].astype(np.int64).sum()), "total_samples_offscreen": int(plain["num_samples"].astype(np.int64).sum())}
    print("guided_get_samples_onscreen", stats["guided_get_samples_onscreen"], flush=True)
    np.savez_compressed(os.path.join(outdir, "ref_guided_get_samples_onscreen.npz"), num_samples=ref["num_samples"],
                        samples=np.where(k[..., None], ref["samples"], np.float32(-1)), cluster_indices=np.where(k, ref["cluster_indices"], -1).astype(np.int16),
                        split=ref["split"], sample=ref["sample"])
    # ---- the frame the render loop launches (cuda_renderer.cpp:141-142): trackers and visit marks AND offscreen == false
    name = "onscreen_both"
    spec = cases.CASES[cases.ONSCREEN[name][0]]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.max_depth, opt.max_sample_count = 5, 9
    tmax, image = cases.onscreen_inputs(name, cam)
    v = tree.host_view()
    counts = np.random.default_rng(7).integers(0, 14, size=(v.capacity, 8)).astype(np.int16)
    path = os.path.join(tmp, "track_onscreen.npz")
    tree.save_npz(path)
    ref = mnv_ref.render_track_npz(path, cam.c, opt, v.capacity, sample_counts=counts, track_visit=True, tmax_px=tmax, rgba8_init=image)
    plain = mnv_ref.render_track_npz(path, cam.c, opt, v.capacity, sample_counts=counts, track_visit=True)
    marks = np.zeros(v.capacity, np.int32)
    o = orc.render(orc.tree_from_view(v, sample_counts=counts), cam.c, opt, want_trackers=True, visited=marks, track_visit=True, tmax_px=tmax, rgba8_init=image)
    stats["trackers_onscreen_both"] = {
        "oracle_vs_ref": cmp(o["rgba"], ref["rgba"]),
        "trackers_equal": bool(np.array_equal(o["split"].view(np.uint32), ref["split"].view(np.uint32)) and
                               np.array_equal(o["sample"].view(np.uint32), ref["sample"].view(np.uint32))),
        "marks_equal": bool(np.array_equal(marks, ref["visited"])),
        "split_rows_changed_by_the_inputs": int((ref["split"] != plain["split"]).any(axis=-1).sum()),
        "sample_rows_changed_by_the_inputs": int((ref["sample"] != plain["sample"]).any(axis=-1).sum()),
        "marks_changed_by_the_inputs": int((ref["visited"] != plain["visited"]).sum())}
    print("trackers_onscreen_both", json.dumps(stats["trackers_onscreen_both"]), flush=True)
    np.savez_compressed(os.path.join(outdir, "ref_onscreen_trackers_both.npz"), rgba=ref["rgba"], split=ref["split"], sample=ref["sample"], visited=ref["visited"])
    with open(os.path.join(outdir, "ref_onscreen_stats.json"), "w") as f:
        json.dump(stats, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "goldens"))

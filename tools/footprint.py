"""Unique 128-byte lines ONE launch of the march touches, per array -- a floor under the bytes the launch must bring in from HBM when its working
set is far larger than the caches (cfg2 / cfg3 / cfg4: yes; see DESIGN.md section 7 for fog).  Runs the bench workloads' launches once on the
test-hook build's diagnostics instantiation (march_accel_kernel<9,256,1,true> with MNV_FOOTPRINT=<file>: every load sets the bit of its line,
mnv_accel_destroy counts them) and prints one JSON line per workload.  bench.py starts this script as a child process (the shipped library it
loads itself has no such hook) and folds the numbers into its roofline objects as `footprint_bytes` / `frac_footprint`.

usage: python tools/footprint.py [cfg2] [cfg3] [cfg4] [fog]        (needs mega-nerf-viewer_amd/testhooks/libmnv.so)"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOOKS = os.path.join(ROOT, "mega-nerf-viewer_amd", "testhooks", "libmnv.so")
FRAME_BY_FRAME = ("cfg2", "cfg2_small", "cfg3")   # workloads that also get the frame-by-frame pass (one accel per distinct pose)


def child(workloads, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch

    import cases
    import mega_nerf_viewer_amd as mnv

    opt = mnv.RenderOptions.cli_defaults()
    trees, out = {}, {}
    for workload in workloads:
        kind = "cfg3" if workload in ("cfg3", "cfg4") else "cfg2" if workload == "cfg2_small" else workload
        if kind not in trees:
            trees.clear()   # one big tree at a time
            torch.cuda.empty_cache()
            tree = cases.make_tree(mnv, {"cfg2": cases.CFG2_TREE, "fog": cases.FOG_TREE, "cfg3": cases.CFG3_FULL}[kind])
            tree.move_to_device()
            trees[kind] = tree
        tree = trees[kind]
        if workload == "cfg2":
            w, h = 1920, 1080
            cams = [cases.cfg2_camera(mnv, p % 16, w, h, 1600.0) for p in range(64)]
        elif workload == "cfg2_small":   # (the tests' size)
            w, h = 640, 360
            cams = [cases.cfg2_camera(mnv, p, w, h, 1600.0 * w / 1920) for p in (0, 5, 9, 13)]
        elif workload == "fog":
            w, h = 1920, 1080
            cams = [cases.cfg2_camera(mnv, p, w, h, 1600.0) for p in range(16)]
        else:
            w, h = (1920, 1080) if workload == "cfg3" else (3840, 2160)
            cams = [cases.cfg3_camera(mnv, p, w, h, fx=1400.0 * w / 1920) for p in range(16)]
        accel = mnv.accel_create(tree.device_view())   # an accel of its own per measurement: the bits of its launches accumulate until it is destroyed
        frames = torch.empty((len(cams), h, w, 4), dtype=torch.float32, device="cuda")
        mnv.render_voxels_accel_batch(accel, cams, opt, rgba=frames)
        torch.cuda.synchronize()
        del frames
        rays = len(cams) * w * h
        mnv.accel_destroy(accel)          # writes MNV_FOOTPRINT's file
        d = json.load(open(os.environ["MNV_FOOTPRINT"]))
        os.remove(os.environ["MNV_FOOTPRINT"])
        lines = sum(v for k, v in d.items() if k.endswith("_lines"))
        d.update(workload=workload, frames_per_launch=len(cams), resolution=f"{w}x{h}", output_bytes=rays * 16, footprint_bytes=lines * 128 + rays * 16)
        # ... and frame by frame: the lines ONE frame touches, for every distinct pose of the launch (an accel each), summed over the launch's
        # frames.  A frame's lines (cfg2: ~0.2 GB) do not survive in 32 MiB of L2 until the same pose comes round again, so a launch brings
        # in at least this much unless the Infinity Cache (256 MiB) carries lines from one frame to the next.
        if workload in FRAME_BY_FRAME:
            distinct = {}
            for i, c in enumerate(cams):
                distinct.setdefault(bytes(c.c), []).append(i)
            per_pose = []
            frame = torch.empty((1, h, w, 4), dtype=torch.float32, device="cuda")
            for key, idx in distinct.items():
                accel = mnv.accel_create(tree.device_view())
                mnv.render_voxels_accel_batch(accel, [cams[idx[0]]], opt, rgba=frame)
                torch.cuda.synchronize()
                mnv.accel_destroy(accel)
                dd = json.load(open(os.environ["MNV_FOOTPRINT"]))
                os.remove(os.environ["MNV_FOOTPRINT"])
                per_pose.append((sum(v for k, v in dd.items() if k.endswith("_lines")), len(idx)))
            del frame
            d.update(frame_lines_min=min(p[0] for p in per_pose), frame_lines_max=max(p[0] for p in per_pose),
                     footprint_frame_by_frame_bytes=sum(n * (ln * 128 + w * h * 16) for ln, n in per_pose))
        out[workload] = d
    with open(out_path, "w") as f:
        json.dump(out, f)


def measure(workloads):
    """-> {workload: dict}; raises if the test-hook build is missing or the pass fails"""
    if not os.path.exists(HOOKS):
        raise RuntimeError("mega-nerf-viewer_amd/testhooks/libmnv.so is missing (make builds it)")
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, MNV_LIB_PATH=HOOKS, MNV_FOOTPRINT=os.path.join(d, "lines.json"))
        res = os.path.join(d, "out.json")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", ",".join(workloads), res], env=env, capture_output=True, text=True, timeout=1500)
        if r.returncode != 0 or not os.path.exists(res):
            raise RuntimeError(f"footprint pass of {workloads} failed: {r.stderr[-800:]}")
        return json.load(open(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2].split(","), sys.argv[3])
    else:
        for wl, d in measure(sys.argv[1:] or ["cfg2", "cfg3", "cfg4", "fog"]).items():
            print(json.dumps(d), flush=True)

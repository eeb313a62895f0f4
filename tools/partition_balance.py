"""Per-rank launch times of the interleaved macro-tile partition, measured one rank at a time on ONE GPU.

For world = 1, 2, 4, 8 every rank's batched launch (the bench step: 64 frames, RGBA8 tiles) is timed alone with HIP
events; max over ranks is the render time a world-size-N run cannot beat, so
    ceiling(N) = t(1) / (N * max_r t_r(N))
is the strong-scaling efficiency the partition allows before any gather cost.  Usage (GPU box):
    python3 tools/partition_balance.py [--macro 128x120] [--laps 4] [--worlds 2,4,8]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import torch  # noqa: E402

import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402
from mega_nerf_viewer_amd.multigpu import TilePartition  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--macro", default="128x120")
    ap.add_argument("--laps", type=int, default=4)
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    W, H, FX = 1920, 1080, 1600.0
    n_frames = 16 * args.laps
    tree = cases.make_tree(mnv, cases.CFG2_TREE)
    tree.move_to_device()
    opt = mnv.RenderOptions.cli_defaults()
    cams = [cases.cfg2_camera(mnv, p % 16, W, H, FX) for p in range(n_frames)]
    dev = torch.device("cuda", 0)

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        mnv.set_timing(True)
        for _ in range(args.reps):
            fn()
        torch.cuda.synchronize()
        ms, n = mnv.take_timing()
        mnv.set_timing(False)
        return ms / n

    full = torch.empty((n_frames, H, W, 4), dtype=torch.uint8, device=dev)
    t1 = timed(lambda: mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba8=full))
    out = {"frames_per_launch": n_frames, "t1_ms": round(t1, 4), "macros": []}
    for macro in args.macro.split(","):
        mw, mh = (int(v) for v in macro.split("x"))
        row = {"macro": macro, "worlds": {}}
        for world in (int(v) for v in args.worlds.split(",")):
            part = TilePartition(W, H, world, mw, mh)
            buf = torch.empty((n_frames, part.j_max, mh, mw, 4), dtype=torch.uint8, device=dev)
            ts = [timed(lambda r=r: mnv.render_voxels_accel_batch(tree.accel, cams, opt, part=(r, world, mw, mh), rgba8=buf)) for r in range(world)]
            row["worlds"][world] = {"per_rank_ms": [round(t, 4) for t in ts], "max_ms": round(max(ts), 4), "sum_ms": round(sum(ts), 4),
                                    "ceiling": round(t1 / (world * max(ts)), 4)}
        out["macros"].append(row)
        print(json.dumps(row), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""Where a single-frame launch of the tuned march spends its time (diagnostics instantiation, MNV_TIMELINE).

    MNV_TIMELINE=/tmp/tl.bin python tools/timeline.py [--workload cfg2|cfg3] [--pose 3] [--frames 1]

The kernel stamps (100 MHz device clock) when every 8x8 tile was grabbed and finished, by which wavefront, and when every
wavefront entered and left the kernel; libmnv writes the records of the last launch when the accel is destroyed.  This script
renders, lets the handle go and reports: kernel span, wavefront exit-time distribution, busy wavefronts over time, the
longest tiles and when they were started -- what an order that starts expensive tiles first could save.
"""
import os as _os
# the MNV_* knobs this tool reads exist in the test-hook build of the library only (csrc/mnv_knobs.h)
_os.environ.setdefault("MNV_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mega-nerf-viewer_amd", "testhooks", "libmnv.so"))
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def analyse(path, label=""):
    raw = np.fromfile(path, dtype=np.uint64)
    n_tiles, n_waves, per_frame = int(raw[0]), int(raw[1]), int(raw[2])
    tiles = raw[3:3 + 4 * n_tiles].reshape(n_tiles, 4).astype(np.int64)
    waves = raw[3 + 4 * n_tiles:3 + 4 * n_tiles + 2 * n_waves].reshape(n_waves, 2).astype(np.int64)
    t0 = waves[:, 0][waves[:, 0] > 0].min()
    entry = (waves[:, 0] - t0) / 100.0   # us
    exit_ = (waves[:, 1] - t0) / 100.0
    done = tiles[:, 1] > 0
    grab = (tiles[done, 0] - t0) / 100.0
    fin = (tiles[done, 1] - t0) / 100.0
    dur = fin - grab
    span = exit_.max()
    out = {"label": label, "tiles": n_tiles, "tiles_recorded": int(done.sum()), "waves": n_waves, "kernel_span_us": round(float(span), 1),
           "wave_entry_us_p50_p99_max": [round(float(np.percentile(entry, q)), 1) for q in (50, 99, 100)],
           "wave_exit_us_p10_p50_p90_p99": [round(float(np.percentile(exit_, q)), 1) for q in (10, 50, 90, 99)],
           "tile_us_mean_p50_p90_p99_max": [round(float(x), 1) for x in (dur.mean(), *np.percentile(dur, (50, 90, 99, 100)))],
           "tiles_per_wave_mean_max": [round(float(done.sum() / n_waves), 2), int(np.bincount(tiles[done, 2]).max())],
           "wave_busy_fraction": round(float(dur.sum() / (span * n_waves)), 3),
           "sum_tile_us_over_waves": round(float(dur.sum() / n_waves), 1)}
    # the 1 % longest tiles: when were they grabbed?
    k = max(1, len(dur) // 100)
    idx = np.argsort(dur)[-k:]
    out["longest_1pct_grab_us_p50_max"] = [round(float(np.percentile(grab[idx], 50)), 1), round(float(grab[idx].max()), 1)]
    out["longest_1pct_dur_us_min"] = round(float(dur[idx].min()), 1)
    # busy wavefronts over time (10 bins)
    bins = np.linspace(0, span, 11)
    busy = []
    for a, b in zip(bins[:-1], bins[1:]):
        ov = np.clip(np.minimum(fin, b) - np.maximum(grab, a), 0, None).sum() / (b - a)
        busy.append(round(float(ov / n_waves), 2))
    out["busy_fraction_by_decile"] = busy
    # lower bound of a longest-first schedule: max(longest tile, total / waves)
    out["lpt_bound_us"] = round(float(max(dur.max(), dur.sum() / n_waves)), 1)
    last = int(np.argmax(fin))
    out["last_tile"] = {"grab_us": round(float(grab[last]), 1), "dur_us": round(float(dur[last]), 1), "iters": int(tiles[done][last, 3])}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--pose", type=int, default=3)
    ap.add_argument("--frames", type=int, default=1, help="frames per launch (batched when > 1)")
    ap.add_argument("--analyse", default=None, help="only analyse this file")
    args = ap.parse_args()
    if args.analyse:
        print(json.dumps(analyse(args.analyse)))
        return
    path = os.environ.get("MNV_TIMELINE")
    if not path:
        raise SystemExit("set MNV_TIMELINE=<file>")
    import torch
    import cases
    import mega_nerf_viewer_amd as mnv

    W, H = 1920, 1080
    if args.workload == "cfg2":
        tree = cases.make_tree(mnv, cases.CFG2_TREE)
        cams = [cases.cfg2_camera(mnv, (args.pose + i) % 16, W, H, 1600.0) for i in range(args.frames)]
    else:
        tree = cases.make_tree(mnv, cases.CFG3_TREE)
        cams = [cases.cfg3_camera(mnv, (args.pose + i) % 16, W, H, fx=1400.0) for i in range(args.frames)]
    tree.move_to_device()
    opt = mnv.RenderOptions.cli_defaults()
    out = torch.empty((args.frames, H, W, 4), device="cuda")
    for _ in range(4):
        if args.frames == 1:
            mnv.render_voxels_accel(tree.accel, cams[0], opt, rgba=out[0])
        else:
            mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba=out)
    torch.cuda.synchronize()
    tree.release() if hasattr(tree, "release") else None
    del tree
    import gc

    gc.collect()
    if not os.path.exists(path):
        raise SystemExit("no timeline file was written (is the accel still referenced?)")
    print(json.dumps(analyse(path, f"{args.workload} pose {args.pose} frames {args.frames}")))


if __name__ == "__main__":
    main()

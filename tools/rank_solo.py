"""One rank of an N-GPU run alone on ONE GPU: back-to-back steps (batched launches of the rank's macro tiles, RGBA8), no gather.
Wall time per step = what the march contributes to a world-size-N step; --reserve applies the CU mask of the multi-GPU path,
--streams 2 alternates launches between two (equally masked) streams so that the tail of one launch overlaps the head of the next.
    python3 tools/rank_solo.py --world 8 --rank 0 --reserve 32 --streams 1
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import torch  # noqa: E402

import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402
from mega_nerf_viewer_amd.multigpu import TilePartition  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", default="8")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--reserves", default="0,32")
    ap.add_argument("--streams", default="1,2")
    ap.add_argument("--macro", default="64x24")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--laps", type=int, default=4)
    ap.add_argument("--workload", choices=["cfg2", "cfg4"], default="cfg2", help="cfg4 = the merged-octree stand-in at 3840x2160 (BASELINE.json configs[3])")
    args = ap.parse_args()
    W, H, FX = (1920, 1080, 1600.0) if args.workload == "cfg2" else (3840, 2160, 2800.0)
    nf = 16 * args.laps
    mw, mh = (int(v) for v in args.macro.split("x"))
    dev = torch.device("cuda", 0)
    tree = cases.make_tree(mnv, cases.CFG2_TREE if args.workload == "cfg2" else cases.CFG3_TREE)
    tree.move_to_device()
    opt = mnv.RenderOptions.cli_defaults()
    if args.workload == "cfg2":
        cams = [cases.cfg2_camera(mnv, p % 16, W, H, FX) for p in range(nf)]
    else:
        cams = [cases.cfg3_camera(mnv, p % 16, W, H, fx=FX) for p in range(nf)]
    rows = []
    for reserve in (int(v) for v in args.reserves.split(",")):
        for n_streams in (int(v) for v in args.streams.split(",")):
            handles = []
            for _ in range(n_streams):
                h, enabled = mnv.stream_create_reserved(reserve)
                handles.append(h)
            mnv.accel_set_cu_budget(tree.accel, enabled)
            for world in (int(v) for v in args.worlds.split(",")):
                part = TilePartition(W, H, world, mw, mh)
                bufs = [torch.empty((nf, part.j_max, mh, mw, 4), dtype=torch.uint8, device=dev) for _ in range(n_streams)]
                kw = dict(part=(args.rank, world, mw, mh)) if world > 1 else {}
                if world == 1:
                    bufs = [torch.empty((nf, H, W, 4), dtype=torch.uint8, device=dev) for _ in range(n_streams)]

                def run(n):
                    for k in range(n):
                        mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba8=bufs[k % n_streams], stream=handles[k % n_streams], **kw)
                    torch.cuda.synchronize()

                run(4)
                t0 = time.perf_counter()
                run(args.steps)
                ms = (time.perf_counter() - t0) / args.steps * 1e3
                row = {"workload": args.workload, "world": world, "rank": args.rank, "reserve": reserve, "streams": n_streams, "ms_per_step": round(ms, 4),
                       "Mrays_per_s_if_all_ranks_alike": round(nf * W * H / ms / 1e3, 1)}
                rows.append(row)
                print(json.dumps(row), flush=True)
            for h in handles:
                mnv.stream_destroy(h)
    mnv.accel_set_cu_budget(tree.accel, 0)


if __name__ == "__main__":
    main()

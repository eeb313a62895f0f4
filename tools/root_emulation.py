"""What rank 0 of an 8-GPU run pays for being the root, emulated on ONE GPU: its own share of the march (CU-masked, two streams) while
a side stream moves the bytes the gather would write (7/8 of the step's RGBA8 tiles, as a device copy) and un-permutes the step's frames.
    python3 tools/root_emulation.py [--world 8]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]

import torch  # noqa: E402

import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402
from mega_nerf_viewer_amd.multigpu import TilePartition  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--root-periods", default="0,7", help="mnv_partition.root_period values to compare (0 = plain round robin)")
    args = ap.parse_args()
    W, H, FX, NF, MW, MH = 1920, 1080, 1600.0, 64, 64, 24
    dev = torch.device("cuda", 0)
    tree = cases.make_tree(mnv, cases.CFG2_TREE)
    tree.move_to_device()
    opt = mnv.RenderOptions.cli_defaults()
    cams = [cases.cfg2_camera(mnv, p % 16, W, H, FX) for p in range(NF)]
    handles = []
    for _ in range(2):
        h, enabled = mnv.stream_create_reserved(32)
        handles.append(h)
    mnv.accel_set_cu_budget(tree.accel, enabled)
    side = torch.cuda.Stream(device=dev)
    for M in (int(v) for v in args.root_periods.split(",")):
        part = TilePartition(W, H, args.world, MW, MH, M)
        local = [torch.zeros((NF, part.j_max, MH, MW, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
        gathered = torch.zeros((args.world, NF, part.j_max, MH, MW, 4), dtype=torch.uint8, device=dev)
        incoming = torch.zeros((args.world - 1, NF, part.j_max, MH, MW, 4), dtype=torch.uint8, device=dev)  # stands for the peers' buffers
        frames = torch.empty((NF, H, W, 4), dtype=torch.uint8, device=dev)
        import bench
        res = {"world": args.world, "root_period": M, "tiles_per_rank": [part.local_tiles(r) for r in range(args.world)], "kernel_source_sha": bench.kernel_source_sha()}
        for rank, root_work in ((1, False), (0, False), (0, True)):
            def run(n):
                for k in range(n):
                    mnv.render_voxels_accel_batch(tree.accel, cams, opt, part=part.part(rank), rgba8=local[k % 2], stream=handles[k % 2])
                    if root_work:
                        with torch.cuda.stream(side):
                            gathered[1:].copy_(incoming, non_blocking=True)       # the bytes RCCL's receive would write
                            part.unpermute(gathered, out=frames)                   # mnv_assemble_tiles on the side stream
                torch.cuda.synchronize()

            run(4)
            t0 = time.perf_counter()
            run(args.steps)
            key = "rank1_march_only" if rank == 1 else ("rank0_with_root_work" if root_work else "rank0_march_only")
            res[key + "_ms"] = round((time.perf_counter() - t0) / args.steps * 1e3, 4)
        res["step_ms"] = max(res["rank1_march_only_ms"], res["rank0_with_root_work_ms"])
        print(json.dumps(res), flush=True)
    for h in handles:
        mnv.stream_destroy(h)


if __name__ == "__main__":
    main()

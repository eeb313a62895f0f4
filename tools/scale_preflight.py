#!/usr/bin/env python3
"""First contact with a multi-GPU node: what bench.py --gpus N and mnv_render --gpus N rely on, checked item by item on however many GPUs
this node has (or --gpus K), with a pass / fail line per item and the numbers next to it.

    peer access        hipDeviceCanAccessPeer between every pair of devices (RCCL's xGMI path needs it)
    communicator       mnv_comm_init_rank on every rank (RCCL bound by libmnv, one process per GPU), RCCL version
    gather 4.1 MB      mnv_gather_tiles to rank 0 at the message size of a 1920x1080 RGBA8 batch share (and 16.6 MB: 3840x2160):
                       payload checked on the root, time and GB/s into the root
    all-gather         mnv_allgather at the tracker-row size of a refinement frame (grouped send / receive between ALL pairs): payload
                       checked on every rank
    masked stream      mnv_stream_create_reserved(32) under N processes: did the CU mask take effect (enabled < device units), and does a
                       kernel on the masked stream still run beside the others

The parent never touches the GPU (it only starts one child per rank; this pool forbids a GPU-initialised process to exec); the ranks
exchange RCCL's id through a gloo store on 127.0.0.1.  Exit code 0 = every item passed.  usage: scale_preflight.py [--gpus K] [--json]"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


NEXT_STEPS = """
next: the scaling curve, one command per point (each prints ONE JSON line; `value` = Mrays/s of the whole job, `per_rank` and `predicted` say where the time went):
    for n in 1 2 4 {n}; do python bench.py --gpus $n --steps 10 --warmup 2 --no-extras > scale_n$n.json; done
the three assumptions behind `predicted` (one-GPU emulation of rank 0, profiles/r*_root_emulation.jsonl) and the number in the line that tests each:
    (1) RCCL's receive costs rank 0 no more than a device copy of the same bytes
            broken if  per_rank[0].gather_ms + per_rank[0].unpermute_ms > 1.0 ms per 64-frame step at N = 8   (emulation: 0.35 ms; `gather 4.1 MB` above: >= 100 GB/s into the root)
    (2) the CU-masked march stream keeps 32 units free in EVERY process
            broken if  per_rank[r].cu_mask_in_effect is false, or per_rank[r].march_ms > 1.15 x predicted.other_ranks_march_ms   (`masked stream` above)
    (3) the peers' sends arrive while rank 0 marches (point-to-point xGMI links are not the bound)
            broken if  per_rank[r > 0].gather_ms > 2 x per_rank[0].gather_ms   (a send that waits for rank 0's receive)
    a step above 1.25 x predicted.ms_per_step with none of the three broken: rank 0's own march (per_rank[0].march_ms against predicted.rank0_march_only_ms) -- raise --root-period
if the first rung fails, bench.py --gpus N walks its ladder by itself (no CU reservation -> float tiles -> torch.distributed's gather) and names every failed rung in `launch`."""


def worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist

    import mega_nerf_viewer_amd as mnv

    res = {"rank": rank, "items": []}

    def item(name, ok, **info):
        res["items"].append(dict(name=name, ok=bool(ok), **info))

    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    # 1. peer access
    peers = {j: bool(torch.cuda.can_device_access_peer(rank, j)) for j in range(world) if j != rank}
    item("peer access", all(peers.values()), peers=peers, device=torch.cuda.get_device_name(dev))
    # 2. communicator
    box = [mnv.comm_get_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    try:
        comm = mnv.Comm(box[0], world, rank)
        item("communicator", True, rccl_version=mnv.rccl_version())
    except mnv.MnvError as e:
        item("communicator", False, error=str(e))
        comm = None
    if comm is not None:
        # 3. the tile gather at the two message sizes of the bench (per-rank share of a 64-frame RGBA8 batch is larger; these are per frame)
        for label, nbytes in (("gather 4.1 MB", 1920 * 1080 * 4 // 2), ("gather 16.6 MB", 3840 * 2160 * 4 // 2)):
            local = torch.full((nbytes,), rank + 1, dtype=torch.uint8, device=dev)
            gathered = torch.zeros((world, nbytes), dtype=torch.uint8, device=dev) if rank == 0 else None
            comm.gather_tiles(local, gathered)   # creates the channels
            torch.cuda.synchronize(dev)
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                comm.gather_tiles(local, gathered)
            e1.record()
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / 5
            ok = True
            if rank == 0:
                ok = all(bool((gathered[r] == r + 1).all().item()) for r in range(world))
            item(label, ok, ms=round(ms, 4), gb_per_s_into_root=round(nbytes * max(world - 1, 1) / (ms * 1e-3) / 1e9, 1) if rank == 0 else None)
        # 4. all-gather between all pairs (tracker rows of a refinement frame: 2 x 12 B per pixel of the rank's share)
        nb = 1920 * 1080 * 12 // world
        table = torch.zeros((world, nb), dtype=torch.uint8, device=dev)
        table[rank].fill_(rank + 1)
        comm.allgather(table)
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        comm.allgather(table)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) * 1e3
        item("all-gather", all(bool((table[r] == r + 1).all().item()) for r in range(world)), ms=round(ms, 4), bytes_per_rank=nb)
    # 5. CU-masked stream under N processes
    try:
        handle, enabled = mnv.stream_create_reserved(32)
        total = torch.cuda.get_device_properties(dev).multi_processor_count
        st = torch.cuda.ExternalStream(handle, device=dev)
        with torch.cuda.stream(st):
            x = torch.ones(1 << 20, device=dev)
            y = (x * 2).sum()
        torch.cuda.synchronize(dev)
        item("masked stream", enabled < total and float(y.item()) == float(2 << 20), enabled_cus=int(enabled), device_cus=int(total))
    except mnv.MnvError as e:
        item("masked stream", False, error=str(e), note="bench.py and mnv_render fall back to an unmasked march stream: the gather then waits for each march to drain")
    dist.barrier()
    if comm is not None:
        comm.close()
    dist.destroy_process_group()
    json.dump(res, open(out_path, "w"))


def gpu_count():
    import bench

    return bench.gpu_count_without_hip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="ranks to start (default: every GPU of the node)")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--worker", nargs=4, metavar=("RANK", "WORLD", "PORT", "OUT"))
    args = ap.parse_args()
    if args.worker:
        worker(int(args.worker[0]), int(args.worker[1]), int(args.worker[2]), args.worker[3])
        return 0
    n = args.gpus or gpu_count() or 1
    import signal
    import tempfile
    import time

    deadline_s = 600   # ONE deadline for the whole run: a wedged rank takes its peers down with it instead of costing 600 s each
    with tempfile.TemporaryDirectory() as d:
        results = None
        for attempt in range(3):   # the rendezvous port is free when probed, not necessarily when the workers bind it: retry on a lost race
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            logs = [open(os.path.join(d, f"r{r}.log"), "w") for r in range(n)]   # files, not pipes: a chatty rank (NCCL_DEBUG) cannot block on a full pipe
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(n), str(port), os.path.join(d, f"r{r}.json")],
                                      stdout=logs[r], stderr=subprocess.STDOUT, start_new_session=True) for r in range(n)]
            t_end, failed = time.time() + deadline_s, None
            while any(p.poll() is None for p in procs):
                bad = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
                if bad or time.time() > t_end:
                    failed = f"rank(s) {bad} exited with an error" if bad else f"no result within {deadline_s} s"
                    for p in procs:   # every worker leads its own session: kill exactly the groups started above
                        if p.poll() is None:
                            try:
                                os.killpg(p.pid, signal.SIGKILL)
                            except ProcessLookupError:
                                pass
                    break
                time.sleep(0.2)
            for p in procs:
                p.wait()
            for f in logs:
                f.close()
            outs = [open(os.path.join(d, f"r{r}.log")).read() for r in range(n)]
            if failed and attempt < 2 and any("EADDRINUSE" in o or "Address already in use" in o for o in outs):
                continue
            results = []
            for r in range(n):
                f = os.path.join(d, f"r{r}.json")
                results.append(json.load(open(f)) if os.path.exists(f) else
                               {"rank": r, "items": [{"name": "rank finished", "ok": False, "why": failed, "output": outs[r][-1500:]}]})
            break
    ok = all(it["ok"] for res in results for it in res["items"])
    if args.json:
        print(json.dumps({"ranks": n, "ok": ok, "results": results}))
    else:
        print(f"scale preflight on {n} rank(s):")
        names = []
        for res in results:
            for it in res["items"]:
                if it["name"] not in names:
                    names.append(it["name"])
        for name in names:
            its = [(res["rank"], it) for res in results for it in res["items"] if it["name"] == name]
            good = all(it["ok"] for _, it in its)
            detail = "; ".join(f"rank {r}: " + ", ".join(f"{k}={v}" for k, v in it.items() if k not in ("name", "ok") and v is not None) for r, it in its[:8])
            print(f"  [{'PASS' if good else 'FAIL'}] {name:16s} {detail}")
        print(NEXT_STEPS.format(n=n))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

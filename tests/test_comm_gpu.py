"""The RCCL tile gather of the C ABI (mnv_comm_*, mnv_gather_tiles) with the one rank a one-GPU box allows:
the root's share travels through ncclSend / ncclRecv to itself, on a caller stream, and lands in the gather table; followed by
mnv_assemble_tiles it reproduces the frame.  World > 1 index arithmetic is covered by the gloo tests and the partition tests."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def test_rccl_gather_single_rank_roundtrip(mnv, orc, torch_gpu):
    torch = torch_gpu
    assert mnv.rccl_version() > 20000          # a real RCCL was bound (2.2x.y -> 22xyy)
    comm = mnv.Comm(mnv.comm_get_unique_id(), 1, 0)
    try:
        st = torch.cuda.Stream()
        for dtype, shape in ((torch.uint8, (3, 7, 24, 64, 4)), (torch.float32, (2, 5, 24, 64, 4))):
            g = torch.Generator(device="cuda").manual_seed(3)
            local = (torch.rand(shape, device="cuda", generator=g) * 255).to(dtype)
            table = torch.zeros((1,) + shape, dtype=dtype, device="cuda")
            st.wait_stream(torch.cuda.current_stream())
            comm.gather_tiles(local, table, root=0, stream=st.cuda_stream)
            st.synchronize()
            assert torch.equal(table[0], local)
        with pytest.raises(mnv.MnvError):
            comm.gather_tiles(local, torch.zeros(5, device="cuda"), root=0)       # table of the wrong size
    finally:
        comm.close()


def test_partitioned_render_gather_assemble_equals_the_oracle(mnv, orc, torch_gpu):
    """One rank's whole multi-GPU step through the C ABI: partitioned batched march -> mnv_gather_tiles -> mnv_assemble_tiles."""
    torch = torch_gpu
    from mega_nerf_viewer_amd.multigpu import TileGatherer, TilePartition

    spec = cases.CASES["sh9_d7_aniso"]
    tree = cases.make_tree(mnv, spec["tree"])
    ot = orc.tree_from_view(tree.host_view())
    tree.move_to_device()
    w, h = 400, 248
    cams = [cases.make_camera(mnv, dict(spec["camera"], width=w, height=h)) for _ in range(3)]
    opt = cases.make_options(mnv, spec["options"])
    part = TilePartition(w, h, 1, 64, 24, 0)
    comm = mnv.Comm(mnv.comm_get_unique_id(), 1, 0)
    try:
        tg = TileGatherer(part, 0, torch.device("cuda", 0), dtype=torch.float32, depth=2, frames=len(cams), comm=comm)
        for slot in (0, 1, 0):
            tg.finish(slot)
            mnv.render_voxels_accel_batch(tree.accel, cams, opt, part=part.part(0), rgba=tg.local(slot),
                                          stream=torch.cuda.current_stream().cuda_stream)
            tg.submit(slot)
        tg.finish_all()
        torch.cuda.synchronize()
        ref = orc.render(ot, cams[0].c, opt)["rgba"]
        for slot in (0, 1):
            for f in range(len(cams)):
                assert np.array_equal(tg.frame(slot)[f].cpu().numpy().view(np.uint32), ref.view(np.uint32)), (slot, f)
    finally:
        comm.close()


def _rank_main(rank, world, lib, idq, resq):
    """One rank of the multi-process gather test (ranks share cuda:0; the transport is tests/shim/fake_rccl.cpp)."""
    import os
    import sys
    import traceback

    try:
        os.environ["MNV_RCCL_LIBRARY"] = lib
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.environ["MNV_LIB_PATH"] = os.path.join(root, "mega-nerf-viewer_amd", "testhooks", "libmnv.so")  # the build that honours MNV_RCCL_LIBRARY (tests/hooks.py)
        for p in (root, os.path.join(root, "tests"), os.path.join(root, "oracle")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch

        import cases as cs
        import mega_nerf_viewer_amd as mnv
        from mega_nerf_viewer_amd.multigpu import TileGatherer, TilePartition

        torch.cuda.set_device(0)
        if rank == 0:
            uid = mnv.comm_get_unique_id()
            for _ in range(world - 1):
                idq.put(uid)
        else:
            uid = idq.get(timeout=120)
        comm = mnv.Comm(uid, world, rank)
        assert mnv.rccl_version() == 29999
        spec = cs.CASES["sh9_d7_aniso"]
        tree = cs.make_tree(mnv, spec["tree"])
        tree.move_to_device()
        w, h = 400, 248
        cams = [cs.make_camera(mnv, dict(spec["camera"], width=w, height=h, center=(-3.0 + 0.2 * k, 2.0, 5.0))) for k in range(3)]
        opt = cs.make_options(mnv, spec["options"])
        part = TilePartition(w, h, world, 64, 24, 3)
        tg = TileGatherer(part, rank, torch.device("cuda", 0), dtype=torch.uint8, depth=2, frames=len(cams), comm=comm)
        for slot in (0, 1, 0):
            tg.finish(slot)
            if part.local_tiles(rank) > 0:
                mnv.render_voxels_accel_batch(tree.accel, cams, opt, part=part.part(rank), rgba8=tg.local(slot), stream=torch.cuda.current_stream().cuda_stream)
            tg.submit(slot)
        tg.finish_all()
        torch.cuda.synchronize()
        if rank == 0:
            full = torch.empty((len(cams), h, w, 4), dtype=torch.uint8, device="cuda")
            mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba8=full)
            torch.cuda.synchronize()
            resq.put(("ok", bool(torch.equal(tg.frame(0), full)) and bool(torch.equal(tg.frame(1), full))))
        comm.close()
    except Exception:  # noqa: BLE001
        resq.put(("error", f"rank {rank}: {traceback.format_exc()}"))


@pytest.mark.parametrize("world", [2, 4])
def test_tile_gatherer_over_the_c_abi_with_several_ranks_on_one_gpu(mnv, torch_gpu, fake_rccl, world):
    """multigpu.TileGatherer with an mnv.Comm and world > 1: every rank a process of its own on cuda:0, mnv_gather_tiles through the
    transport stand-in, un-permute on rank 0, root-relieving deal (root_period 3): assembled frames == the single-launch frames."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    idq, resq = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, fake_rccl, idq, resq)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        kind, val = resq.get(timeout=600)
        assert kind == "ok" and val is True, val
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()   # exactly the processes started above
    assert all(p.exitcode == 0 for p in procs)


def _late_rank_main(rank, world, lib, idq, resq):
    """Six steps through a ring of two slots, every step with its own cameras; rank 1 stalls for 1.5 s (hundreds of step times) before
    steps 2 and 4.  Rank 0 keeps what every slot held when its turn came round again."""
    import os
    import sys
    import time
    import traceback

    try:
        os.environ["MNV_RCCL_LIBRARY"] = lib
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.environ["MNV_LIB_PATH"] = os.path.join(root, "mega-nerf-viewer_amd", "testhooks", "libmnv.so")
        for p in (root, os.path.join(root, "tests"), os.path.join(root, "oracle")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch

        import cases as cs
        import mega_nerf_viewer_amd as mnv
        from mega_nerf_viewer_amd.multigpu import TileGatherer, TilePartition

        torch.cuda.set_device(0)
        if rank == 0:
            uid = mnv.comm_get_unique_id()
            for _ in range(world - 1):
                idq.put(uid)
        else:
            uid = idq.get(timeout=120)
        comm = mnv.Comm(uid, world, rank)
        spec = cs.CASES["sh9_d7_aniso"]
        tree = cs.make_tree(mnv, spec["tree"])
        tree.move_to_device()
        w, h, n_steps, depth = 400, 248, 6, 2
        cams_of = [[cs.make_camera(mnv, dict(spec["camera"], width=w, height=h, center=(-3.0 + 0.2 * k + 0.07 * s, 2.0 - 0.05 * s, 5.0))) for k in range(2)]
                   for s in range(n_steps)]
        opt = cs.make_options(mnv, spec["options"])
        part = TilePartition(w, h, world, 64, 24, 3)
        tg = TileGatherer(part, rank, torch.device("cuda", 0), dtype=torch.uint8, depth=depth, frames=2, comm=comm)
        kept = {}
        for s in range(n_steps):
            slot = s % depth
            if rank == 0 and s >= depth:
                tg.wait_frame(slot)
                kept[s - depth] = tg.frame(slot).clone()
            tg.finish(slot)
            if rank == 1 and s in (2, 4):
                torch.cuda.synchronize()
                time.sleep(1.5)
            mnv.render_voxels_accel_batch(tree.accel, cams_of[s], opt, part=part.part(rank), rgba8=tg.local(slot), stream=torch.cuda.current_stream().cuda_stream)
            tg.submit(slot)
        tg.finish_all()
        torch.cuda.synchronize()
        if rank == 0:
            for s in range(n_steps - depth, n_steps):
                kept[s] = tg.frame(s % depth).clone()
            bad = []
            full = torch.empty((2, h, w, 4), dtype=torch.uint8, device="cuda")
            for s in range(n_steps):
                mnv.render_voxels_accel_batch(tree.accel, cams_of[s], opt, rgba8=full)
                torch.cuda.synchronize()
                if not torch.equal(kept[s], full):
                    bad.append(s)
            distinct = not torch.equal(kept[0], kept[1])
            resq.put(("ok", (bad, distinct)))
        comm.close()
    except Exception:  # noqa: BLE001
        resq.put(("error", f"rank {rank}: {traceback.format_exc()}"))


def test_a_late_rank_cannot_hand_rank_0_a_half_written_slot(mnv, torch_gpu, fake_rccl):
    """World 2 over the transport stand-in, ring of two slots: rank 1 arrives more than a step late, twice.  Rank 0's march runs ahead
    by at most the ring's depth (finish(slot) orders the next render into a slot after the gather that read it), its gather waits for
    the peer on the slot's own side stream, and every step's assembled frames equal the single-launch frames of that step's cameras --
    the ordering `bench.py --gpus N` and mnv_render --gpus N rely on."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    idq, resq = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_late_rank_main, args=(r, 2, fake_rccl, idq, resq)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        kind, val = resq.get(timeout=600)
        assert kind == "ok", val
        bad, distinct = val
        assert bad == [] and distinct, val
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()   # exactly the processes started above
    assert all(p.exitcode == 0 for p in procs)

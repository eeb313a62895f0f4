"""N > 1 path on CPU: world_size 2 and 3 over gloo.  Every rank fills its compact local-tile-major
buffer (here with the CPU oracle standing in for the HIP launch -- the GPU version of the same
check is tests/test_parity_gpu.py::test_cfg2_interleaved_partition_reassembles_bit_exact), the
buffers are gathered to rank 0 with the production TileGatherer and un-permuted; rank 0 compares
with the full-frame render bit for bit.  Also checks the pure index math against the C ABI."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tile_w, tile_h, n_frames, root_period, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import mega_nerf_viewer_amd as mnv
        import mnv_oracle as orc
        from mega_nerf_viewer_amd.multigpu import TileGatherer, TilePartition

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        spec = cases.CASES["sh4_d6"]
        tree = cases.make_tree(mnv, spec["tree"])
        opt = cases.make_options(mnv, spec["options"])
        ot = orc.tree_from_view(tree.host_view())
        cams = []
        for k in range(n_frames):
            cs = dict(spec["camera"])
            cs["center"] = (cs["center"][0] + 0.1 * k, cs["center"][1], cs["center"][2])
            cams.append(cases.make_camera(mnv, cs))
        W, H = cams[0].width, cams[0].height
        part = TilePartition(W, H, world, tile_w, tile_h, root_period)
        assert part.local_tiles(rank) == mnv.partition_local_tiles((0, 0, W, H), rank, world, tile_w, tile_h, root_period)
        tg = TileGatherer(part, rank, "cpu", depth=2)
        ok = True
        for k, cam in enumerate(cams):
            slot = k % 2
            tg.finish(slot)
            buf = tg.local(slot)
            buf.fill_(float("nan"))
            for j, m in enumerate(part.tiles_of(rank)):
                x0, y0, w, h = part.tile_rect(m)
                buf[j, :h, :w] = torch.from_numpy(orc.render(ot, cam.c, opt, tile=(x0, y0, w, h))["rgba"])
            tg.submit(slot)
            if k >= 1:  # frame k-1 overlapped with the render of frame k
                tg.finish(1 - slot)
                if rank == 0:
                    full = orc.render(ot, cams[k - 1].c, opt)["rgba"]
                    ok &= bool(np.array_equal(tg.frame(1 - slot).numpy().view(np.uint32), full.view(np.uint32)))
        tg.finish_all()
        if rank == 0:
            full = orc.render(ot, cams[-1].c, opt)["rgba"]
            ok &= bool(np.array_equal(tg.frame((n_frames - 1) % 2).numpy().view(np.uint32), full.view(np.uint32)))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, ok, ""))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


@pytest.mark.parametrize("world,tile,root_period", [(2, (64, 40), 0), (3, (48, 56), 0), (2, (16, 8), 3)])
def test_gather_and_unpermute_over_gloo(mnv, orc, world, tile, root_period):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tile[0], tile[1], 3, root_period, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in results:
        assert ok, f"rank {rank}: {err}"


def test_unpermute_with_frame_dimension(mnv):
    from mega_nerf_viewer_amd.multigpu import TilePartition

    part = TilePartition(100, 60, 3, 24, 16)
    g = torch.arange(3 * 2 * part.j_max * 16 * 24 * 4, dtype=torch.float32).reshape(3, 2, part.j_max, 16, 24, 4)
    both = part.unpermute(g)
    assert both.shape == (2, 60, 100, 4)
    for f in range(2):
        assert torch.equal(both[f], part.unpermute(g[:, f].contiguous()))


def test_partition_index_math_matches_c_abi(mnv):
    from mega_nerf_viewer_amd.multigpu import TilePartition

    for (W, H, world, tw, th, M) in [(1920, 1080, 8, 64, 24, 0), (1920, 1080, 8, 64, 24, 8), (1920, 1080, 2, 64, 24, 32), (1920, 1080, 8, 128, 120, 0),
                                     (1920, 1080, 3, 200, 136, 2), (100, 50, 4, 8, 8, 3), (7, 5, 2, 8, 8, 2), (1920, 1080, 1, 64, 24, 5)]:
        part = TilePartition(W, H, world, tw, th, M)
        seen = []
        for r in range(world):
            assert part.local_tiles(r) == mnv.partition_local_tiles((0, 0, W, H), r, world, tw, th, M)
            tiles = part.tiles_of(r)
            assert [part.owner(m) for m in tiles] == [(r, j) for j in range(len(tiles))]   # local order = increasing macro tile number
            seen += tiles
        assert sorted(seen) == list(range(part.n_macro))
        if world > 1 and M >= 2 and part.n_macro >= 4 * world * M:  # rank 0 is left out of every M-th round of the deal
            assert part.local_tiles(0) < part.local_tiles(1) and abs(part.local_tiles(0) / part.local_tiles(1) - (M - 1) / M) < 0.08
        # un-permute of a synthetic gathered table puts every macro tile where its rect says
        g = torch.zeros((world, part.j_max, th, tw, 1))
        for r in range(world):
            for j, m in enumerate(part.tiles_of(r)):
                g[r, j] = float(m + 1)
        frame = part.unpermute(g)
        assert frame.shape == (H, W, 1)
        for m in range(part.n_macro):
            x0, y0, w, h = part.tile_rect(m)
            assert torch.all(frame[y0:y0 + h, x0:x0 + w] == m + 1)

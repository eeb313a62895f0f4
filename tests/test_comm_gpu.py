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

"""What the batched march is short of: the headline launch (cfg2, 64 frames at 1920x1080 in one launch) and one frame per launch on streams that may
use a half / a quarter of the compute units (hipExtStreamCreateWithCUMask; only contiguous runs of mask bits are honoured).  A launch bound
by what a compute unit can keep in flight takes 2x / 4x the time; one that is short of something the chip shares (fabric, HBM, power) takes less.
python3 tools/march_cu_mask.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402

W, H = 1920, 1080
tree = cases.make_tree(mnv, cases.CFG2_TREE)
tree.move_to_device()
opt = mnv.RenderOptions.cli_defaults()
opt.basis_minmax[1] = 8
cams = [cases.cfg2_camera(mnv, p % 16, W, H, 1600.0) for p in range(64)]
out = torch.empty((64, H, W, 4), device="cuda")
hip = mnv._hip()
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words) == 0
    return s.value


def timed(fn, stream, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1, ms = C.c_void_p(), C.c_void_p(), C.c_float()
    assert hip.hipEventCreate(C.byref(e0)) == 0 and hip.hipEventCreate(C.byref(e1)) == 0
    best = 1e9
    for _ in range(reps):
        hip.hipEventRecord(e0, C.c_void_p(stream))
        fn()
        hip.hipEventRecord(e1, C.c_void_p(stream))
        hip.hipEventSynchronize(e1)
        hip.hipEventElapsedTime(C.byref(ms), e0, e1)
        best = min(best, ms.value)
    return best


base = {}
for name, bits in (("256 CUs", (1 << 256) - 1), ("128 CUs (low half of the mask)", (1 << 128) - 1), ("64 CUs (low quarter)", (1 << 64) - 1)):
    st = masked_stream(bits)
    n = bin(bits).count("1")
    batch = timed(lambda: mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba=out, stream=st), st, 4)
    one = timed(lambda: mnv.render_voxels_accel(tree.accel, cams[5], opt, rgba=out[0], stream=st), st, 8)
    base.setdefault("batch", batch), base.setdefault("one", one)
    print(f"{name:32s} 64 frames in one launch {batch:8.3f} ms = {64 * W * H / batch / 1e3:7.0f} Mrays/s (x{batch / base['batch']:5.2f} time for x{256 / n:3.0f} fewer CUs)"
          f"   one frame {one:6.3f} ms (x{one / base['one']:5.2f})")

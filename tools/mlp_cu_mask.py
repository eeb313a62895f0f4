"""Is mlp_forward_kernel held back by the chip's power / clock management rather than by its own instruction stream?
The same 8 M samples on streams that may use 256 / 128 / 64 / 32 of the compute units (hipExtStreamCreateWithCUMask): if the time grows by less
than the factor the CUs shrink by, a CU is faster when fewer of them draw power.  python3 tools/mlp_cu_mask.py [w128|w64]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import mega_nerf_viewer_amd as mnv  # noqa: E402
import mlp_cases  # noqa: E402

kw = dict(hidden_width=128, hidden_layers=4, out_dim=29, pos_octaves=10, dir_octaves=4, need_viewdir=True)
if len(sys.argv) > 1 and sys.argv[1] == "w64":
    kw = dict(hidden_width=64, hidden_layers=2, out_dim=29, pos_octaves=10, dir_octaves=4, need_viewdir=True)
desc = mnv.mlp_desc(n_clusters=8, **kw)
mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=2))
m = 8_000_000
x = torch.rand((m, 6), device="cuda") * 2 - 1
cl = torch.randint(0, 8, (m,), device="cuda", dtype=torch.int16)
res = torch.empty((m, desc.out_dim), device="cuda")
hip = mnv._hip()
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return s.value


def timed(stream, reps=4):
    for _ in range(2):
        mlp.query(cl, x, res, stream=stream)
    torch.cuda.synchronize()
    e0, e1, ms = C.c_void_p(), C.c_void_p(), C.c_float()
    assert hip.hipEventCreate(C.byref(e0)) == 0 and hip.hipEventCreate(C.byref(e1)) == 0
    t = []
    for _ in range(reps):
        hip.hipEventRecord(e0, C.c_void_p(stream))
        mlp.query(cl, x, res, stream=stream)
        hip.hipEventRecord(e1, C.c_void_p(stream))
        hip.hipEventSynchronize(e1)
        hip.hipEventElapsedTime(C.byref(ms), e0, e1)
        t.append(ms.value)
    return min(t), sum(t) / len(t)


full = (1 << 256) - 1
masks = {
    "256 CUs": full,
    "128 CUs (low half of the mask)": (1 << 128) - 1,
    "64 CUs (low quarter of the mask)": (1 << 64) - 1,
    "32 CUs (low eighth of the mask)": (1 << 32) - 1,
    "128 CUs (every second bit)": int("01" * 128, 2),
    "64 CUs (every fourth bit)": int("0001" * 64, 2),
    "32 CUs (every eighth bit)": int("00000001" * 32, 2),
}
base = None
for name, bits in masks.items():
    best, mean = timed(masked_stream(bits))
    n = bin(bits).count("1")
    base = base or best
    print(f"{name:34s} {best:7.3f} ms (mean {mean:7.3f})   x{best / base:5.2f} time for x{256 / n:4.1f} fewer CUs   per-CU rate {base * 256 / (best * n):5.2f} of the full chip's")

"""Which compute unit does bit k of a hipExtStreamCreateWithCUMask mask enable?  One single-bit stream per k, a few workgroups
each reporting XCC_ID / HW_ID.  Prints the (xcd, se, sh, cu) of every bit and per-XCD unit counts of 'low n bits' masks."""
import ctypes as C
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
import torch  # noqa: E402

spin = C.CDLL(os.path.join(HERE, "libspin.so"))
spin.whoami_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
spin.masked_stream_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_uint32)]
spin.stream_destroy.argtypes = [C.c_void_p]
n_cus = torch.cuda.get_device_properties(0).multi_processor_count
words = (n_cus + 31) // 32


def run(bits, n_blocks, usec):
    arr = (C.c_uint32 * words)(*[(bits >> (32 * i)) & 0xffffffff for i in range(words)])
    h = C.c_void_p()
    rc = spin.masked_stream_create(C.byref(h), words, arr)
    assert rc == 0, rc
    out = torch.zeros(n_blocks, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert spin.whoami_launch(h, n_blocks, usec, out.data_ptr()) == 0
    torch.cuda.synchronize()
    spin.stream_destroy(h)
    return out.cpu().numpy().astype("uint32")


def decode(v):
    return (int(v >> 16) & 15, int(v >> 13) & 7, int(v >> 12) & 1, int(v >> 8) & 15)


bitmap = {}
for k in range(n_cus):
    vals = set(decode(v) for v in run(1 << k, 8, 5))
    bitmap[k] = sorted(vals)
for k in range(0, n_cus, 8):
    print(k, [bitmap[j] for j in range(k, k + 8)])
for n in (248, 240, 224, 192):
    vals = run((1 << n) - 1, 8192, 20)
    units = set(decode(v) for v in vals)
    per = {}
    for u in units:
        per[u[0]] = per.get(u[0], 0) + 1
    print("low", n, "bits ->", len(units), "units; per XCD", dict(sorted(per.items())))
json.dump({str(k): v for k, v in bitmap.items()}, open(os.path.join(HERE, "..", "..", "gpurun_out", "cumask_bitmap.json"), "w"))

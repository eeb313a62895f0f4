"""CU-mask experiments on ONE GPU: can the march leave compute units free for the gather's RCCL kernels?

1. march time of the bench step on streams masked to 256 - R units (low bits set: on gfx942/gfx950 mask bit i belongs to XCD i % 8,
   so clearing the top R bits removes R / 8 units from every XCD);
2. concurrency: a stand-in for RCCL's channel workgroups (tools/cumask/spin.hip: 256 threads holding 100 KB of LDS, which like
   RCCL's 288-VGPR waves cannot sit beside a full set of march workgroups) launched on a second stream while the march runs --
   on the unmasked stream it has to wait for the persistent workgroups to exit, next to a masked march it runs at once.
"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
HERE = os.path.dirname(os.path.abspath(__file__))

import torch  # noqa: E402

import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402


def main():
    so = os.path.join(HERE, "libspin.so")
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(HERE, "spin.hip"), "-o", so])
    spin = C.CDLL(so)
    spin.spin_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    spin.masked_stream_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_uint32)]
    dev = torch.device("cuda", 0)
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    W, H, FX, NF = 1920, 1080, 1600.0, 64
    tree = cases.make_tree(mnv, cases.CFG2_TREE)
    tree.move_to_device()
    opt = mnv.RenderOptions.cli_defaults()
    cams = [cases.cfg2_camera(mnv, p % 16, W, H, FX) for p in range(NF)]
    out = torch.empty((NF, H, W, 4), dtype=torch.uint8, device=dev)
    sink = torch.zeros(4, dtype=torch.int32, device=dev)

    def masked(reserve):
        words = (n_cus + 31) // 32
        bits = (1 << (n_cus - reserve)) - 1
        arr = (C.c_uint32 * words)(*[(bits >> (32 * i)) & 0xffffffff for i in range(words)])
        h = C.c_void_p()
        rc = spin.masked_stream_create(C.byref(h), words, arr)
        if rc != 0:
            raise SystemExit(f"hipExtStreamCreateWithCUMask failed: {rc}")
        return torch.cuda.ExternalStream(h.value, device=dev)

    side = torch.cuda.Stream(device=dev)
    res = {"device_cus": n_cus, "march": {}, "overlap": {}}
    for reserve in (0, 8, 16, 32, 64):
        st = masked(reserve) if reserve else torch.cuda.Stream(device=dev)
        mnv.accel_set_cu_budget(tree.accel, n_cus - reserve)

        def march():
            mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba8=out, stream=st.cuda_stream)

        for _ in range(2):
            march()
        st.synchronize()
        mnv.set_timing(True)
        for _ in range(5):
            march()
        st.synchronize()
        ms, n = mnv.take_timing()
        mnv.set_timing(False)
        res["march"][reserve] = round(ms / n, 4)
        # a 1 ms spin of 16 workgroups started right after the march was launched: when does it finish?
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        march()
        time.sleep(0.003)  # the march is running
        e0.record(side)
        rc = spin.spin_launch(side.cuda_stream, 16, 100 * 1024, 1000, sink.data_ptr())
        assert rc == 0, rc
        e1.record(side)
        side.synchronize()
        t_spin_done = time.perf_counter() - t0
        st.synchronize()
        t_all = time.perf_counter() - t0
        res["overlap"][reserve] = {"spin_event_ms": round(e0.elapsed_time(e1), 3), "spin_done_after_ms": round(t_spin_done * 1e3, 3),
                                   "march_done_after_ms": round(t_all * 1e3, 3)}
        print(reserve, res["march"][reserve], res["overlap"][reserve], flush=True)
    mnv.accel_set_cu_budget(tree.accel, 0)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

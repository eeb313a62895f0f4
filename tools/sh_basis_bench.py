"""Batched march (16 poses, 1080p) on the cfg2 shell tree with other SH orders: python3 tools/sh_basis_bench.py [basis ...]
(MNV_LIB_PATH selects a variant library; symbols a variant lacks are skipped)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402

h = C.CDLL(mnv.LIB_PATH)
for name in list(mnv._SIGNATURES):
    if not hasattr(h, name):
        del mnv._SIGNATURES[name]
W, H = 1920, 1080
for basis in [int(a) for a in sys.argv[1:]] or [4, 9, 16, 25]:
    tree = cases.make_tree(mnv, dict(cases.CFG2_TREE, basis_dim=basis, depth=10 if basis <= 9 else 9))
    tree.move_to_device()
    opt = mnv.RenderOptions.cli_defaults()
    opt.basis_minmax[1] = basis - 1
    cams = [cases.cfg2_camera(mnv, p, W, H, 1600.0) for p in range(16)]
    out = torch.empty((16, H, W, 4), dtype=torch.float32, device="cuda")
    for _ in range(2):
        mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        mnv.render_voxels_accel_batch(tree.accel, cams, opt, rgba=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"SH{basis}: {ms:.3f} ms per 16 frames, {16 * W * H / ms / 1e3:.0f} Mrays/s, checksum {float(out.double().sum()):.6f}")
    del tree, out

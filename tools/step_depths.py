"""Where do a workload's march steps land?  Histogram of steps by leaf depth, empty | dense, from the CPU oracle (analysis hook
orc_set_depth_histogram; CPU only, no GPU needed).   usage: python tools/step_depths.py cfg2|cfg3 [pose] [width height]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402
import mnv_oracle as orc  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
pose = int(sys.argv[2]) if len(sys.argv) > 2 else 0
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (960, 540)
if wl == "cfg2":
    tree, cam = cases.make_tree(mnv, cases.CFG2_TREE), cases.cfg2_camera(mnv, pose, w, h, 1600.0 * w / 1920)
else:
    tree, cam = cases.make_tree(mnv, cases.CFG3_FULL), cases.cfg3_camera(mnv, pose, w, h, 1400.0 * w / 1920)
opt = mnv.RenderOptions.cli_defaults()
hist = np.zeros((2, 32), np.uint64)
orc.lib().orc_set_depth_histogram(C.c_void_p(hist.ctypes.data))
r = orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt)
orc.lib().orc_set_depth_histogram(None)
rays = w * h
print(f"{wl} pose {pose} {w}x{h}: {tree.capacity} chunks, steps/ray {hist.sum() / rays:.2f}, dense/ray {hist[1].sum() / rays:.2f}")
for d in range(32):
    if hist[0, d] or hist[1, d]:
        print(f"  leaf depth {d:2d}: empty {hist[0, d] / rays:7.3f} per ray, dense {hist[1, d] / rays:7.3f} per ray")

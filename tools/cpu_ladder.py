"""The CPU baseline (the oracle's march, plain C + OpenMP) at 1 / 2 / 4 / ... / all physical cores on the cfg2 frame bench.py times: Mrays/s and
Mrays/s per core at every rung, with and without the first-touch copy of the tree -- what SURVEY.md 8(d)'s single-core measurement
(0.27 Mrays/s on one 2.1 GHz Xeon core) is to be compared with, and where the per-core rate of the full machine goes (memory latency under
load, not scheduling).  No GPU needed.   usage: python tools/cpu_ladder.py [poses per rung, default 2]"""
import os
import sys
import time

if hasattr(os, "sched_getaffinity"):
    os.environ.setdefault("MNV_ORACLE_CPUS", str(len(os.sched_getaffinity(0))))   # before an OpenMP runtime binds the main thread to one place
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402
import mnv_oracle as orc  # noqa: E402

n_poses = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tree = cases.make_tree(mnv, cases.CFG2_TREE)
ot = orc.tree_from_view(tree.host_view())
opt = mnv.RenderOptions.cli_defaults()
phys, hw = orc.physical_cores()
phys = phys or hw
rungs = sorted({t for t in (1, 2, 4, 8, 16, 32, 64, 128, 256) if t < phys} | {phys})
print(f"cgroup CPU quota: {orc.cpu_quota()} CPUs; bench.py's baseline would use {orc.baseline_threads()[0]} threads")
print(f"host: {phys} physical cores, {hw} hardware threads; cfg2 tree {tree.capacity} chunks; {n_poses} pose(s) of 1920x1080 per rung")
for copy in (False, True):
    for t in rungs:
        w, h = (1920, 1080) if t >= 8 else (960, 540)  # the low rungs march a quarter frame (same rays per pixel footprint, a quarter of the time)
        tr = orc.copy_first_touch(ot, t) if copy else ot
        t0 = time.perf_counter()
        for pose in range(n_poses):
            orc.render(tr, cases.cfg2_camera(mnv, pose, w, h, 1600.0 * w / 1920).c, opt, n_threads=t)
        el = time.perf_counter() - t0
        if copy:
            orc.free_copy(tr)
        rate = n_poses * w * h / el / 1e6
        print(f"{'first-touch copy' if copy else 'tree as built   '} threads {t:4d}: {rate:8.3f} Mrays/s  {rate / t:7.4f} per core", flush=True)

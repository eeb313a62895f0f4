"""Writes tests/golden/fog_counters.json: the integer work counters of the CPU oracle for each of the 16 poses of the FOG workload
(tests/cases.py::FOG_TREE under the cfg2 cameras: long dense runs).  bench.py turns them into the algorithmic bytes per frame of
SURVEY.md 8(d).  Deterministic: same tree generator, same cameras.  About half an hour on 8 cores."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402
import mnv_oracle as orc  # noqa: E402

tree = cases.make_tree(mnv, cases.FOG_TREE)
ot = orc.tree_from_view(tree.host_view())
opt = mnv.RenderOptions.cli_defaults()
out = {"workload": "fog: depth-9 SH9 thick shell (half-thickness 0.06, sigma U(5,40)), 1920x1080, fx 1600, orbit radius 2.6 elevation 20, CLI options",
       "capacity": tree.capacity, "poses": {}}
for pose in range(16):
    cam = cases.cfg2_camera(mnv, pose)
    c = orc.render(ot, cam.c, opt)["counters"].as_dict()
    c["algorithmic_bytes"] = orc.algorithmic_bytes(orc.OrcCounters(**{k: c[k] for k in c}), 1, 9)
    out["poses"][str(pose)] = c
    print(pose, c, flush=True)
    with open(os.path.join(HERE, "fog_counters.json"), "w") as f:
        json.dump(out, f, indent=1)

"""Writes tests/golden/cfg3_counters.json: the integer work counters of the CPU oracle for each of the 16 cfg3 poses (BASELINE.json
configs[2]: the 7.2 M-chunk depth-11 anisotropic terrain, cases.CFG3_FULL) at 1920x1080 and, under "cfg4", at 3840x2160
(configs[3] on one GPU).  bench.py turns them into the algorithmic bytes per frame of SURVEY.md 8(d) and re-derives two poses in every
run.  Deterministic: same tree generator, same cameras.  usage: make_cfg3_counters.py [cfg3|cfg4|both]"""
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

which = sys.argv[1] if len(sys.argv) > 1 else "both"
path = os.path.join(HERE, "cfg3_counters.json")
tree = cases.make_tree(mnv, cases.CFG3_FULL)
ot = orc.tree_from_view(tree.host_view())
opt = mnv.RenderOptions.cli_defaults()
out = json.load(open(path)) if os.path.exists(path) else {}
out["workload"] = "cfg3 depth-11 SH9 anisotropic terrain (cases.CFG3_FULL), oblique orbit (cases.cfg3_camera), fx 1400 x width / 1920, CLI options"
out["capacity"] = tree.capacity
for name, (w, h) in (("cfg3", (1920, 1080)), ("cfg4", (3840, 2160))):
    if which not in (name, "both"):
        continue
    sec = {"resolution": f"{w}x{h}", "poses": {}}
    for pose in range(16):
        cam = cases.cfg3_camera(mnv, pose, w, h, fx=1400.0 * w / 1920)
        c = orc.render(ot, cam.c, opt)["counters"].as_dict()
        c["algorithmic_bytes"] = orc.algorithmic_bytes(orc.OrcCounters(**{k: c[k] for k in c}), 1, 9)
        sec["poses"][str(pose)] = c
        print(name, pose, c, flush=True)
    out[name] = sec
    with open(path, "w") as f:
        json.dump(out, f, indent=1)

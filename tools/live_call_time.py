"""bench.py's `live_call` object alone (the reference's literal per-frame call on three servers, checked against the oracle): python tools/live_call_time.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch, cases, bench, mega_nerf_viewer_amd as mnv, mnv_oracle as orc
tree = cases.make_tree(mnv, cases.CFG2_TREE); tree.move_to_device()
cams = [cases.cfg2_camera(mnv, p, bench.W, bench.H, bench.FX) for p in range(16)]
r = bench.extras_live_call(mnv, cases, orc, torch, torch.device("cuda", 0), tree, cams, mnv.RenderOptions.cli_defaults())
r.pop("what"); print(json.dumps(r))

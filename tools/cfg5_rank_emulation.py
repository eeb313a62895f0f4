"""What one rank of eight does in a refinement frame (BASELINE.json configs[4]), timed on one GPU: the partitioned tracker march of every
rank against the whole-frame march, cfg2 tree at 1920x1080, with and without guided sampling (fused kernel).  The rest of a frame (vote,
split networks, tree edit, accel patch: tools/cfg5_frame_time.py) runs replicated on every rank and does not shrink with the rank count."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402
import mlp_cases  # noqa: E402

W, H, WORLD, TW, TH = 1920, 1080, 8, 64, 24
tree = cases.make_tree(mnv, cases.CFG2_TREE)
v = tree.host_view()
tree.move_to_device(need_sample_counts=True)
dv = tree.device_view()
opt = mnv.RenderOptions.cli_defaults()
opt.basis_minmax[1] = 8
opt.max_guided_samples = 32
opt.max_depth, opt.max_sample_count = 12, 64
desc = mnv.mlp_desc(n_clusters=8, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=4))
g = mnv.ClusterGrid()
g.grid_dim[0], g.grid_dim[1] = 4, 2
for i in range(3):
    g.min_position[i], g.range[i] = -1.0, 2.0
j_max = max(mnv.partition_local_tiles((0, 0, W, H), r, WORLD, TW, TH, 0) for r in range(WORLD))
px = j_max * TW * TH
full = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
trk = [torch.full((H * W, 3), -1.0, dtype=torch.float32, device="cuda") for _ in range(2)]
local = torch.empty((px, 4), dtype=torch.float32, device="cuda")
ltrk = [torch.full((px, 3), -1.0, dtype=torch.float32, device="cuda") for _ in range(2)]
sc = torch.full((v.capacity, 8), 8, dtype=torch.int16, device="cuda")


def timed(fn, n=6):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = {}
for pose in (0, 5):
    cam = cases.cfg2_camera(mnv, pose, W, H, 1600.0)
    for label, whole, part in (
        ("trackers", lambda: mnv.render_voxels_accel_track(tree.accel, cam, opt, rgba=full, split_track=trk[0], sample_track=trk[1], sample_counts=sc),
         lambda r: mnv.render_voxels_accel_visit(tree.accel, cam, opt, None, None, rgba=local, split_track=ltrk[0], sample_track=ltrk[1], sample_counts=sc,
                                                 part=(r, WORLD, TW, TH, 0))),
        ("trackers + guided (fused)", lambda: mnv.render_guided_fused(tree.accel, cam, opt, mlp, g, rgba=full, split_track=trk[0], sample_track=trk[1], sample_counts=sc),
         lambda r: mnv.render_guided_fused_part(tree.accel, cam, opt, mlp, g, (r, WORLD, TW, TH, 0), rgba=local, split_track=ltrk[0], sample_track=ltrk[1],
                                                sample_counts=sc))):
        t_whole = timed(whole)
        t_ranks = [timed(lambda r=r: part(r)) for r in range(WORLD)]
        out[f"pose {pose}, {label}"] = {"whole_frame_ms": round(t_whole, 3), "rank_ms": [round(t, 3) for t in t_ranks], "slowest_rank_ms": round(max(t_ranks), 3),
                                      "ratio": round(t_whole / max(t_ranks), 2)}
import json
for k, val in out.items():
    print(k, json.dumps(val))

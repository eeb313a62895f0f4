"""Guided-sampling frame on the cfg2 tree at 1920x1080: the fused kernel (mnv_render_guided_fused) against the four-step path
(sample march on the accel -> mnv_compact_guided_samples -> mnv_query_submodules -> mnv_render_nerf_results), device time per
frame from HIP events, frames compared bit for bit.  usage: python3 tools/guided_bench.py [max_guided_samples] [pos_octaves]"""
import json
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

max_g = int(sys.argv[1]) if len(sys.argv) > 1 else 32
octaves = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W, H = 1920, 1080
tree = cases.make_tree(mnv, cases.CFG2_TREE)
v = tree.host_view()
tree.move_to_device()
opt = mnv.RenderOptions.cli_defaults()
opt.basis_minmax[1] = 8
opt.max_guided_samples = max_g
desc = mnv.mlp_desc(n_clusters=8, pos_octaves=octaves, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=4))
g = mnv.ClusterGrid()
g.grid_dim[0], g.grid_dim[1] = 4, 2
for i in range(3):
    g.min_position[i], g.range[i] = -1.0, 2.0
n_px, dim, dd = W * H, 4, v.data_dim
num = torch.zeros(n_px, dtype=torch.int16, device="cuda")
guided = torch.zeros((n_px, max_g, dim), dtype=torch.float32, device="cuda")
clusters = torch.zeros((n_px, max_g), dtype=torch.int16, device="cuda")
offsets = torch.empty(n_px, dtype=torch.int64, device="cuda")
cap = 24_000_000
z = torch.empty(cap, dtype=torch.float32, device="cuda")
rows = torch.empty((cap, dim - 1), dtype=torch.float32, device="cuda")
rcl = torch.empty(cap, dtype=torch.int16, device="cuda")
values = torch.empty((cap, dd + 1), dtype=torch.float32, device="cuda")
out_a = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
out_b = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
counter = torch.zeros(16, dtype=torch.int64, device="cuda")


def four_step(cam):
    num.zero_()
    mnv.get_samples_from_voxels_accel(tree.accel, cam, opt, num, guided, clusters, g)
    total = mnv.compact_guided_samples(num, guided, clusters, offsets, z, rows, rcl)
    mlp.query(rcl, rows, values, n=total)
    mnv.render_nerf_results(tree.device_view(), cam, opt, values, z, offsets, rgba=out_a)
    return total


def fused(cam):
    mnv.render_guided_fused(tree.accel, cam, opt, mlp, g, rgba=out_b, sample_counter=counter)


def timed(fn, cams, reps=3):
    for c in cams[:2]:
        fn(c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for c in cams:
            fn(c)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(cams))


cams = [cases.cfg2_camera(mnv, p, W, H, 1600.0) for p in range(8)]
res = {"max_guided_samples": max_g, "pos_octaves": octaves, "batch_min": os.environ.get("MNV_FUSED_BATCH_MIN", "64")}
res["four_step_ms"] = round(timed(four_step, cams), 4)
res["fused_ms"] = round(timed(fused, cams), 4)
total = four_step(cams[3])
counter.zero_()
fused(cams[3])
torch.cuda.synchronize()
res["samples"] = int(total)
res["fused_samples"] = int(counter[0].item())
if os.environ.get("MNV_FUSED_DIAG"):
    res["passes"], res["march_iters"] = int(counter[1].item()), int(counter[2].item())
    res["passes_cut_by_cluster"], res["drain_passes"] = int(counter[3].item()), int(counter[4].item())
    res["wave_us_network"], res["wave_us_total"], res["wave_us_layer0"] = [round(int(counter[i].item()) / 100.0 / 2048, 1) for i in (5, 6, 7)]
    res["wave_us_hidden_layers"], res["wave_us_column_eval"], res["wave_us_apply"] = [round(int(counter[i].item()) / 100.0 / 2048, 1) for i in (8, 9, 10)]
    res["us_per_run"] = round(int(counter[5].item()) / 100.0 / max(1, res["passes"]), 2)
    res["lanes_per_pass"] = round(res["fused_samples"] / max(1, res["passes"]), 2)
res["bit_identical"] = bool(torch.equal(out_a.view(torch.int32), out_b.view(torch.int32)))
print(json.dumps(res))

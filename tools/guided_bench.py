"""Guided-sampling frame on the cfg2 tree at 1920x1080: the fused kernel (mnv_render_guided_fused) against the four-step path
(sample march on the accel -> mnv_compact_guided_samples -> mnv_query_submodules -> mnv_render_nerf_results), device time per
frame from HIP events, frames compared bit for bit.  usage: python3 tools/guided_bench.py [max_guided_samples] [pos_octaves]"""
import os as _os
# the MNV_* knobs this tool reads exist in the test-hook build of the library only (csrc/mnv_knobs.h)
_os.environ.setdefault("MNV_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mega-nerf-viewer_amd", "testhooks", "libmnv.so"))
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
counter = torch.zeros(1, dtype=torch.int64, device="cuda")


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
total = four_step(cams[3])
res["samples"] = int(total)
diag = torch.zeros(32, dtype=torch.int64, device="cuda")
for version, name in ((2, "producer_consumer"), (1, "one_role")):
    mnv.set_fused_kernel(version)
    r = {"ms": round(timed(fused, cams), 4)}
    counter.zero_()
    out_b.fill_(float("nan"))
    fused(cams[3])
    torch.cuda.synchronize()
    r["samples"] = int(counter[0].item())
    r["bit_identical"] = bool(torch.equal(out_a.view(torch.int32), out_b.view(torch.int32)))
    if os.environ.get("MNV_FUSED_DIAG"):
        diag.zero_()
        mnv.set_fused_diag(diag)
        r["ms_with_diag"] = round(timed(fused, cams[3:4], reps=1), 4)
        diag.zero_()
        fused(cams[3])
        torch.cuda.synchronize()
        mnv.set_fused_diag(None)
        d = [int(x) for x in diag.tolist()]
        if version == 1:
            waves = 2048
            r.update(passes=d[1], march_iters=d[2], passes_cut_by_cluster=d[3], drain_passes=d[4],
                     wave_us_network=round(d[5] / 100.0 / waves, 1), wave_us_total=round(d[6] / 100.0 / waves, 1), wave_us_layer0=round(d[7] / 100.0 / waves, 1),
                     wave_us_hidden_layers=round(d[8] / 100.0 / waves, 1), wave_us_column_eval=round(d[9] / 100.0 / waves, 1), wave_us_apply=round(d[10] / 100.0 / waves, 1),
                     us_per_run=round(d[5] / 100.0 / max(1, d[1]), 2), lanes_per_pass=round(r["samples"] / max(1, d[1]), 2))
        else:
            names = ["-", "runs", "march_iters", "windows", "weight_reloads", "runs_from_l2", "cons_busy", "cons_total", "prod_total", "prod_ring_wait",
                     "prod_flush_wait", "encode_l0", "layers_total", "eval", "columns", "watchdog"]
            dd = dict(zip(names, d))
            r.update(runs=dd["runs"], weight_reloads=dd["weight_reloads"], us_per_reload=round(dd["runs_from_l2"] / 100.0 / max(1, dd["weight_reloads"]), 2), march_iters=dd["march_iters"],
                     columns_per_run=round(dd["columns"] / max(1, dd["runs"]), 2), watchdog=dd["watchdog"],
                     us_per_run=round(dd["cons_busy"] / 100.0 / max(1, dd["runs"]), 2),
                     us_per_run_encode_l0=round(dd["encode_l0"] / 100.0 / max(1, dd["runs"]), 2),
                     us_per_run_layers=round(dd["layers_total"] / 100.0 / max(1, dd["runs"]), 2),
                     us_per_run_eval=round(dd["eval"] / 100.0 / max(1, dd["runs"]), 2),
                     consumer_busy_frac=round(dd["cons_busy"] / max(1, dd["cons_total"]), 3),
                     producer_ring_wait_frac=round(dd["prod_ring_wait"] / max(1, dd["prod_total"]), 3),
                     producer_flush_wait_frac=round(dd["prod_flush_wait"] / max(1, dd["prod_total"]), 3),
                     sum_consumer_ticks=dd["cons_total"], sum_producer_ticks=dd["prod_total"], consumers_per_simd=d[16:20], producers_per_simd=d[20:24],
                     producer_frac_walk_setup_step_push=[round(d[i] / max(1, dd["prod_total"]), 3) for i in (24, 25, 26, 27)])
    res[name] = r
mnv.set_fused_kernel(0)
print(json.dumps(res))

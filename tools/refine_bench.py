"""Timings of the refinement / guided-sampling side kernels at full-frame sizes (1920x1080 trackers, 8M-sample
network batches, pruning the 1.5M-chunk cfg2 tree).  Diagnostic, not the headline bench.
  python tools/refine_bench.py > gpurun_out/refine_bench.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402
import mlp_cases  # noqa: E402


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    out = {}
    rng = np.random.default_rng(1)
    n = 1920 * 1080
    chunk = (rng.zipf(1.3, n) % 1_400_000).astype(np.int64)
    track = np.stack([(1 + chunk % 10).astype(np.float32), chunk.astype(np.float32), (chunk * 5 % 8).astype(np.float32)], 1)
    track[rng.random(n) < 0.6] = (11.0, -1.0, -1.0)
    d_track = torch.from_numpy(track).cuda()
    nodes = torch.empty((4096, 2), dtype=torch.int32, device="cuda")
    out["select_split_ms_1080p"] = timed(lambda: mnv.select_split_candidates(d_track, 4096, nodes))
    out["select_sample_ms_1080p"] = timed(lambda: mnv.select_sample_candidates(d_track, 4096, nodes))

    def torch_vote():
        cand = d_track[d_track[:, 1] >= 0]
        u, c = torch.unique(cand, dim=0, sorted=False, return_counts=True)
        t = torch.cat([-c.to(torch.int32).unsqueeze(-1), u], -1)
        t = t[t[:, 0] < -1]
        t = torch.unique(t, dim=0)
        return t[:4096, 2:].to(torch.int32)

    out["torch_unique_dim_vote_ms_1080p"] = timed(torch_vote, reps=3, warm=1)

    for name, kw in {"w64_l2_out5": dict(hidden_width=64, hidden_layers=2, out_dim=5, pos_octaves=10),
                     "w64_l2_out29_dir": dict(hidden_width=64, hidden_layers=2, out_dim=29, pos_octaves=10, dir_octaves=4, need_viewdir=True),
                     "w128_l4_out29_dir": dict(hidden_width=128, hidden_layers=4, out_dim=29, pos_octaves=10, dir_octaves=4, need_viewdir=True)}.items():
        desc = mnv.mlp_desc(n_clusters=8, **kw)
        params = mlp_cases.make_params(mnv, desc, seed=2)
        mlp = mnv.Mlp(desc, params)
        m = 8_000_000
        cols = 6 if desc.need_viewdir else 3
        x = torch.rand((m, cols), device="cuda") * 2 - 1
        cl = torch.randint(0, 8, (m,), device="cuda", dtype=torch.int16)
        res = torch.empty((m, desc.out_dim), device="cuda")
        ms = timed(lambda: mlp.query(cl, x, res))
        n_pos, n_dir = 3 + 6 * desc.pos_octaves, (3 + 6 * desc.dir_octaves) if desc.need_viewdir else 0
        in_dim, w = n_pos + n_dir, desc.hidden_width
        flops = 2.0 * m * (in_dim * w + (desc.hidden_layers - 1) * w * w + w * desc.out_dim)
        out[f"mlp_{name}"] = {"ms_per_8M_samples": ms, "Msamples_per_s": m / ms / 1e3, "TFLOP_per_s": flops / ms / 1e9,
                              "io_GB_per_s": m * (cols * 4 + 2 + desc.out_dim * 4 + 4) / ms / 1e6}

    # prune on the cfg2 tree with the marks of one 1080p track_visit frame
    tree = cases.make_tree(mnv, cases.CFG2_TREE)
    v = tree.host_view()
    cap, dd = v.capacity, v.data_dim
    tree.move_to_device(need_parent=True, need_sample_counts=True)
    dv = tree.device_view()
    cam = cases.cfg2_camera(mnv, 0)
    opt = mnv.RenderOptions.cli_defaults()
    opt.basis_minmax[1] = 8
    visited = torch.zeros(cap, dtype=torch.int32, device="cuda")
    rgba = torch.empty((1080, 1920, 4), device="cuda")
    out["march_ref_layout_track_visit_ms"] = timed(lambda: mnv.render_voxels(dv, cam, opt, rgba=rgba, visited=visited, track_visit=True), reps=3, warm=1)
    data, child, parent = tree.host_arrays()
    d_child, d_parent = torch.from_numpy(child).cuda(), torch.from_numpy(parent).cuda()
    d_data = torch.from_numpy(data.view(np.int16)).cuda()
    marks = visited.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    edit = mnv.tree_edit(d_child, d_parent, list(v.offset), list(v.scale), cap)
    new_cap, n_del = mnv.prune_tree(edit, d_data, dd, None, marks, cap)
    torch.cuda.synchronize()
    out["prune_cfg2"] = {"ms": (time.perf_counter() - t0) * 1e3, "capacity": cap, "new_capacity": new_cap, "deleted": n_del}
    del d_child, d_parent, d_data, marks, visited, rgba, tree
    torch.cuda.empty_cache()

    # whole frames through the VolumeRenderer loop on the cfg2 tree at 1920x1080 (wall time per render(), device idle at both ends)
    def renderer(**opts):
        t = cases.make_tree(mnv, cases.CFG2_TREE)
        desc = mnv.mlp_desc(n_clusters=8, pos_octaves=10, hidden_width=64, hidden_layers=2, out_dim=t.host_view().data_dim + 1)
        r = mnv.Renderer()
        r.resize(1920, 1080)
        r.set(t, 4 * t.capacity)  # room to grow: no visit tracking, no prune (the reference reserves 20 M chunks)
        g = mnv.ClusterGrid()
        g.grid_dim[0], g.grid_dim[1] = 4, 2
        for i in range(3):
            g.min_position[i], g.range[i] = -1.0, 2.0
        r.set_model(desc, mlp_cases.make_params(mnv, desc, seed=4), g)
        o = r.options
        o.background_brightness, o.step_size, o.stop_thresh, o.sigma_thresh = 0.0, 1e-4, 1e-2, 1e-2
        o.split_batch_size, o.samples_per_corner, o.max_guided_samples = 4096, 8, 128
        for k, v_ in opts.items():
            setattr(o, k, v_)
        return r, t

    def frames(r, n, move=True):
        log = []
        for f in range(n):
            if move:
                az = np.deg2rad(22.5 * f)
                el = np.deg2rad(20.0)
                c = np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
                r.set_camera(tuple(2.6 * c), tuple(c), fx=1600.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = r.render()
            r.download()  # waits for the frame (copies 41 MB to the host: ~2 ms of the figure)
            st["ms"] = (time.perf_counter() - t0) * 1e3
            log.append(st)
        return log

    r, t = renderer()
    out["renderer_plain_frames"] = [round(s["ms"], 3) for s in frames(r, 4)]
    del r, t
    r, t = renderer(use_splitting=True, max_depth=11, max_sample_count=64)
    out["renderer_splitting_frames"] = [{k: (round(v, 3) if k == "ms" else v) for k, v in s.items() if k in ("ms", "added", "resampled", "used_accel", "capacity", "split_candidates")}
                                        for s in frames(r, 6)]
    del r, t
    r, t = renderer(use_guided_sampling=True, max_guided_samples=32)
    out["renderer_guided_frames"] = [{k: (round(v, 3) if k == "ms" else v) for k, v in s.items() if k in ("ms", "guided_samples")} for s in frames(r, 3)]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

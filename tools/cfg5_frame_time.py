"""Wall time per VolumeRenderer::render() on the cfg2 tree at 1920x1080 (no download): plain, guided (fused / four-step), splitting,
and both switches (BASELINE.json configs[4])."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch, cases, mlp_cases, mega_nerf_viewer_amd as mnv
from test_renderer_refine_gpu import make_grid

def run(label, fused=True, frames=12, **opts):
    tree = cases.make_tree(mnv, cases.CFG2_TREE)
    v = tree.host_view()
    desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, dir_octaves=2, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    r = mnv.Renderer(); r.resize(1920, 1080); r.set(tree, v.capacity + 1_000_000)
    r.set_model(desc, mlp_cases.make_params(mnv, desc, seed=21), make_grid(mnv)); r.set_seed(7); r.set_fused_guided(fused)
    for k, val in opts.items(): setattr(r.options, k, val)
    ts = []
    for f in range(frames):
        cam = cases.cfg2_camera(mnv, f % 16); m = cam.c2w
        r.set_camera(tuple(m[9:12]), tuple(m[6:9]), fx=1600.0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = r.render()
        r.sync_tree() if False else None
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{label:38s} ms/frame median {np.median(ts[2:]):.3f}  (fused={st['fused']}, guided_samples={st['guided_samples']}, added={st['added']})")

run("plain (no model switches)")
run("guided sampling, fused", use_guided_sampling=True, max_guided_samples=32)
run("guided sampling, four-step", fused=False, use_guided_sampling=True, max_guided_samples=32)
run("splitting", use_splitting=True, max_depth=12, split_batch_size=4096, samples_per_corner=8)
run("splitting + guided, fused (configs[4])", use_splitting=True, use_guided_sampling=True, max_depth=12, split_batch_size=4096, samples_per_corner=8, max_guided_samples=32)
run("splitting + guided, four-step", fused=False, use_splitting=True, use_guided_sampling=True, max_depth=12, split_batch_size=4096, samples_per_corner=8, max_guided_samples=32)

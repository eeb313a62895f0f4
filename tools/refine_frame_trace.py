import os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch, cases, mlp_cases, mega_nerf_viewer_amd as mnv
from test_renderer_refine_gpu import make_grid
tree = cases.make_tree(mnv, cases.CFG2_TREE); v = tree.host_view()
desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, dir_octaves=2, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
r = mnv.Renderer(); r.resize(1920, 1080); r.set(tree, v.capacity + 1_000_000)
r.set_model(desc, mlp_cases.make_params(mnv, desc, seed=21), make_grid(mnv)); r.set_seed(7)
o = r.options; o.use_splitting=True; o.use_guided_sampling = (sys.argv[1] == "both"); o.max_depth=12; o.split_batch_size=4096; o.samples_per_corner=8; o.max_guided_samples=32
ts=[]
for f in range(14):
    cam = cases.cfg2_camera(mnv, f % 16); m = cam.c2w
    r.set_camera(tuple(m[9:12]), tuple(m[6:9]), fx=1600.0)
    torch.cuda.synchronize(); t0=time.perf_counter(); st=r.render(); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
print(sys.argv[1], "median ms", np.median(ts[2:]), st)

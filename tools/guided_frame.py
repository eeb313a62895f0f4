"""A few guided-sampling / splitting frames through the VolumeRenderer loop on the cfg2 tree (for rocprofv3 --kernel-trace --stats).
usage: python3 tools/guided_frame.py [guided|split] [frames]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402,F401

import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402
import mlp_cases  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "guided"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
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
o.split_batch_size, o.samples_per_corner, o.max_guided_samples = 4096, 8, 32
if mode == "guided":
    o.use_guided_sampling = True
else:
    o.use_splitting, o.max_depth, o.max_sample_count = True, 11, 64
for f in range(n):
    az, el = np.deg2rad(22.5 * f), np.deg2rad(20.0)
    c = np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
    r.set_camera(tuple(2.6 * c), tuple(c), fx=1600.0)
    print(r.render())
r.download()

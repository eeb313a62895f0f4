"""The fused guided-sampling frame (cfg2, 1080p, 9.4 M network evaluations) with 1, 2 and 3 frames in flight on HIP streams: a frame's tail
(workgroups that have run dry while the longest tiles finish) under the next frame's start.  Frames compared bit for bit with the
one-stream frames.    python3 tools/guided_in_flight.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402
import mlp_cases  # noqa: E402

W, H = 1920, 1080
tree = cases.make_tree(mnv, cases.CFG2_TREE)
v = tree.host_view()
tree.move_to_device()
opt = mnv.RenderOptions.cli_defaults()
opt.basis_minmax[1] = 8
opt.max_guided_samples = 32
desc = mnv.mlp_desc(n_clusters=8, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=4))
g = mnv.ClusterGrid()
g.grid_dim[0], g.grid_dim[1] = 4, 2
for i in range(3):
    g.min_position[i], g.range[i] = -1.0, 2.0
cams = [cases.cfg2_camera(mnv, p, W, H, 1600.0) for p in range(16)]
outs = torch.empty((16, H, W, 4), dtype=torch.float32, device="cuda")
ref = None
for k in (1, 2, 3, 4):
    sts = [torch.cuda.Stream() for _ in range(k)]

    def lap():
        for i, cam in enumerate(cams):
            mnv.render_guided_fused(tree.accel, cam, opt, mlp, g, rgba=outs[i], stream=sts[i % k].cuda_stream)

    lap()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        lap()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 64 * 1e3
    if ref is None:
        ref = outs.clone()
    print(f"{k} frame(s) in flight: {ms:.4f} ms per frame, {W * H / ms / 1e3:.0f} Mrays/s, bit-identical to one stream: {bool(torch.equal(outs, ref))}", flush=True)

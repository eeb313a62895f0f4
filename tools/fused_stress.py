"""Many guided-sampling frames of the cfg2 tree through the fused kernels against the four-step path, frame by frame (a race shows up
as an occasional mismatch).  usage: fused_stress.py [reps] [kernel version, default 2] [track]
"track": the instantiation that also writes tracker rows and visit marks (configs[4]: refinement on as well); the rows and marks of every
launch are compared with those of the sample march on the accel.  MNV_STRESS_OCTAVES=10: 63 encoded inputs (two K tiles in the first layer)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch, cases, mlp_cases, mega_nerf_viewer_amd as mnv

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
version = int(sys.argv[2]) if len(sys.argv) > 2 else 2
track = len(sys.argv) > 3 and sys.argv[3] == "track"
W, H = 1920, 1080
tree = cases.make_tree(mnv, cases.CFG2_TREE); v = tree.host_view()
tree.move_to_device(need_parent=track, need_sample_counts=track) if track else tree.move_to_device()
opt = mnv.RenderOptions.cli_defaults(); opt.basis_minmax[1] = 8; opt.max_guided_samples = 32
desc = mnv.mlp_desc(n_clusters=8, pos_octaves=int(os.environ.get("MNV_STRESS_OCTAVES", "4")), hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=4))
g = mnv.ClusterGrid(); g.grid_dim[0], g.grid_dim[1] = 4, 2
for i in range(3): g.min_position[i], g.range[i] = -1.0, 2.0
n_px, dd = W * H, v.data_dim
num = torch.zeros(n_px, dtype=torch.int16, device="cuda"); guided = torch.zeros((n_px, 32, 4), dtype=torch.float32, device="cuda")
clusters = torch.zeros((n_px, 32), dtype=torch.int16, device="cuda"); offsets = torch.empty(n_px, dtype=torch.int64, device="cuda")
cap = 24_000_000
z = torch.empty(cap, dtype=torch.float32, device="cuda"); rows = torch.empty((cap, 3), dtype=torch.float32, device="cuda")
rcl = torch.empty(cap, dtype=torch.int16, device="cuda"); values = torch.empty((cap, dd + 1), dtype=torch.float32, device="cuda")
ref = torch.empty((H, W, 4), dtype=torch.float32, device="cuda"); out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
diag = torch.zeros(32, dtype=torch.int64, device="cuda")
mnv.set_fused_kernel(version); mnv.set_fused_diag(diag)
bad_frames, frames = [], 0
channel_counts = [0, 0, 0, 0]
if track:
    import numpy as np
    opt.max_depth, opt.max_sample_count = 9, 9
    sc = np.full((v.capacity, 8), 8, np.int16); sc[::3] = 12
    sc_dev = torch.from_numpy(sc).cuda(); dv = tree.device_view()
    split0 = torch.empty((H, W, 3), dtype=torch.float32, device="cuda"); sample0 = torch.empty_like(split0)
    split = torch.empty_like(split0); sample = torch.empty_like(split0)
    visited0 = torch.zeros(v.capacity, dtype=torch.int32, device="cuda"); visited = torch.zeros_like(visited0)
for pose in range(16):
    cam = cases.cfg2_camera(mnv, pose, W, H, 1600.0)
    num.zero_()
    if track:
        split0.fill_(-1.0); sample0.fill_(-1.0); visited0.zero_()
        mnv.get_samples_from_voxels_accel_visit(tree.accel, cam, opt, visited0, dv.parent, num, guided, clusters, g, split_track=split0, sample_track=sample0, sample_counts=sc_dev)
    else:
        mnv.get_samples_from_voxels_accel(tree.accel, cam, opt, num, guided, clusters, g)
    total = mnv.compact_guided_samples(num, guided, clusters, offsets, z, rows, rcl)
    mlp.query(rcl, rows, values, n=total)
    mnv.render_nerf_results(tree.device_view(), cam, opt, values, z, offsets, rgba=ref)
    for r in range(reps):
        out.fill_(float("nan"))
        if track:
            split.fill_(-1.0); sample.fill_(-1.0); visited.zero_()
            mnv.render_guided_fused(tree.accel, cam, opt, mlp, g, rgba=out, split_track=split, sample_track=sample, sample_counts=sc_dev, visited=visited, parent=dv.parent)
        else:
            mnv.render_guided_fused(tree.accel, cam, opt, mlp, g, rgba=out)
        torch.cuda.synchronize()
        frames += 1
        n_bad = int((out.view(torch.int32) != ref.view(torch.int32)).any(dim=-1).sum().item())
        if track and not (torch.equal(split, split0) and torch.equal(sample, sample0) and torch.equal(visited, visited0)):
            bad_frames.append((pose, r, "tracker rows / visit marks differ", int((split != split0).any(dim=-1).sum()), int((sample != sample0).any(dim=-1).sum()), int((visited != visited0).sum())))
        if n_bad:
            bad = (out.view(torch.int32) != ref.view(torch.int32)).any(dim=-1)
            ys, xs = torch.nonzero(bad, as_tuple=True)
            d = (out - ref).abs().amax(dim=-1)[bad]
            per_channel = [int(x) for x in (out.view(torch.int32) != ref.view(torch.int32))[bad].sum(dim=0).tolist()]
            channel_counts = [a + b for a, b in zip(channel_counts, per_channel)]
            bad_frames.append((pose, r, n_bad, list(zip(xs.tolist()[:64], ys.tolist()[:64])), [round(float(v), 6) for v in d.tolist()[:64]]))
mnv.set_fused_diag(None); mnv.set_fused_kernel(0)
print({"kernel": version, "track": track, "frames": frames, "bad_frames": bad_frames[:20], "n_bad_frames": len(bad_frames), "differing pixels per channel (r, g, b, a)": channel_counts, "watchdog": int(diag[15].item()),
       "checks(weights, overwrite, registration, twice)": [int(x) for x in diag[28:32].tolist()]})

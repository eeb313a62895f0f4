"""Many guided-sampling frames of the cfg2 tree through the fused kernels against the four-step path, frame by frame (a race shows up
as an occasional mismatch).  usage: fused_stress.py [reps] [kernel version, default 2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch, cases, mlp_cases, mega_nerf_viewer_amd as mnv

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
version = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W, H = 1920, 1080
tree = cases.make_tree(mnv, cases.CFG2_TREE); v = tree.host_view(); tree.move_to_device()
opt = mnv.RenderOptions.cli_defaults(); opt.basis_minmax[1] = 8; opt.max_guided_samples = 32
desc = mnv.mlp_desc(n_clusters=8, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
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
for pose in range(16):
    cam = cases.cfg2_camera(mnv, pose, W, H, 1600.0)
    num.zero_()
    mnv.get_samples_from_voxels_accel(tree.accel, cam, opt, num, guided, clusters, g)
    total = mnv.compact_guided_samples(num, guided, clusters, offsets, z, rows, rcl)
    mlp.query(rcl, rows, values, n=total)
    mnv.render_nerf_results(tree.device_view(), cam, opt, values, z, offsets, rgba=ref)
    for r in range(reps):
        out.fill_(float("nan"))
        mnv.render_guided_fused(tree.accel, cam, opt, mlp, g, rgba=out)
        torch.cuda.synchronize()
        frames += 1
        n_bad = int((out.view(torch.int32) != ref.view(torch.int32)).any(dim=-1).sum().item())
        if n_bad:
            bad = (out.view(torch.int32) != ref.view(torch.int32)).any(dim=-1)
            ys, xs = torch.nonzero(bad, as_tuple=True)
            d = (out - ref).abs().amax(dim=-1)[bad]
            bad_frames.append((pose, r, n_bad, list(zip(xs.tolist()[:64], ys.tolist()[:64])), [round(float(v), 6) for v in d.tolist()[:64]]))
mnv.set_fused_diag(None); mnv.set_fused_kernel(0)
print({"kernel": version, "frames": frames, "bad_frames": bad_frames[:20], "n_bad_frames": len(bad_frames), "watchdog": int(diag[15].item()),
       "checks(weights, overwrite, registration, twice)": [int(x) for x in diag[28:32].tolist()]})

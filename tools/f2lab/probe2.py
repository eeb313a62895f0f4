"""Second-generation probe (hand-edited listings Q1..Q4): GREEN is the unmodified channel-1 path (shows the failure), BLUE's channel sum is replaced,
in the first 32 columns of a window, by the raw float value of an intermediate register of the channel-1 arithmetic of the same lane.  Constant
network, one sample per ray: X = -ln(1 / colour - 1) recovers both.  Prints, for pixels whose green is wrong, the probe value beside a neighbour's.
usage: probe2.py [poses] [reps] [max_g]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch, cases, mlp_cases
import mega_nerf_viewer_amd as mnv

n_poses = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
MAXG = int(sys.argv[3]) if len(sys.argv) > 3 else 1
W, H = 1920, 1080
tree = cases.make_tree(mnv, cases.CFG2_TREE); v = tree.host_view(); tree.move_to_device()
opt = mnv.RenderOptions.cli_defaults(); opt.basis_minmax[1] = 8; opt.max_guided_samples = MAXG
desc = mnv.mlp_desc(n_clusters=8, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
params = mlp_cases.make_params(mnv, desc, seed=4).copy()
per = mnv.Mlp.param_count(desc); IN, dd = 3 + 6 * desc.pos_octaves, v.data_dim
P = params.reshape(8, per)
o_b0 = 64 * IN; o_w1 = o_b0 + 64; o_b1 = o_w1 + 64 * 64; o_w2 = o_b1 + 64
P[:, :o_b0] = 0; P[:, o_w1:o_b1] = 0
P[:, o_b0:o_w1] = np.abs(P[:, o_b0:o_w1]) * 4 + np.float16(0.05); P[:, o_b1:o_w2] = np.abs(P[:, o_b1:o_w2]) * 4 + np.float16(0.05)
P[:] = P[0]
mlp = mnv.Mlp(desc, params)
g = mnv.ClusterGrid(); g.grid_dim[0], g.grid_dim[1] = 4, 2
for i in range(3): g.min_position[i], g.range[i] = -1.0, 2.0
n_px = W * H
num = torch.zeros(n_px, dtype=torch.int16, device="cuda"); guided = torch.zeros((n_px, MAXG, 4), dtype=torch.float32, device="cuda")
clusters = torch.zeros((n_px, MAXG), dtype=torch.int16, device="cuda"); offsets = torch.empty(n_px, dtype=torch.int64, device="cuda")
cap = 6_000_000
z = torch.empty(cap, dtype=torch.float32, device="cuda"); rows = torch.empty((cap, 3), dtype=torch.float32, device="cuda")
rcl = torch.empty(cap, dtype=torch.int16, device="cuda"); values = torch.empty((cap, dd + 1), dtype=torch.float32, device="cuda")
ref = torch.empty((H, W, 4), dtype=torch.float32, device="cuda"); out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
mnv.set_fused_kernel(2)
X_of = lambda c: -torch.log(1.0 / c.double() - 1.0)
recs, frames, groups = [], 0, 0
for pose in range(n_poses):
    cam = cases.cfg2_camera(mnv, pose, W, H, 1600.0)
    num.zero_(); mnv.get_samples_from_voxels_accel(tree.accel, cam, opt, num, guided, clusters, g)
    total = mnv.compact_guided_samples(num, guided, clusters, offsets, z, rows, rcl)
    mlp.query(rcl, rows, values, n=total)
    mnv.render_nerf_results(tree.device_view(), cam, opt, values, z, offsets, rgba=ref); torch.cuda.synchronize()
    one = (num.view(H, W) == 1)
    for r in range(reps):
        out.fill_(float("nan")); mnv.render_guided_fused(tree.accel, cam, opt, mlp, g, rgba=out); torch.cuda.synchronize()
        frames += 1
        badg = one & (out[..., 1].view(torch.int32) != ref[..., 1].view(torch.int32))
        if not bool(badg.any()): continue
        groups += 1
        Xg, Xg0, Xb, Xb0 = X_of(out[..., 1]), X_of(ref[..., 1]), X_of(out[..., 2]), X_of(ref[..., 2])
        ys, xs = torch.nonzero(badg, as_tuple=True)
        for x, y in list(zip(xs.tolist(), ys.tolist()))[:48]:
            nb = [(x + dx, y) for dx in (8, -8, 16, -16) if 0 <= x + dx < W and bool(one[y, x + dx]) and not bool(badg[y, x + dx])]
            rec = {"pose": pose, "rep": r, "px": [x, y], "lane": (y % 8) * 8 + x % 8, "dXg": float(Xg[y, x] - Xg0[y, x]), "probe(blue X)": float(Xb[y, x]), "blue_ref_X": float(Xb0[y, x])}
            if nb:
                rec["neighbour_probe"] = float(Xb[nb[0][1], nb[0][0]]); rec["neighbour_dXg"] = float(Xg[nb[0][1], nb[0][0]] - Xg0[nb[0][1], nb[0][0]])
            recs.append(rec)
print(json.dumps({"frames": frames, "frames_with_wrong_green": groups, "lib": os.environ.get("MNV_LIB_PATH", "")}))
for rec in recs[:120]: print(json.dumps(rec))

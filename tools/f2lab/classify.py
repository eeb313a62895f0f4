"""What is wrong in a wrong pixel of guided_fused2_kernel?  (LAB_NOTEBOOK.md, "the rare wrong denominator".)

Runs the cfg2 guided-sampling frame through the four-step path (kept: samples, z, network outputs) and through the fused kernel of
the library under test (MNV_LIB_PATH: a failing build), and for every pixel whose GREEN differs solves for the one sample whose
second colour denominator explains the difference; the implied change of that sample's channel-1 sum is then compared with what a
list of candidate mechanisms predicts (a CPU evaluation of the network with the partial sums of the output layer's two K tiles):
  ch0 / ch2   the channel-1 sum was taken from another channel's rows (a wrong address in the evaluation)
  noK0 / noK1 rows 12-15 of the output tile (what lanes 48-63 hold of the first 16-row block) lack the first / second K tile
  bias        rows 12-15 hold the bias only;   zero: rows 12-15 are zero;   the same four for ALL nine rows, for rows 9-15, for rows 16-17
usage: classify.py [poses] [reps]        prints one JSON line per analysed pixel group and a summary."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch
import cases
import mlp_cases
import mega_nerf_viewer_amd as mnv

n_poses = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W, H = 1920, 1080
tree = cases.make_tree(mnv, cases.CFG2_TREE)
v = tree.host_view()
tree.move_to_device()
opt = mnv.RenderOptions.cli_defaults()
opt.basis_minmax[1] = 8
opt.max_guided_samples = 32
desc = mnv.mlp_desc(n_clusters=8, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
params = mlp_cases.make_params(mnv, desc, seed=4)
mlp = mnv.Mlp(desc, params)
g = mnv.ClusterGrid()
g.grid_dim[0], g.grid_dim[1] = 4, 2
for i in range(3):
    g.min_position[i], g.range[i] = -1.0, 2.0
n_px, dd = W * H, v.data_dim
num = torch.zeros(n_px, dtype=torch.int16, device="cuda")
guided = torch.zeros((n_px, 32, 4), dtype=torch.float32, device="cuda")
clusters = torch.zeros((n_px, 32), dtype=torch.int16, device="cuda")
offsets = torch.empty(n_px, dtype=torch.int64, device="cuda")
cap = 24_000_000
z = torch.empty(cap, dtype=torch.float32, device="cuda")
rows = torch.empty((cap, 3), dtype=torch.float32, device="cuda")
rcl = torch.empty(cap, dtype=torch.int16, device="cuda")
values = torch.empty((cap, dd + 1), dtype=torch.float32, device="cuda")
ref = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
diag = torch.zeros(32, dtype=torch.int64, device="cuda")
mnv.set_fused_kernel(2)
if not os.environ.get("F2_NO_DIAG"):
    mnv.set_fused_diag(diag)

per = mnv.Mlp.param_count(desc)
IN = 3 + 6 * desc.pos_octaves
P32 = params.astype(np.float32).reshape(desc.n_clusters, per)


def layers_of(c):
    p, off, res = P32[c].astype(np.float64), 0, []
    for o, i in [(64, IN), (64, 64), (dd + 1, 64)]:
        res.append((p[off:off + o * i].reshape(o, i), p[off + o * i:off + o * i + o]))
        off += o * i + o
    return res


def f16(a):
    return np.asarray(a, np.float64).astype(np.float16).astype(np.float64)


def tri(t):
    return 4.0 * np.abs(t - np.floor(t + 0.5)) - 1.0


def network(pos, c):
    """-> outputs [28], partial sums of the output layer over hidden units 0-31 / 32-63 [28] each, its bias [28]"""
    p = pos.astype(np.float64)  # centre 0, extent 1
    feats = [p]
    for k in range(desc.pos_octaves):
        feats += [tri(np.float32(p * 2.0 ** k).astype(np.float64)), tri((np.float32(p * 2.0 ** k) + np.float32(0.25)).astype(np.float64))]
    h = f16(np.concatenate(feats))
    (w0, b0), (w1, b1), (w2, b2) = layers_of(c)
    pre1 = w0 @ h + b0
    h1 = f16(np.maximum(pre1, 0))
    pre2 = w1 @ h1 + b1
    h2 = f16(np.maximum(pre2, 0))
    k0, k1 = w2[:, :32] @ h2[:32], w2[:, 32:] @ h2[32:]
    return k0 + k1 + b2, k0, k1, b2, pre1, pre2


def sh9(d):
    x, y, zz_ = d
    xx, yy, zz = x * x, y * y, zz_ * zz_
    return np.array([0.28209479177387814, -0.4886025119029199 * y, 0.4886025119029199 * zz_, -0.4886025119029199 * x,
                     1.0925484305920792 * x * y, -1.0925484305920792 * y * zz_, 0.31539156525252005 * (2.0 * zz - xx - yy),
                     -1.0925484305920792 * x * zz_, 0.5462742152960396 * (xx - yy)])


hist, shown, n_analysed, n_unsolved = {}, 0, 0, 0
best_log = []
for pose in range(n_poses):
    cam = cases.cfg2_camera(mnv, pose, W, H, 1600.0)
    c2w = np.array(list(cam.c.c2w), np.float64)
    num.zero_()
    mnv.get_samples_from_voxels_accel(tree.accel, cam, opt, num, guided, clusters, g)
    total = mnv.compact_guided_samples(num, guided, clusters, offsets, z, rows, rcl)
    mlp.query(rcl, rows, values, n=total)
    mnv.render_nerf_results(tree.device_view(), cam, opt, values, z, offsets, rgba=ref)
    torch.cuda.synchronize()
    for r in range(reps):
        out.fill_(float("nan"))
        mnv.render_guided_fused(tree.accel, cam, opt, mlp, g, rgba=out)
        torch.cuda.synchronize()
        neq = out.view(torch.int32) != ref.view(torch.int32)
        bad = neq.any(dim=-1)
        nb = int(bad.sum().item())
        per_channel = [int(x) for x in neq.view(-1, 4).sum(dim=0).tolist()]
        print(json.dumps({"pose": pose, "rep": r, "bad_pixels": nb, "per_channel": per_channel}))
        if nb == 0:
            continue
        ys, xs = torch.nonzero(bad, as_tuple=True)
        d_g = (out[..., 1] - ref[..., 1])[bad].abs()
        order = torch.argsort(d_g, descending=True)[:150].tolist()
        for o in order:
            x, y = int(xs[o]), int(ys[o])
            p = y * W + x
            n = int(num[p].item())
            end = int(offsets[p].item())  # inclusive prefix sums (include/mnv.h)
            start = end - n
            vals = values[start:end].double().cpu().numpy()
            zz = z[start:end].double().cpu().numpy()
            pos = rows[start:end].cpu().numpy()
            cl = rcl[start:end].cpu().numpy()
            # the ray's SH basis (mnv_device.h: setup_ray; no rot_dirs in this camera path)
            xyz = np.array([(x + 0.5 - cam.c.cx) / cam.c.fx, -(y + 0.5 - cam.c.cy) / cam.c.fy, -1.0])
            d = np.array([c2w[0] * xyz[0] + c2w[3] * xyz[1] + c2w[6] * xyz[2], c2w[1] * xyz[0] + c2w[4] * xyz[1] + c2w[7] * xyz[2],
                          c2w[2] * xyz[0] + c2w[5] * xyz[1] + c2w[8] * xyz[2]])
            basis = sh9(d / np.linalg.norm(d))
            ti, acc3, wts, X, dens = 1.0, np.zeros(3), [], [], []
            for i in range(n):
                if i < n - 1:
                    wc = np.exp(-vals[i][3] * (zz[i + 1] - zz[i]))
                    wgt = ti * (1.0 - wc)
                else:
                    wc, wgt = 0.0, ti
                xs3 = [float(basis @ vals[i][9 * t:9 * t + 9]) for t in range(3)]
                den = [1.0 + np.exp(-t) for t in xs3]
                acc3 += wgt / np.array(den)
                wts.append(wgt); X.append(xs3); dens.append(den)
                ti *= wc
            good = ref[y, x].double().cpu().numpy()
            got = out[y, x].double().cpu().numpy()
            model_err = float(np.abs(acc3 - good[:3]).max())
            D = got[1] - good[1]
            rec = {"px": [x, y], "lane": (y % 8) * 8 + x % 8, "n": n, "dG": D, "model_err": model_err, "dRB": [got[0] - good[0], got[2] - good[2]]}
            if model_err > 2e-5 or n == 0:
                rec["note"] = "composite model does not reproduce the good pixel"
                print(json.dumps(rec)); continue
            cands = []
            for i in range(n):
                if wts[i] <= 0:
                    continue
                inv = 1.0 / dens[i][1] + D / wts[i]
                if not (0.0 < inv < 1.0):
                    continue
                xp = -np.log(1.0 / inv - 1.0)
                dX = xp - X[i][1]
                o_, k0, k1, b2, pre1, pre2 = network(pos[i], int(cl[i]))
                net_err = float(np.abs(o_ - vals[i]).max())
                B = basis
                def rowsum(vec, fs):
                    return float(sum(B[f] * vec[9 + f] for f in fs))
                hyp = {"ch0": X[i][0] - X[i][1], "ch2": X[i][2] - X[i][1]}
                for name, fs in (("r12_15", (3, 4, 5, 6)), ("all9", range(9)), ("r9_15", range(7)), ("r16_17", (7, 8)), ("r8_11", (0, 1, 2))):
                    hyp[f"noK0:{name}"] = -rowsum(k0, fs)
                    hyp[f"noK1:{name}"] = -rowsum(k1, fs)
                    hyp[f"bias:{name}"] = -rowsum(k0 + k1, fs)
                    hyp[f"zero:{name}"] = -rowsum(o_, fs)
                    hyp[f"nobias:{name}"] = -rowsum(b2, fs)
                    hyp[f"2xK0:{name}"] = rowsum(k0, fs)
                    hyp[f"2xK1:{name}"] = rowsum(k1, fs)
                    # rows replaced by the hidden layers' pre-activations of the same units (a D that was never written)
                    hyp[f"pre2:{name}"] = float(sum(B[f] * (pre2[9 + f] - o_[9 + f]) for f in fs))
                    hyp[f"pre1:{name}"] = float(sum(B[f] * (pre1[9 + f] - o_[9 + f]) for f in fs))
                cands.append((i, dX, net_err, hyp))
            if not cands:
                n_unsolved += 1
                rec["note"] = "no single sample explains the difference"
                print(json.dumps(rec)); continue
            n_analysed += 1
            best = {}
            for i, dX, net_err, hyp in cands:
                for name, pred in hyp.items():
                    e = abs(pred - dX) / max(abs(dX), 1e-9)
                    if name not in best or e < best[name][0]:
                        best[name] = (e, i, dX, pred)
            ranked = sorted(best.items(), key=lambda kv: kv[1][0])[:4]
            rec["best"] = [{"hyp": k, "rel_err": round(b[0], 5), "sample": b[1], "dX": round(b[2], 6), "pred": round(b[3], 6)} for k, b in ranked]
            rec["net_err"] = max(c[2] for c in cands)
            if ranked[0][1][0] < 0.02:
                hist[ranked[0][0]] = hist.get(ranked[0][0], 0) + 1
            else:
                hist["(none < 2 %)"] = hist.get("(none < 2 %)", 0) + 1
            if shown < 60:
                print(json.dumps(rec)); shown += 1
mnv.set_fused_diag(None)
print(json.dumps({"summary": hist, "analysed": n_analysed, "unsolved": n_unsolved, "watchdog": int(diag[15].item())}))

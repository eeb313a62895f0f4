"""Second instrument for the wrong denominator of guided_fused2_kernel: networks whose outputs do not depend on the sample.

modes (argv[1]):
  A  hidden-layer weights zero, biases random: every sample of sub-module c gets the SAME 28 outputs out_c (clusters differ)
  B  the same, and all sub-modules share one parameter set: every sample of the frame gets the same outputs
  C  the random network of the stress test, all sub-modules sharing one parameter set
With constant outputs a wrong channel-1 sum X' of a sample on pixel p obeys  X' - X = sum_f basis_p[f] * delta[f]  for ONE vector delta
(per sub-module in A): a least-squares fit over all wrong pixels gives delta -- which of the nine rows 9..17 of the output tile were
wrong, and by how much -- and the residual says whether "one sample, constant rows" is the right picture at all.  delta is then held
against the candidates: the two K-tile partial sums of the output layer, its bias, other channels' rows, other sub-modules' rows.
usage: constnet.py <A|B|C> [poses] [reps]"""
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

mode = sys.argv[1]
n_poses = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
W, H = 1920, 1080
tree = cases.make_tree(mnv, cases.CFG2_TREE)
v = tree.host_view()
tree.move_to_device()
opt = mnv.RenderOptions.cli_defaults()
opt.basis_minmax[1] = 8
MAXG = int(os.environ.get('F2_MAXG', '32'))
opt.max_guided_samples = MAXG
NCL = 8
desc = mnv.mlp_desc(n_clusters=NCL, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
params = mlp_cases.make_params(mnv, desc, seed=4).copy()
per = mnv.Mlp.param_count(desc)
IN, dd = 3 + 6 * desc.pos_octaves, v.data_dim
P = params.reshape(NCL, per)
o_w0, o_b0 = 0, 64 * IN
o_w1, o_b1 = o_b0 + 64, o_b0 + 64 + 64 * 64
o_w2 = o_b1 + 64
o_b2 = o_w2 + (dd + 1) * 64
if mode in ("A", "B"):
    P[:, o_w0:o_b0] = 0
    P[:, o_w1:o_b1] = 0
    P[:, o_b0:o_w1] = np.abs(P[:, o_b0:o_w1]) * 4 + np.float16(0.05)   # relu keeps them
    P[:, o_b1:o_w2] = np.abs(P[:, o_b1:o_w2]) * 4 + np.float16(0.05)
if mode in ("B", "C") and not os.environ.get("F2_KEEP_CLUSTERS"):
    P[:] = P[0]
mlp = mnv.Mlp(desc, params)
g = mnv.ClusterGrid()
g.grid_dim[0], g.grid_dim[1] = 4, 2
for i in range(3):
    g.min_position[i], g.range[i] = -1.0, 2.0
n_px = W * H
num = torch.zeros(n_px, dtype=torch.int16, device="cuda")
guided = torch.zeros((n_px, MAXG, 4), dtype=torch.float32, device="cuda")
clusters = torch.zeros((n_px, MAXG), dtype=torch.int16, device="cuda")
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
mnv.set_fused_diag(diag)


def f16(a):
    return np.asarray(a, np.float64).astype(np.float16).astype(np.float64)


# constant outputs and the partial sums of the output layer, per sub-module (modes A, B)
const = []
for c in range(NCL):
    p = P[c].astype(np.float64)
    h2 = f16(np.maximum(p[o_b1:o_w2], 0))
    w2, b2 = p[o_w2:o_b2].reshape(dd + 1, 64), p[o_b2:o_b2 + dd + 1]
    k0, k1 = w2[:, :32] @ h2[:32], w2[:, 32:] @ h2[32:]
    const.append(dict(out=k0 + k1 + b2, k0=k0, k1=k1, b2=b2, h1pre=p[o_b0:o_w1], h2pre=p[o_b1:o_w2]))


def sh9(d):
    x, y, zz_ = d
    xx, yy, zz = x * x, y * y, zz_ * zz_
    return np.array([0.28209479177387814, -0.4886025119029199 * y, 0.4886025119029199 * zz_, -0.4886025119029199 * x,
                     1.0925484305920792 * x * y, -1.0925484305920792 * y * zz_, 0.31539156525252005 * (2.0 * zz - xx - yy),
                     -1.0925484305920792 * x * zz_, 0.5462742152960396 * (xx - yy)])


pix = []  # per wrong pixel: basis [9], per candidate sample (index from the end, cluster, dX)
counts = []
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
        counts.append({"pose": pose, "rep": r, "bad_pixels": nb, "per_channel": [int(x) for x in neq.view(-1, 4).sum(dim=0).tolist()]})
        if nb == 0:
            continue
        ys, xs = torch.nonzero(bad, as_tuple=True)
        for x, y in zip(xs.tolist()[:400], ys.tolist()[:400]):
            p = y * W + x
            n = int(num[p].item())
            end = int(offsets[p].item())
            start = end - n
            vals = values[start:end].double().cpu().numpy()
            zz = z[start:end].double().cpu().numpy()
            cl = rcl[start:end].cpu().numpy()
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
            if float(np.abs(acc3 - good[:3]).max()) > 2e-5:
                continue
            D = got[1] - good[1]
            cands = []
            for i in range(n):
                if wts[i] <= 1e-7:
                    continue
                inv = 1.0 / dens[i][1] + D / wts[i]
                if 0.0 < inv < 1.0:
                    cands.append((n - 1 - i, int(cl[i]), -np.log(1.0 / inv - 1.0) - X[i][1], wts[i]))
            if cands:
                pix.append(dict(px=(x, y), pose=pose, rep=r, basis=basis, cands=cands, n=n, D=D, wts=wts, X=X, cl=cl.tolist(), got=got.tolist(), good=good.tolist()))
mnv.set_fused_diag(None)
print(json.dumps({"mode": mode, "frames": counts, "watchdog": int(diag[15].item())}))
if pix:
    import pickle
    os.makedirs(os.path.join(ROOT, "gpurun_out", "f2lab"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "f2lab", f"raw_{mode}{os.environ.get('F2_TAG', '')}.pkl"), "wb") as f:
        pickle.dump(dict(pix=pix, const=const, params=params), f)
if mode == "C" or not pix:
    sys.exit(0)

# ---- fit: alternate between choosing each pixel's sample and solving delta (per sub-module in A, one vector in B)
keys = range(NCL) if mode == "A" else [0]
choice = [max(range(len(p["cands"])), key=lambda k: p["cands"][k][3]) for p in pix]  # start: the heaviest sample
delta = {}
for it in range(8):
    for c in keys:
        rowsA, rhs = [], []
        for p, ch in zip(pix, choice):
            fe, cl_, dX, _ = p["cands"][ch]
            if mode == "B" or cl_ == c:
                rowsA.append(p["basis"]); rhs.append(dX)
        if len(rhs) >= 12:
            delta[c] = np.linalg.lstsq(np.array(rowsA), np.array(rhs), rcond=None)[0]
    new_choice = []
    for p in pix:
        best, bk = None, 0
        for k, (fe, cl_, dX, _) in enumerate(p["cands"]):
            c = 0 if mode == "B" else cl_
            if c not in delta:
                continue
            e = abs(float(p["basis"] @ delta[c]) - dX)
            if best is None or e < best:
                best, bk = e, k
        new_choice.append(bk)
    if new_choice == choice:
        break
    choice = new_choice
res = {}
for c in keys:
    if c not in delta:
        continue
    errs, from_end, used = [], [], 0
    for p, ch in zip(pix, choice):
        fe, cl_, dX, _ = p["cands"][ch]
        if mode == "B" or cl_ == c:
            errs.append(float(p["basis"] @ delta[c]) - dX); from_end.append(fe); used += 1
    K = const[c]
    cand = {"-K0": -K["k0"][9:18], "-K1": -K["k1"][9:18], "-(K0+K1)": -(K["k0"] + K["k1"])[9:18], "-out": -K["out"][9:18], "-b2": -K["b2"][9:18],
            "ch0-ch1": K["out"][0:9] - K["out"][9:18], "ch2-ch1": K["out"][18:27] - K["out"][9:18]}
    for c2 in range(NCL):
        if c2 != c and mode == "A":
            cand[f"out[{c2}]-out"] = const[c2]["out"][9:18] - K["out"][9:18]
            cand[f"b2[{c2}]-b2"] = const[c2]["b2"][9:18] - K["b2"][9:18]
    res[c] = {"pixels": used, "rms_residual": float(np.sqrt(np.mean(np.square(errs)))), "rms_dX": float(np.sqrt(np.mean([p["cands"][ch][2] ** 2 for p, ch in zip(pix, choice)]))),
              "delta_rows_9_17": [round(float(t), 5) for t in delta[c]],
              "samples_from_end_hist": {str(k): from_end.count(k) for k in sorted(set(from_end))[:8]},
              "candidates(rows 9..17)": {k: [round(float(t), 5) for t in vv] for k, vv in cand.items()}}
print(json.dumps({"fit": res}, indent=None))

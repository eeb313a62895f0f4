"""Debugging aid for the producer / consumer kernel built with -DMNV_F2_LOG: every owner lane logs the samples it composites (results,
flag word, slot, transmittance before); a frame that differs from the four-step path is compared, sample by sample, with the log of a
frame that matched.  usage: fused_log_diff.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch, cases, mlp_cases, mega_nerf_viewer_amd as mnv

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, H = 1920, 1080
tree = cases.make_tree(mnv, cases.CFG2_TREE); v = tree.host_view(); tree.move_to_device()
opt = mnv.RenderOptions.cli_defaults(); opt.basis_minmax[1] = 8; opt.max_guided_samples = 32
desc = mnv.mlp_desc(n_clusters=8, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=4))
g = mnv.ClusterGrid(); g.grid_dim[0], g.grid_dim[1] = 4, 2
for i in range(3): g.min_position[i], g.range[i] = -1.0, 2.0
n_px = W * H
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda"); good = torch.empty_like(out)
logs = [torch.zeros((n_px, 40, 8), dtype=torch.float32, device="cuda") for _ in range(2)]
vlogs = [torch.zeros((n_px, 64, 2, 24), dtype=torch.float32, device="cuda") for _ in range(2)]
diag = torch.zeros(32, dtype=torch.int64, device="cuda")
mnv.set_fused_kernel(2); mnv.set_fused_diag(diag)
shown = 0
for pose in range(16):
    cam = cases.cfg2_camera(mnv, pose, W, H, 1600.0)
    have_good = False
    for r in range(reps):
        which = 1 if have_good else 0
        logs[which].zero_(); diag[31] = logs[which].data_ptr(); vlogs[which].zero_(); diag[30] = vlogs[which].data_ptr()
        out.fill_(float("nan"))
        mnv.render_guided_fused(tree.accel, cam, opt, mlp, g, rgba=out)
        torch.cuda.synchronize()
        if not have_good:
            good.copy_(out); have_good = True  # (checked against the later frames: the majority wins)
            continue
        bad = (out.view(torch.int32) != good.view(torch.int32)).any(dim=-1)
        if not bool(bad.any()): continue
        ys, xs = torch.nonzero(bad, as_tuple=True)
        print(f"pose {pose} rep {r}: {int(bad.sum())} pixels differ")
        for x, y in list(zip(xs.tolist(), ys.tolist()))[:6]:
            p = y * W + x
            a, b = logs[0][p].cpu(), logs[1][p].cpu()
            na, nb = int(a[39, 0]), int(b[39, 0])
            print(f"  pixel ({x},{y}) lane {(y % 8) * 8 + x % 8}: composited {na} vs {nb}; ns {a[39,5]} vs {b[39,5]}; out {a[39,1:4].tolist()} vs {b[39,1:4].tolist()}")
            for i in range(max(na, nb)):
                ra, rb = a[i], b[i]
                same = torch.equal(ra[:4].view(torch.int32), rb[:4].view(torch.int32))
                ma, mb = int(ra[4:5].view(torch.int32)), int(rb[4:5].view(torch.int32))
                sa, sb = int(ra[5:6].view(torch.int32)), int(rb[5:6].view(torch.int32))
                if not same:
                    ia, ib = int(ra[7:8].view(torch.int32)) & 63, int(rb[7:8].view(torch.int32)) & 63
                    va, vb = vlogs[0][p, ia].cpu(), vlogs[1][p, ib].cpu()
                    for sh in range(2):
                        d = [(k, float(va[sh, k]), float(vb[sh, k])) for k in range(24) if k != 21 and va[sh, k].view(torch.int32) != vb[sh, k].view(torch.int32)]
                        wa, wb = int(va[sh, 21:22].view(torch.int32)), int(vb[sh, 21:22].view(torch.int32))
                        print(f"      share {sh}: differing words {d}; column/n/half/run {wa & 255}/{(wa >> 8) & 255}/{(wa >> 16) & 15}/{wa >> 20} vs {wb & 255}/{(wb >> 8) & 255}/{(wb >> 16) & 15}/{wb >> 20}")
                if not same or (ma & 0xffff7f) != (mb & 0xffff7f) or ra[6] != rb[6]:
                    print(f"    sample {i}: res {[round(float(t), 6) for t in ra[:4]]} vs {[round(float(t), 6) for t in rb[:4]]} meta {ma:#x} vs {mb:#x} slot {sa} vs {sb} ti {float(ra[6]):.6g} vs {float(rb[6]):.6g}")
        shown += 1
        if shown >= 10: break
    if shown >= 10: break
print("checks", [int(x) for x in diag[28:30].tolist()], "frames with differences shown:", shown)

"""Both fused kernels against the four-step path on the small parity cases: mismatch counts instead of assertions (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch, cases, mlp_cases, mega_nerf_viewer_amd as mnv
from test_renderer_refine_gpu import make_grid
from test_guided_fused_gpu import four_step_frame

CASES = [("rgba_d5", False, 0, 16, 6), ("sh9_d7_aniso", False, 0, 128, 6), ("sh4_d6", True, 0, 8, 6), ("shell_d7_sh9", True, 3, 128, 6),
         ("sh9_d7_aniso", False, 0, 3, 4), ("sh16_d4", False, 0, 32, 6)]
for case, need_viewdir, n_emb, max_g, n_clusters in CASES:
    spec = cases.CASES[case]
    tree = cases.make_tree(mnv, spec["tree"]); v = tree.host_view(); tree.move_to_device()
    cam = cases.make_camera(mnv, spec["camera"]); opt = cases.make_options(mnv, spec["options"])
    opt.basis_minmax[0], opt.basis_minmax[1] = 0, max(v.basis_dim - 1, 0)
    opt.max_guided_samples = max_g; opt.need_viewdir = need_viewdir; opt.appearance_embedding = 1 if n_emb else -1
    desc = mnv.mlp_desc(n_clusters=n_clusters, pos_octaves=4, dir_octaves=2, need_viewdir=need_viewdir, n_embeddings=n_emb, embedding_dim=8 if n_emb else 0,
                        hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=21)); grid = make_grid(mnv)
    dim = 4 + (3 if need_viewdir else 0) + (1 if n_emb else 0)
    ref, ref8, total = four_step_frame(mnv, torch, tree, cam, opt, mlp, grid, max_g, dim)
    for version in (2, 1):
        mnv.set_fused_kernel(version)
        diag = torch.zeros(32, dtype=torch.int64, device="cuda")
        mnv.set_fused_diag(diag)
        out = torch.full((cam.height, cam.width, 4), float("nan"), dtype=torch.float32, device="cuda")
        counter = torch.zeros(1, dtype=torch.int64, device="cuda")
        try:
            mnv.render_guided_fused(tree.accel, cam, opt, mlp, grid, rgba=out, sample_counter=counter)
            torch.cuda.synchronize()
        except mnv.MnvError as e:
            print(case, version, "ERROR", e); continue
        finally:
            mnv.set_fused_diag(None)
        got = out.cpu().numpy()
        bad = (cases.bits(got) != cases.bits(ref)).any(axis=-1)
        ys, xs = np.nonzero(bad)
        print(case, f"{cam.width}x{cam.height} max_g={max_g} kernel={version} samples={int(counter.item())}/{total} bad_pixels={int(bad.sum())} nan_pixels={int(np.isnan(got).any(axis=-1).sum())} "
              f"watchdog={int(diag[15].item())} runs={int(diag[1].item())} first_bad={list(zip(xs[:4].tolist(), ys[:4].tolist()))}")
mnv.set_fused_kernel(0)

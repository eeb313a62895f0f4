"""mlp_forward_kernel alone (mnv_query_submodules, 8 M samples) for profiling: python3 tools/mlp_prof.py [w128|w64]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import mega_nerf_viewer_amd as mnv  # noqa: E402
import mlp_cases  # noqa: E402

kw = dict(hidden_width=128, hidden_layers=4, out_dim=29, pos_octaves=10, dir_octaves=4, need_viewdir=True)
if len(sys.argv) > 1 and sys.argv[1] == "w64":
    kw = dict(hidden_width=64, hidden_layers=2, out_dim=29, pos_octaves=10, dir_octaves=4, need_viewdir=True)
for a in sys.argv[2:]:  # overrides: name=value (ablation timing: hidden_layers=1, need_viewdir=0, ...)
    k, v = a.split("=")
    kw[k] = type(kw[k])(int(v))
desc = mnv.mlp_desc(n_clusters=8, **kw)
mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=2))
m = 8_000_000
x = torch.rand((m, 6 if desc.need_viewdir else 3), device="cuda") * 2 - 1
cl = torch.randint(0, 8, (m,), device="cuda", dtype=torch.int16)
res = torch.empty((m, desc.out_dim), device="cuda")
for _ in range(2):
    mlp.query(cl, x, res)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4):
    mlp.query(cl, x, res)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 4
n_pos, n_dir = 3 + 6 * desc.pos_octaves, (3 + 6 * desc.dir_octaves if desc.need_viewdir else 0)
w = desc.hidden_width
flops = 2.0 * m * ((n_pos + n_dir) * w + (desc.hidden_layers - 1) * w * w + w * desc.out_dim)
print(f"{ms:.3f} ms per 8 M samples, {flops / ms / 1e9:.0f} TFLOP/s  {' '.join(sys.argv[1:])}")

"""Soak of the VolumeRenderer refinement loop: many frames, moving camera, a capacity small enough to force prunes; after every
frame the tree's links are checked and the packed accel (when in use) is compared with the reference-layout kernel.
  python3 tools/refine_soak.py [frames [room [case]]]      case: a name of tests/cases.py CASES (default sh4_d6; shell_d7_sh9: a tree whose packed layout
has inline cell words from the start and gets brick records while the loop deepens it)
On the test-hook build with MNV_REFRESH_DEBUG=2 every refresh / prune of the run also verifies every patched lookup word (tools/soak_all.sh)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):  # oracle: only because the test helpers import it
    sys.path.insert(0, p)

import torch  # noqa: E402

import cases  # noqa: E402
import mega_nerf_viewer_amd as mnv  # noqa: E402
import mlp_cases  # noqa: E402
from test_renderer_refine_gpu import check_tree_links, make_grid  # noqa: E402

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
room = int(sys.argv[2]) if len(sys.argv) > 2 else 900
case = sys.argv[3] if len(sys.argv) > 3 else "sh4_d6"
spec = cases.CASES[case]
tree = cases.make_tree(mnv, spec["tree"])
cap0 = tree.capacity
desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=tree.host_view().data_dim + 1)
r = mnv.Renderer()
w, h = 200, 160
r.resize(w, h)
r.set(tree, cap0 + room)
r.set_model(desc, mlp_cases.make_params(mnv, desc, seed=3), make_grid(mnv))
r.set_seed(1, accel_rebuild_after=1)
o = r.options
o.background_brightness, o.use_splitting, o.max_depth, o.split_batch_size, o.samples_per_corner, o.max_sample_count = 0.0, True, 9, 300, 2, 16
cam_check = cases.make_camera(mnv, spec["camera"])
opt_check = cases.make_options(mnv, spec["options"])
opt_check.basis_minmax[1] = max(tree.host_view().basis_dim - 1, 0)
a = torch.empty((cam_check.height, cam_check.width, 4), device="cuda")
b = torch.empty_like(a)
stats = dict(added=0, resampled=0, pruned=0, prunes=0, accel_frames=0, mismatches=0)
for f in range(n_frames):
    ang = 0.15 * f
    r.set_camera((-2.4 * np.cos(ang) - 1.1 * np.sin(ang), 1.1 * np.cos(ang) - 2.4 * np.sin(ang), 1.6 + 0.3 * np.sin(0.4 * f)),
                 (-0.72 * np.cos(ang) - 0.33 * np.sin(ang), 0.33 * np.cos(ang) - 0.72 * np.sin(ang), 0.48), fx=700.0)
    st = r.render()
    stats["added"] += st["added"]
    stats["resampled"] += st["resampled"]
    stats["accel_frames"] += st["used_accel"]
    if st["pruned"] > 0:
        stats["pruned"] += st["pruned"]
        stats["prunes"] += 1
    r.sync_tree()
    _, child, parent = tree.host_arrays()
    assert child.shape[0] == st["capacity"], (child.shape, st)
    check_tree_links(child, parent, st["capacity"])
    assert np.isfinite(r.download()).all()
    if st["used_accel"] and st["pruned"] <= 0:  # the accel is current (a prune in this frame would have invalidated it): it must
        # agree with the reference-layout kernel on the tree as it is now
        mnv.render_voxels(tree.device_view(), cam_check, opt_check, rgba=a)
        mnv.render_voxels_accel(tree.accel, cam_check, opt_check, rgba=b)
        torch.cuda.synchronize()
        if not torch.equal(a.view(torch.int32), b.view(torch.int32)):
            stats["mismatches"] += 1
            print("frame", f, "accel != reference layout", st)
print("frames", n_frames, "capacity", cap0, "->", st["capacity"], stats)
assert stats["mismatches"] == 0

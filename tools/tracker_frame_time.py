"""The reference's every-frame call on the packed layout (render_voxels with both tracker tensors, cuda_renderer.cpp:141-142), cfg2 at 1920x1080, one
frame at a time: the tracker rows against the oracle's (one pose), then HIP-event times of the plain frame, the split tracker alone, the sample tracker
alone, both, and both without a sample_counts array.  Under rocprofv3 --kernel-trace --stats the tracker instantiation is march_accel_kernel<9,256,2,true>."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch, cases, mega_nerf_viewer_amd as mnv, mnv_oracle as orc
W, H = 1920, 1080
tree = cases.make_tree(mnv, cases.CFG2_TREE)
v = tree.host_view(); cap = v.capacity
tree.move_to_device()
opt = mnv.RenderOptions.cli_defaults()
opt.basis_minmax[1] = 8
print("max_depth", opt.max_depth, "max_sample_count", opt.max_sample_count, "bg", opt.background_brightness)
cam = cases.cfg2_camera(mnv, 5, W, H, 1600.0)
counts = torch.full((cap, 8), 8, dtype=torch.int16, device="cuda")
split = torch.full((H*W, 3), -1.0, device="cuda"); sample = torch.full((H*W, 3), -1.0, device="cuda")
img = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda"); depth = torch.full((H, W), 1e9, device="cuda")
mnv.render_voxels_accel_visit(tree.accel, cam, opt, None, None, rgba8=img, split_track=split, sample_track=sample, sample_counts=counts, tmax_px=depth, rgba8_init=img)
torch.cuda.synchronize()
ch = np.full((cap, 8), 8, np.int16)
ot = orc.tree_from_view(v, sample_counts=ch)
want = orc.render(ot, cam.c, opt, want_rgba8=True, want_trackers=True, tmax_px=np.full((H, W), 1e9, np.float32), rgba8_init=np.zeros((H, W, 4), np.uint8))
s, a = split.cpu().numpy().reshape(H, W, 3), sample.cpu().numpy().reshape(H, W, 3)
for k in range(3):
    print("col", k, "split diff", int((s[..., k] != want["split"][..., k]).sum()), "sample diff", int((a[..., k] != want["sample"][..., k]).sum()))
bad = np.argwhere(s[..., 0] != want["split"][..., 0])
if len(bad):
    y, x = bad[0]; print("first", y, x, s[y, x], want["split"][y, x], a[y, x], want["sample"][y, x])

def timed(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = torch.empty((H, W, 4), device="cuda")
for pose in (0, 5):
    cam = cases.cfg2_camera(mnv, pose, W, H, 1600.0)
    print("pose", pose,
          "plain %.3f" % timed(lambda: mnv.render_voxels_accel(tree.accel, cam, opt, rgba=out)),
          "split only %.3f" % timed(lambda: mnv.render_voxels_accel_track(tree.accel, cam, opt, rgba=out, split_track=split)),
          "sample only (counts) %.3f" % timed(lambda: mnv.render_voxels_accel_track(tree.accel, cam, opt, rgba=out, sample_track=sample, sample_counts=counts)),
          "both %.3f" % timed(lambda: mnv.render_voxels_accel_track(tree.accel, cam, opt, rgba=out, split_track=split, sample_track=sample, sample_counts=counts)),
          "both, no counts %.3f" % timed(lambda: mnv.render_voxels_accel_track(tree.accel, cam, opt, rgba=out, split_track=split, sample_track=sample)))

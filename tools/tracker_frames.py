"""Nothing but tracker frames (render_voxels with both tracker tensors on the packed layout, cuda_renderer.cpp:141-142): the cfg2 tree at 1920x1080, the 16-pose
orbit, `laps` times -- the program the PMC passes of tools/prof_tracker_traffic.sh run.  With the test-hook build, MNV_BRICK_LEVELS=0 builds the accel without inline
cell words / brick records: the tracker kernel then walks the node words as it did until round 5."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, cases, mega_nerf_viewer_amd as mnv
W, H = 1920, 1080
laps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tree = cases.make_tree(mnv, cases.CFG2_TREE)
cap = tree.host_view().capacity
tree.move_to_device()
opt = mnv.RenderOptions.cli_defaults()
counts = torch.full((cap, 8), 8, dtype=torch.int16, device="cuda")
split = torch.full((H * W, 3), -1.0, device="cuda"); sample = torch.full((H * W, 3), -1.0, device="cuda")
out = torch.empty((H, W, 4), device="cuda")
cams = [cases.cfg2_camera(mnv, p, W, H, 1600.0) for p in range(16)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for lap in range(laps + 1):
    if lap == 1: e0.record()
    for c in cams:
        mnv.render_voxels_accel_track(tree.accel, c, opt, rgba=out, split_track=split, sample_track=sample, sample_counts=counts)
e1.record(); torch.cuda.synchronize()
print({"ms_per_tracker_frame": round(e0.elapsed_time(e1) / (16 * laps), 4), "accel": mnv.accel_info(tree.accel)})

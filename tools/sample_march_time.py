"""Device time of the sample-emitting march alone (mnv_get_samples_from_voxels_accel) on the cfg2 tree at 1080p; MNV_BLOCKS_PER_CU sets
the wavefronts per SIMD.  Used to split the fused guided frame's time into march and network."""
import os as _os
# the MNV_* knobs this tool reads exist in the test-hook build of the library only (csrc/mnv_knobs.h)
_os.environ.setdefault("MNV_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mega-nerf-viewer_amd", "testhooks", "libmnv.so"))
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, cases, mega_nerf_viewer_amd as mnv
W, H = 1920, 1080
tree = cases.make_tree(mnv, cases.CFG2_TREE); tree.move_to_device()
opt = mnv.RenderOptions.cli_defaults(); opt.basis_minmax[1] = 8; opt.max_guided_samples = 32
g = mnv.ClusterGrid(); g.grid_dim[0], g.grid_dim[1] = 4, 2
for i in range(3): g.min_position[i], g.range[i] = -1.0, 2.0
n_px = W * H
num = torch.zeros(n_px, dtype=torch.int16, device="cuda")
guided = torch.zeros((n_px, 32, 4), dtype=torch.float32, device="cuda")
clusters = torch.zeros((n_px, 32), dtype=torch.int16, device="cuda")
cams = [cases.cfg2_camera(mnv, p, W, H, 1600.0) for p in range(8)]
def run():
    for c in cams:
        num.zero_(); mnv.get_samples_from_voxels_accel(tree.accel, c, opt, num, guided, clusters, g)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); run(); e1.record(); torch.cuda.synchronize()
print("blocks_per_cu", os.environ.get("MNV_BLOCKS_PER_CU", "default(6)"), "sample march ms/frame %.4f" % (e0.elapsed_time(e1) / 16))

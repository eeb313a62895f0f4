import os as _os
# the MNV_* knobs this tool reads exist in the test-hook build of the library only (csrc/mnv_knobs.h)
_os.environ.setdefault("MNV_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mega-nerf-viewer_amd", "testhooks", "libmnv.so"))
import sys, os
ROOT="/root/repo"; sys.path[:0]=[ROOT, os.path.join(ROOT,"tests")]
import torch, cases, mega_nerf_viewer_amd as mnv
tree = cases.make_tree(mnv, cases.CFG2_TREE); tree.move_to_device()
opt = mnv.RenderOptions.cli_defaults(); opt.basis_minmax[1] = 8
out = torch.empty((1080,1920,4), device="cuda")
cam = cases.cfg2_camera(mnv, 0, 1920, 1080, 1600.0)
mnv.render_voxels_accel(tree.accel, cam, opt, rgba=out)
torch.cuda.synchronize()
del tree

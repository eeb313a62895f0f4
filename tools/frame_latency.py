import sys, os, numpy as np
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"tests")]
import torch, cases, mega_nerf_viewer_amd as mnv
tree = cases.make_tree(mnv, cases.CFG2_TREE); tree.move_to_device()
opt = mnv.RenderOptions.cli_defaults(); opt.basis_minmax[1] = 8
for (w,h,fx) in [(800,800,1111.0),(1920,1080,1600.0),(3840,2160,3200.0),(256,256,400.0)]:
    cam = cases.cfg2_camera(mnv, 3, w, h, fx)
    out = torch.empty((h,w,4), device="cuda")
    for _ in range(3): mnv.render_voxels_accel(tree.accel, cam, opt, rgba=out)
    torch.cuda.synchronize(); mnv.set_timing(True)
    for _ in range(20): mnv.render_voxels_accel(tree.accel, cam, opt, rgba=out)
    torch.cuda.synchronize(); ms,n = mnv.take_timing(); mnv.set_timing(False)
    print(w,h, "ms/frame %.4f" % (ms/n), "Mrays/s %.0f" % (w*h/(ms/n)/1e3))

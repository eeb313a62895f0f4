import sys, os, time, numpy as np
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"tests"), os.path.join(os.getcwd(),"oracle")]
import torch, cases, mega_nerf_viewer_amd as mnv, mnv_oracle as orc
spec=dict(cases.CFG3_TREE); spec["depth"]=11
t0=time.time(); tree=cases.make_tree(mnv, spec); print("chunks", tree.capacity, "synth s", round(time.time()-t0,1), flush=True)
t0=time.time(); tree.move_to_device(); print("upload+accel s", round(time.time()-t0,1), flush=True)
cam=cases.cfg3_camera(mnv, 5, 960, 540, 700.0); opt=mnv.RenderOptions.cli_defaults(); opt.basis_minmax[1]=8
out=torch.empty((540,960,4),device="cuda"); out2=torch.empty_like(out)
mnv.render_voxels_accel(tree.accel, cam, opt, rgba=out); mnv.render_voxels(tree.device_view(), cam, opt, rgba=out2); torch.cuda.synchronize()
t0=time.time(); ref=orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt)["rgba"]; print("oracle s", round(time.time()-t0,1))
a=out.cpu().numpy(); b=out2.cpu().numpy()
print("accel==oracle", np.array_equal(a.view(np.uint32), ref.view(np.uint32)), "ref_layout==oracle", np.array_equal(b.view(np.uint32), ref.view(np.uint32)), "alpha>0 px", int((ref[...,3]>0).sum()))

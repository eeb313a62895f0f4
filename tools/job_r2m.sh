export TMPDIR=/tmp
cat > /tmp/chk.py <<'PY'
import os, sys
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"tests")]
import torch, cases, mega_nerf_viewer_amd as mnv
W,H=1920,1080
tree = cases.make_tree(mnv, cases.CFG2_TREE); tree.move_to_device()
opt = mnv.RenderOptions.cli_defaults(); opt.basis_minmax[1]=8; opt.max_guided_samples=32
g = mnv.ClusterGrid(); g.grid_dim[0], g.grid_dim[1] = 4, 2
for i in range(3): g.min_position[i], g.range[i] = -1.0, 2.0
cam = cases.cfg2_camera(mnv, 3, W, H, 1600.0)
n_px=W*H
num = torch.zeros(n_px, dtype=torch.int16, device="cuda")
guided = torch.zeros((n_px, 32, 4), dtype=torch.float32, device="cuda")
clusters = torch.full((n_px, 32), -1, dtype=torch.int16, device="cuda")
mnv.get_samples_from_voxels_accel(tree.accel, cam, opt, num, guided, clusters, g)
torch.cuda.synchronize()
c0 = clusters[:,0].view(H,W)
n = num.view(H,W)
hit = n>0
print("rays with samples", int(hit.sum()))
# cluster of first sample per pixel: how many 8x8 tiles have more than one cluster among their hit pixels?
t = c0.view(H//8,8,W//8,8).permute(0,2,1,3).reshape(-1,64)
th = hit.view(H//8,8,W//8,8).permute(0,2,1,3).reshape(-1,64)
mixed=0; tot=0
for i in range(t.shape[0]):
    v = t[i][th[i]]
    if v.numel():
        tot+=1
        if (v!=v[0]).any(): mixed+=1
print("tiles with samples", tot, "mixed-cluster tiles", mixed)
print("cluster histogram", torch.bincount(c0[hit].long()+1).tolist())
# within a ray: do clusters change?
cl = clusters.long(); valid = cl>=0
chg = ((cl[:,1:]!=cl[:,:-1]) & valid[:,1:] & valid[:,:-1]).sum()
print("cluster changes along rays", int(chg), "of", int(valid.sum()))
PY
python /tmp/chk.py 2>&1 | tail -6

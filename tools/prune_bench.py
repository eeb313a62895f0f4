"""Prune of the cfg2 tree with the marks of one 1080p view: time of mnv_prune_tree alone, with the accel following in place
(mnv_prune_tree_accel), and of the in-place rebuild it replaces."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, cases, mega_nerf_viewer_amd as mnv

def setup():
    tree = cases.make_tree(mnv, cases.CFG2_TREE)
    v = tree.host_view()
    tree.move_to_device(max_capacity=v.capacity + 64, need_parent=True, need_sample_counts=True)
    dv = tree.device_view()
    cam = cases.cfg2_camera(mnv, 3)
    opt = mnv.RenderOptions.cli_defaults(); opt.basis_minmax[1] = 8
    visited = torch.zeros(v.capacity + 64, dtype=torch.int32, device="cuda")
    out = torch.empty((1080, 1920, 4), dtype=torch.float32, device="cuda")
    mnv.render_voxels_accel_visit(tree.accel, cam, opt, visited, dv.parent, rgba=out)
    torch.cuda.synchronize()
    edit = mnv.TreeEdit(); edit.child, edit.parent, edit.N, edit.capacity = dv.child, dv.parent, 2, v.capacity
    for i in range(3): edit.offset[i], edit.scale[i] = v.offset[i], v.scale[i]
    return tree, v, dv, edit, visited

for mode in ("plain", "plain", "accel", "accel"):
    tree, v, dv, edit, visited = setup()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    new_cap, n_del = mnv.prune_tree(edit, dv.data, v.data_dim, dv.sample_counts, visited, v.capacity + 64, accel=tree.accel if mode == "accel" else 0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(mode, "prune ms %.3f" % ((t1 - t0) * 1e3), "chunks", v.capacity, "->", new_cap)
    if mode == "plain":
        pv = mnv.TreeView()
        for f in ("data", "child", "parent", "sample_counts", "offset", "scale", "N", "data_dim", "format", "basis_dim"): setattr(pv, f, getattr(dv, f))
        pv.capacity = new_cap
        t0 = time.perf_counter(); mnv.accel_rebuild(tree.accel, pv); torch.cuda.synchronize(); t1 = time.perf_counter()
        print("rebuild ms %.3f" % ((t1 - t0) * 1e3))

"""Shared inputs of the guided-sampling goldens (tests/golden/ref_guided_*.npz)."""
import numpy as np

import cases


def cluster_grid(cls):
    g = cls()
    g.grid_dim[0], g.grid_dim[1] = 3, 2
    for i, (lo, rng) in enumerate([(-1.0, 2.0), (-1.1, 2.2), (-0.9, 1.8)]):
        g.min_position[i], g.range[i] = lo, rng
    return g


def get_samples_setup(mnv):
    spec = cases.CASES["sh4_d6"]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.need_viewdir, opt.appearance_embedding, opt.max_guided_samples = True, 7, 8
    opt.rot_dirs[0], opt.rot_dirs[1] = 0.2, -0.1
    return tree, cam, opt, 8  # samples_dim = 4 + 3 + 1


def onscreen_tmax(cam, seed=105):
    """Stand-in for the GL depth attachment of the reference's offscreen == false call of get_samples_from_voxels (renderer_kernel.cu:354-357):
    world-space ray limits scattered around the camera's distance to the scene centre, 15 % "no mesh" (1e9f), 5 % "mesh at the lens" (0)."""
    rng = np.random.default_rng(seed)
    h, w = cam.height, cam.width
    dist = float(np.linalg.norm(np.array(list(cam.c.c2w), np.float64)[9:12]))
    tmax = (dist * rng.uniform(0.55, 1.35, size=(h, w))).astype(np.float32)
    u = rng.uniform(size=(h, w))
    tmax[u < 0.15] = np.float32(1e9)
    tmax[u > 0.95] = np.float32(0.0)
    return tmax


def nerf_results_setup(mnv, case):
    spec = cases.CASES[case]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    v = tree.host_view()
    n = cam.width * cam.height
    rng = np.random.default_rng(11)
    counts = rng.integers(0, 6, n).astype(np.int64)
    counts[:5] = [0, 1, 0, 2, 1]
    offsets = np.cumsum(counts)
    total = int(offsets[-1])
    values = rng.normal(0, 1.0, (total, v.data_dim + 1)).astype(np.float32)
    values[:, 3] = np.abs(values[:, 3]) * 20
    z = np.concatenate([np.sort(rng.uniform(0.5, 6.0, c)) for c in counts]).astype(np.float32)
    return tree, cam, opt, values, z, offsets

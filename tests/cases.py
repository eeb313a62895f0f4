"""Shared parity cases: (tree parameters, camera, options) triples used by the CPU tests, the GPU
parity tests and the golden-vector generator.  The list follows SURVEY.md 8(c)'s fixture list:
cfg1 depth-4 SH1 256x256; depth-6 SH4; depth-7 SH9 anisotropic invradius3; RGBA-format tree;
render_depth; clipped render_bbox; non-zero rot_dirs; restricted basis_minmax; camera inside the
volume; ray-misses-everything; plus SH16 / SH25 trees."""
import numpy as np


def make_tree(mnv, spec):
    kind = spec["kind"]
    args = {k: v for k, v in spec.items() if k != "kind"}
    if kind == "random":
        return mnv.N3Tree.synth_random(**args)
    if kind == "shell":
        return mnv.N3Tree.synth_shell(**args)
    if kind == "terrain":
        return mnv.N3Tree.synth_terrain(**args)
    raise ValueError(kind)


def make_camera(mnv, spec):
    if "orbit" in spec:
        o = spec["orbit"]
        return mnv.orbit_camera(spec["width"], spec["height"], spec["fx"], o["radius"], o["azimuth"], o["elevation"])
    cam = mnv.Camera(spec.get("width", 256), spec.get("height", 256), spec.get("fx", 1111.0), spec.get("fy", -1.0),
                     spec.get("cx", -1.0), spec.get("cy", -1.0))
    if "center" in spec:
        cam.set_pose(spec["center"], spec["back"], spec.get("up", (0.0, 0.0, 1.0)))
    return cam


def make_options(mnv, spec):
    opt = mnv.RenderOptions.cli_defaults() if spec.get("base") == "cli" else mnv.RenderOptions.defaults()
    for k, v in spec.items():
        if k == "base":
            continue
        cur = getattr(opt, k)
        if hasattr(cur, "__len__"):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(opt, k, v)
    return opt


_CFG1_TREE = dict(kind="random", depth=4, basis_dim=1, refine_prob=0.6, empty_prob=0.5, sigma_max=30.0, coef_sd=1.5, seed=0)

CASES = {
    # cfg1: BASELINE.json configs[0] -- Camera ctor defaults, struct-default options, basis_minmax {0,0}
    "cfg1_sh1_d4": dict(tree=_CFG1_TREE, camera=dict(), options=dict(basis_minmax=(0, 0))),
    "sh4_d6": dict(tree=dict(kind="random", depth=6, basis_dim=4, refine_prob=0.55, empty_prob=0.6, sigma_max=40.0, coef_sd=1.0, seed=1),
                   camera=dict(width=200, height=160, fx=700.0, center=(-2.4, 1.1, 1.6), back=(-0.72, 0.33, 0.48)),
                   options=dict(base="cli")),
    "sh9_d7_aniso": dict(tree=dict(kind="random", depth=7, basis_dim=9, refine_prob=0.5, empty_prob=0.7, sigma_max=60.0, coef_sd=1.0, seed=2,
                                   offset=(0.5, 0.5, 0.5), scale=(0.5, 0.25, 0.125)),
                         camera=dict(width=240, height=136, fx=500.0, center=(-3.0, 2.0, 5.0), back=(-0.45, 0.3, 0.75)),
                         options=dict(base="cli", background_brightness=0.25)),
    "rgba_d5": dict(tree=dict(kind="random", depth=5, basis_dim=-1, fmt=0, refine_prob=0.6, empty_prob=0.5, sigma_max=25.0, seed=3),
                    camera=dict(width=192, height=192, fx=900.0), options=dict()),
    "depth_mode": dict(tree=_CFG1_TREE, camera=dict(), options=dict(render_depth=True)),
    "bbox_clipped": dict(tree=dict(kind="random", depth=5, basis_dim=4, refine_prob=0.6, empty_prob=0.4, sigma_max=30.0, seed=4),
                         camera=dict(width=160, height=160, fx=600.0),
                         options=dict(render_bbox=(0.2, 0.1, 0.3, 0.8, 0.9, 0.75))),
    "rot_dirs": dict(tree=dict(kind="random", depth=5, basis_dim=9, refine_prob=0.6, empty_prob=0.5, sigma_max=30.0, seed=5),
                     camera=dict(width=160, height=120, fx=500.0), options=dict(rot_dirs=(0.3, -0.2, 0.5))),
    "basis_minmax": dict(tree=dict(kind="random", depth=5, basis_dim=9, refine_prob=0.6, empty_prob=0.5, sigma_max=30.0, seed=6),
                         camera=dict(width=160, height=120, fx=500.0), options=dict(basis_minmax=(1, 5))),
    "camera_inside": dict(tree=dict(kind="random", depth=6, basis_dim=4, refine_prob=0.5, empty_prob=0.8, sigma_max=20.0, seed=7),
                          camera=dict(width=160, height=160, fx=120.0, center=(0.1, -0.2, 0.05), back=(0.6, 0.64, 0.48)),
                          options=dict(base="cli")),
    "ray_miss": dict(tree=_CFG1_TREE, camera=dict(width=64, height=64, fx=800.0, center=(0.0, 0.0, 5.0), back=(0.0, 0.6, -0.8)),
                     options=dict()),
    "sh16_d4": dict(tree=dict(kind="random", depth=4, basis_dim=16, refine_prob=0.6, empty_prob=0.5, sigma_max=30.0, seed=8),
                    camera=dict(width=128, height=128, fx=600.0), options=dict()),
    "sh25_d4": dict(tree=dict(kind="random", depth=4, basis_dim=25, refine_prob=0.6, empty_prob=0.5, sigma_max=30.0, seed=9),
                    camera=dict(width=128, height=128, fx=600.0), options=dict()),
    "shell_d7_sh9": dict(tree=dict(kind="shell", depth=7, basis_dim=9, radius=0.35, half_thickness=1.5 / 128, seed=0),
                         camera=dict(width=320, height=180, fx=266.0, orbit=dict(radius=2.6, azimuth=22.5, elevation=20.0)),
                         options=dict(base="cli")),
    "terrain_d7_aniso": dict(tree=dict(kind="terrain", depth=7, basis_dim=9, bricks_y=4, bricks_z=2, noise_cells=6, base=0.25, amplitude=0.35,
                                       thickness=1.5 / 128, scale=(0.5, 0.125, 0.125), seed=0),
                             camera=dict(width=256, height=144, fx=190.0, center=(2.2, 4.76, 2.75), back=(0.4, 0.79, 0.46), up=(1.0, 0.0, 0.0)),
                             options=dict(base="cli")),
    "thresholds": dict(tree=dict(kind="random", depth=5, basis_dim=4, refine_prob=0.6, empty_prob=0.3, sigma_max=80.0, seed=10),
                       camera=dict(width=128, height=96, fx=400.0),
                       options=dict(step_size=1e-3, sigma_thresh=5.0, stop_thresh=0.1, background_brightness=0.5)),
}

# BASELINE.json configs[2] stand-in: anisotropic multi-brick terrain ("merged Mega-NeRF octree"), oblique aerial camera
CFG3_TREE = dict(kind="terrain", depth=10, basis_dim=9, bricks_y=4, bricks_z=2, noise_cells=6, base=0.25, amplitude=0.35,
                 thickness=1.5 / 1024, scale=(0.5, 0.125, 0.125), seed=0)
CFG3_SMALL = dict(CFG3_TREE, depth=7, thickness=1.5 / 128)
# the size SURVEY.md 8(d) states for configs[2] / [3] (5 - 10 M chunks): the same terrain one level deeper with a gentler relief --
# 7,218,572 chunks (3.2 GB of voxel rows, 57.7 M voxels), far beyond L2 + MALL; what bench.py times as cfg3 / cfg4
CFG3_FULL = dict(CFG3_TREE, depth=11, amplitude=0.25, noise_cells=4, thickness=1.5 / 2048)


def cfg3_camera(mnv, pose=0, width=1920, height=1080, fx=1400.0):
    """Oblique aerial pose: world x is height (extent [-1, 1]), y and z span [-4, 4]."""
    import numpy as _np
    az = _np.deg2rad(22.5 * pose)
    center = _np.float32([2.2, 5.5 * _np.cos(az), 5.5 * _np.sin(az)])
    target = _np.float32([-0.2, 0.0, 0.0])
    back = (center - target) / _np.linalg.norm(center - target)
    return mnv.Camera(width, height, fx).set_pose(center, back.astype(_np.float32), (1.0, 0.0, 0.0))


def cfg3_cluster_grid(mnv):
    """The 4 x 2 sub-module grid over world y, z of the merged-octree stand-in (rt_core.cuh:541-549): one network per terrain brick."""
    g = mnv.ClusterGrid()
    g.grid_dim[0], g.grid_dim[1] = 4, 2
    for i, (lo, rng) in enumerate([(-1.0, 2.0), (-4.0, 8.0), (-4.0, 8.0)]):
        g.min_position[i], g.range[i] = lo, rng
    return g


# BASELINE.json configs[1]: depth-10 SH9 shell, 1920x1080, fx 1600, orbit radius 2.6, elevation 20
CFG2_TREE = dict(kind="shell", depth=10, basis_dim=9, radius=0.35, half_thickness=1.5 / 1024, sigma_lo=50.0, sigma_hi=400.0, seed=0)


def cfg2_camera(mnv, pose=0, width=1920, height=1080, fx=1600.0):
    return mnv.orbit_camera(width, height, fx, 2.6, 22.5 * pose, 20.0)


# A second distribution for the headline's kernel: cfg2's shell is opaque -- 4.5 dense samples in 20.9 steps per ray, every
# ray over after a handful of voxels -- while real PlenOctrees carry semi-transparent volume.  FOG: the same generator with a THICK shell
# (half-thickness 0.06 of the unit cube: 61 depth-9 voxels across) of thin density (sigma ~ U(5, 40): ~50 voxels to reach stop_thresh):
# 3,784,521 chunks (1.7 GB of voxel rows), 33 dense samples in 44 steps per ray (76 per ray that hits), same 16-pose orbit and camera.
FOG_TREE = dict(kind="shell", depth=9, basis_dim=9, radius=0.35, half_thickness=0.06, sigma_lo=5.0, sigma_hi=40.0, seed=0)


def bits(a):
    """uint32 view for bit-exact comparison of float32 arrays."""
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# The reference's live call shape, offscreen == false (cuda_renderer.cpp:141-142; renderer_kernel.cu:230-234,260-264,277-280): a depth
# attachment (per-pixel t_max) and an image under the volume.  name -> (base case, which inputs, seed)
ONSCREEN = {
    "onscreen_depth": ("shell_d7_sh9", ("tmax",), 101),           # clipped by a depth image only; background_brightness composite
    "onscreen_image": ("sh4_d6", ("image",), 102),                # composite over an image only; t_max = 1e9f
    "onscreen_both": ("sh9_d7_aniso", ("tmax", "image"), 103),    # what the viewer's render loop passes
    "onscreen_terrain": ("terrain_d7_aniso", ("tmax", "image"), 104),
}


def onscreen_inputs(name, cam):
    """Deterministic stand-ins for the two GL attachments: tmax_px [h][w] float32 -- world-space ray limits scattered around the
    camera's distance to the scene centre, 15 % "no mesh" (1e9f, the clear value), 5 % "mesh right in front of the camera" (0) --
    and rgba8_init [h][w][4] uint8 -- blocks + noise, alpha byte arbitrary.  Either is None when the case does not use it."""
    _, which, seed = ONSCREEN[name]
    rng = np.random.default_rng(seed)
    h, w = cam.height, cam.width
    c2w = np.array(list(cam.c.c2w), np.float64)
    dist = float(np.linalg.norm(c2w[9:12]))
    tmax = (dist * rng.uniform(0.55, 1.35, size=(h, w))).astype(np.float32)
    u = rng.uniform(size=(h, w))
    tmax[u < 0.15] = np.float32(1e9)
    tmax[u > 0.95] = np.float32(0.0)
    yy, xx = np.mgrid[0:h, 0:w]
    image = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    block = (((yy // 16) + (xx // 16)) % 2).astype(bool)
    image[block, :3] = (image[block, :3] // 4 + np.uint8(180))          # bright blocks
    image[(yy % 37 == 0) | (xx % 41 == 0), :3] = (0, 255, 1)            # saturated and near-zero channel values
    return (tmax if "tmax" in which else None), (np.ascontiguousarray(image) if "image" in which else None)

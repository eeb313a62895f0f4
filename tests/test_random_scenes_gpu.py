"""Seeded random scenes: trees, cameras and options nobody chose by hand, every HIP kernel of the plain frame against the C oracle bit for
bit (float RGBA by uint32 equality, RGBA8 by byte equality, tracker rows by equality).  The hand-made cases of tests/cases.py follow the
reference's fixture list; these widen them to the combinations that list does not name -- odd frame sizes, cameras inside
and behind the volume, rotated SH frames together with clipped boxes and basis windows, anisotropic scales on every format, thresholds
from nothing-is-skipped to almost-everything-is."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

N_SCENES = 160


def _scene(seed):
    rng = np.random.default_rng(1000 + seed)
    basis = int(rng.choice([-1, 1, 4, 9, 16, 25], p=[0.15, 0.15, 0.2, 0.3, 0.1, 0.1]))
    depth = int(rng.integers(2, 8 if basis <= 9 else 6))
    tree = dict(kind="random", depth=depth, basis_dim=basis, refine_prob=float(rng.uniform(0.3, 0.7)), empty_prob=float(rng.uniform(0.2, 0.9)),
                sigma_max=float(10 ** rng.uniform(1.0, 2.4)), coef_sd=float(rng.uniform(0.3, 2.5)), seed=int(rng.integers(0, 1 << 30)))
    if basis == -1:
        tree["fmt"] = 0
    if rng.random() < 0.5:  # anisotropic: the world box is offset / scale away from the unit cube
        tree["scale"] = tuple(float(x) for x in 2.0 ** -rng.integers(0, 4, 3))
        tree["offset"] = tuple(float(x) for x in rng.uniform(0.3, 0.7, 3))
    # camera: anywhere from inside the volume to far outside, looking at a point near the box -- or, sometimes, anywhere
    sc = np.float32(tree.get("scale", (0.5, 0.5, 0.5)))
    off = np.float32(tree.get("offset", (0.5, 0.5, 0.5)))
    box_c, box_r = (0.5 - off) / sc, 0.5 / sc  # world centre and half extent of the tree's box
    d = rng.normal(size=3)
    d /= np.linalg.norm(d)
    center = box_c + d * box_r * float(rng.choice([0.2, 0.9, 1.5, 3.0, 6.0]))
    target = box_c + rng.uniform(-0.4, 0.4, 3) * box_r
    back = center - target
    if rng.random() < 0.12:
        back = rng.normal(size=3)  # an arbitrary direction: most rays miss
    back /= np.linalg.norm(back)
    up = rng.normal(size=3)
    while abs(np.dot(up / np.linalg.norm(up), back)) > 0.95:
        up = rng.normal(size=3)
    w, h = int(rng.integers(9, 200)), int(rng.integers(9, 160))
    fov = np.deg2rad(rng.uniform(25.0, 120.0))  # horizontal field of view
    cam = dict(width=w, height=h, fx=float(0.5 * w / np.tan(0.5 * fov)), center=tuple(map(float, center)), back=tuple(map(float, back)),
               up=tuple(map(float, up / np.linalg.norm(up))))
    if rng.random() < 0.3:
        cam["fy"] = cam["fx"] * float(rng.uniform(0.6, 1.5))
        cam["cx"], cam["cy"] = float(rng.uniform(0.2, 0.8) * w), float(rng.uniform(0.2, 0.8) * h)
    opt = dict(step_size=float(10 ** rng.uniform(-5, -2)), sigma_thresh=float(10 ** rng.uniform(-3, 0.8)), stop_thresh=float(10 ** rng.uniform(-4, -0.4)),
               background_brightness=float(rng.choice([0.0, 1.0, rng.uniform(0, 1)])))
    if rng.random() < 0.5:
        opt["base"] = "cli"
    if rng.random() < 0.15:
        opt["render_depth"] = True
    if rng.random() < 0.3:
        lo = rng.uniform(0.0, 0.6, 3)
        opt["render_bbox"] = tuple(map(float, np.concatenate([lo, lo + rng.uniform(0.05, 0.4, 3)])))
    if basis > 1 and rng.random() < 0.4:
        opt["rot_dirs"] = tuple(map(float, rng.normal(size=3) * rng.choice([0.2, 1.0, 3.0])))
    if basis > 1 and rng.random() < 0.4:
        a = int(rng.integers(0, basis))
        opt["basis_minmax"] = (a, int(rng.integers(a, basis)))
    elif basis >= 1:
        opt["basis_minmax"] = (0, basis - 1)
    return dict(tree=tree, camera=cam, options=opt), rng


@pytest.mark.parametrize("seed", range(N_SCENES))
def test_random_scene_bit_exact(mnv, orc, torch_gpu, seed):
    torch = torch_gpu
    spec, rng = _scene(seed)
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    v = tree.host_view()
    with_trackers = seed % 2 == 1 and not spec["options"].get("render_depth")
    sc = rng.integers(0, 14, size=(v.capacity, 8)).astype(np.int16) if with_trackers else None
    if with_trackers:
        opt.max_depth = int(rng.integers(1, spec["tree"]["depth"] + 2))
        opt.max_sample_count = int(rng.integers(1, 14))
    ref = orc.render(orc.tree_from_view(v, sample_counts=sc), cam.c, opt, want_rgba8=True, want_trackers=with_trackers)
    tree.move_to_device(need_sample_counts=with_trackers)
    h, w = cam.height, cam.width
    what = f"scene {seed}: {spec}"

    def fresh():
        return (torch.full((h, w, 4), float("nan"), dtype=torch.float32, device="cuda"), torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda"))

    def same_frame(rgba, rgba8, who):
        got = rgba.cpu().numpy()
        assert not np.isnan(got).any(), f"{who} left pixels unwritten; {what}"
        diff = cases.bits(got) != cases.bits(ref["rgba"])
        assert not diff.any(), f"{who}: {int(diff.any(axis=-1).sum())} pixels differ, max|d| = {np.abs(got - ref['rgba']).max():.3e}; {what}"
        assert np.array_equal(rgba8.cpu().numpy(), ref["rgba8"]), f"{who}: RGBA8; {what}"

    # the walking kernel on the reference's arrays: with the per-launch table and without
    for min_rays in (0, -1):
        rgba, rgba8 = fresh()
        mnv.render_voxels(tree.device_view(), cam, opt, rgba=rgba, rgba8=rgba8, table_min_rays=min_rays)
        torch.cuda.synchronize()
        same_frame(rgba, rgba8, f"march_ref_layout_kernel (table min rays {min_rays})")
    # the tuned kernel on the packed layout
    rgba, rgba8 = fresh()
    mnv.render_voxels_accel(tree.accel, cam, opt, rgba=rgba, rgba8=rgba8)
    torch.cuda.synchronize()
    same_frame(rgba, rgba8, "march_accel_kernel")
    if seed % 4 == 2:
        # the reference's live call shape on a scene nobody chose: a depth image scattered around the camera-to-box distance (some pixels "no
        # mesh", some "mesh at the lens") and / or a random image under the volume, both kernels
        kind = (seed // 4) % 3
        dist = float(np.linalg.norm(np.float64(spec["camera"]["center"])) + 1e-3)
        tmax = (dist * rng.uniform(0.0, 2.0, size=(h, w))).astype(np.float32) if kind != 1 else None
        if tmax is not None:
            u = rng.uniform(size=(h, w))
            tmax[u < 0.2] = np.float32(1e9)
            tmax[u > 0.93] = np.float32(0.0)
        image = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8) if kind != 0 else None
        want = orc.render(orc.tree_from_view(v, sample_counts=sc), cam.c, opt, want_rgba8=True, tmax_px=tmax, rgba8_init=image)
        t_dev = None if tmax is None else torch.from_numpy(tmax).cuda()
        i_dev = None if image is None else torch.from_numpy(image).cuda()
        for who in ("ref_layout", "accel"):
            rgba, rgba8 = fresh()
            if who == "accel":
                mnv.render_voxels_accel(tree.accel, cam, opt, rgba=rgba, rgba8=rgba8, tmax_px=t_dev, rgba8_init=i_dev)
            else:
                mnv.render_voxels(tree.device_view(), cam, opt, rgba=rgba, rgba8=rgba8, tmax_px=t_dev, rgba8_init=i_dev)
            torch.cuda.synchronize()
            got = rgba.cpu().numpy()
            assert np.array_equal(cases.bits(got), cases.bits(want["rgba"])), f"{who} with on-screen inputs (kind {kind}): {int((cases.bits(got) != cases.bits(want['rgba'])).any(axis=-1).sum())} pixels differ; {what}"
            assert np.array_equal(rgba8.cpu().numpy(), want["rgba8"]), f"{who} with on-screen inputs: RGBA8; {what}"
    if not with_trackers:
        return
    sc_dev = torch.from_numpy(sc).cuda()
    for who in ("accel", "ref_layout"):
        rgba, rgba8 = fresh()
        split = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
        sample = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
        if who == "accel":
            mnv.render_voxels_accel_track(tree.accel, cam, opt, rgba=rgba, rgba8=rgba8, split_track=split, sample_track=sample, sample_counts=sc_dev)
        else:
            dv = tree.device_view()
            dv.sample_counts = sc_dev.data_ptr()
            mnv.render_voxels(dv, cam, opt, rgba=rgba, rgba8=rgba8, split_track=split, sample_track=sample)
        torch.cuda.synchronize()
        same_frame(rgba, rgba8, f"{who} with trackers")
        assert np.array_equal(split.cpu().numpy().reshape(-1, 3), ref["split"].reshape(-1, 3)), f"{who}: split tracker; {what}"
        assert np.array_equal(sample.cpu().numpy().reshape(-1, 3), ref["sample"].reshape(-1, 3)), f"{who}: sample tracker; {what}"


@pytest.mark.parametrize("seed", range(12))
def test_random_deep_scene_bit_exact(mnv, orc, torch_gpu, seed):
    """Random trees DEEP enough to carry brick records (depth 10 - 12 under a level-8 grid: leaves two to four levels below it, node words
    below the records) and mostly empty, so that rays reach those levels -- random row formats, cameras, options, a negative sigma_thresh
    now and then (the launch then takes the node words), on-screen inputs on every third: the tuned kernel against the oracle bit for bit."""
    torch = torch_gpu
    spec, rng = _scene(7000 + seed)
    basis = int(rng.choice([-1, 1, 4, 9]))
    spec["tree"].update(depth=int(rng.integers(10, 13)), basis_dim=basis, refine_prob=float(rng.uniform(0.38, 0.43)), empty_prob=float(rng.uniform(0.88, 0.95)),
                        sigma_max=float(10 ** rng.uniform(1.0, 2.0)))
    spec["tree"].pop("fmt", None)
    if basis == -1:
        spec["tree"]["fmt"] = 0
    spec["options"].pop("rot_dirs", None)
    spec["options"].pop("render_bbox", None)
    spec["options"].update(basis_minmax=(0, max(basis - 1, 0)), step_size=float(10 ** rng.uniform(-5, -3.5)), stop_thresh=float(10 ** rng.uniform(-4, -2)),
                           sigma_thresh=-0.25 if seed % 4 == 3 else float(10 ** rng.uniform(-3, -1)))
    # a camera that looks AT the box from just outside it or from inside: the rays cross the levels the records cover
    sc, off = np.float32(spec["tree"].get("scale", (0.5, 0.5, 0.5))), np.float32(spec["tree"].get("offset", (0.5, 0.5, 0.5)))
    box_c, box_r = (0.5 - off) / sc, 0.5 / sc
    d = rng.normal(size=3)
    d /= np.linalg.norm(d)
    center = box_c + d * box_r * float(rng.choice([0.3, 1.2, 2.0]))
    back = center - (box_c + rng.uniform(-0.2, 0.2, 3) * box_r)
    w, h = int(rng.integers(96, 256)), int(rng.integers(64, 192))
    spec["camera"] = dict(width=w, height=h, fx=float(0.5 * w / np.tan(0.5 * np.deg2rad(rng.uniform(40.0, 90.0)))), center=tuple(map(float, center)),
                          back=tuple(map(float, back / np.linalg.norm(back))), up=spec["camera"]["up"])
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    v = tree.host_view()
    h, w = cam.height, cam.width
    tmax = image = None
    if seed % 3 == 0:
        dist = float(np.linalg.norm(np.float64(spec["camera"]["center"])) + 1e-3)
        tmax = (dist * rng.uniform(0.0, 2.0, size=(h, w))).astype(np.float32)
        tmax[rng.uniform(size=(h, w)) < 0.25] = np.float32(1e9)
        image = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    want = orc.render(orc.tree_from_view(v), cam.c, opt, want_rgba8=True, tmax_px=tmax, rgba8_init=image)
    what = f"deep scene {seed}: {spec}, {v.capacity} chunks, steps/ray {want['counters'].steps / max(1, want['counters'].rays):.1f}, levels/step {want['counters'].levels / max(1, want['counters'].steps):.1f}"
    tree.move_to_device()
    info = mnv.accel_info(tree.accel)
    assert info["brick_levels"] == 2, (info, spec["tree"])
    assert want["counters"].steps > 2 * want["counters"].rays and want["counters"].levels > 1.5 * want["counters"].steps, what
    rgba = torch.full((h, w, 4), float("nan"), dtype=torch.float32, device="cuda")
    rgba8 = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    mnv.render_voxels_accel(tree.accel, cam, opt, rgba=rgba, rgba8=rgba8, tmax_px=None if tmax is None else torch.from_numpy(tmax).cuda(),
                            rgba8_init=None if image is None else torch.from_numpy(image).cuda())
    torch.cuda.synchronize()
    got = rgba.cpu().numpy()
    assert not np.isnan(got).any(), what
    assert np.array_equal(cases.bits(got), cases.bits(want["rgba"])), f"{int((cases.bits(got) != cases.bits(want['rgba'])).any(axis=-1).sum())} pixels differ; {what}"
    assert np.array_equal(rgba8.cpu().numpy(), want["rgba8"]), what


@pytest.mark.parametrize("seed", range(80))
def test_random_scene_guided_samples_bit_exact(mnv, orc, torch_gpu, seed):
    """The sample-emitting march (rt_core.cuh:418-576) of both kernels on the same kind of scene: counts, sample rows (z, world position,
    view direction, embedding), cluster ids from a random cluster grid that covers the world box only partly, tracker rows."""
    torch = torch_gpu
    spec, rng = _scene(500 + seed)
    spec["options"].pop("render_depth", None)
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    need_viewdir, embedding, quota = bool(rng.integers(0, 2)), int(rng.choice([-1, -1, 0, 3, 11])), int(rng.integers(1, 20))
    opt.need_viewdir, opt.appearance_embedding, opt.max_guided_samples = need_viewdir, embedding, quota
    opt.max_depth, opt.max_sample_count = int(rng.integers(1, spec["tree"]["depth"] + 2)), int(rng.integers(1, 14))
    dim = 4 + (3 if need_viewdir else 0) + (1 if embedding != -1 else 0)
    sc_w = np.float32(spec["tree"].get("scale", (0.5, 0.5, 0.5)))
    off_w = np.float32(spec["tree"].get("offset", (0.5, 0.5, 0.5)))
    lo_w, ext_w = -off_w / sc_w, 1.0 / sc_w  # the tree's world box
    grid = mnv.ClusterGrid()
    grid.grid_dim[0], grid.grid_dim[1] = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    for i in range(3):
        grid.min_position[i] = float(lo_w[i] + rng.uniform(-0.1, 0.3) * ext_w[i])
        grid.range[i] = float(rng.uniform(0.5, 1.1) * ext_w[i])
    v = tree.host_view()
    sc = rng.integers(0, 14, size=(v.capacity, 8)).astype(np.int16)
    # every third scene in the reference's live call shape: a depth image scattered around the camera-to-box distance limits the rays
    tmax = None
    if seed % 3 == 1:
        dist = float(np.linalg.norm(np.float64(spec["camera"]["center"])) + 1e-3)
        tmax = (dist * rng.uniform(0.0, 2.0, size=(cam.height, cam.width))).astype(np.float32)
        u = rng.uniform(size=(cam.height, cam.width))
        tmax[u < 0.2] = np.float32(1e9)
        tmax[u > 0.93] = np.float32(0.0)
    ref = orc.get_samples(orc.tree_from_view(v, sample_counts=sc), cam.c, opt, grid, dim, tmax_px=tmax)
    tree.move_to_device(need_sample_counts=True)
    sc_dev = torch.from_numpy(sc).cuda()
    d_tmax = None if tmax is None else torch.from_numpy(tmax).cuda()
    n = cam.width * cam.height
    what = f"scene {500 + seed}: {spec}, viewdir {need_viewdir}, embedding {embedding}, quota {quota}, depth image {tmax is not None}"
    k = np.arange(quota)[None, :] < ref["num_samples"][:, None]  # emitted rows; the rest keep the caller's fill on both sides
    for who in ("ref_layout", "accel"):
        num = torch.zeros(n, dtype=torch.int16, device="cuda")
        samples = torch.full((n, quota, dim), -1.0, dtype=torch.float32, device="cuda")
        clusters = torch.full((n, quota), -1, dtype=torch.int16, device="cuda")
        split = torch.full((n, 3), -1.0, dtype=torch.float32, device="cuda")
        sample = torch.full((n, 3), -1.0, dtype=torch.float32, device="cuda")
        if who == "accel":
            mnv.get_samples_from_voxels_accel(tree.accel, cam, opt, num, samples, clusters, grid, split_track=split, sample_track=sample, sample_counts=sc_dev,
                                              tmax_px=d_tmax)
        else:
            dv = tree.device_view()
            dv.sample_counts = sc_dev.data_ptr()
            mnv.get_samples_from_voxels(dv, cam, opt, num, samples, clusters, grid, split_track=split, sample_track=sample, tmax_px=d_tmax)
        torch.cuda.synchronize()
        assert np.array_equal(num.cpu().numpy(), ref["num_samples"]), f"{who}: counts; {what}"
        got_s, got_c = samples.cpu().numpy(), clusters.cpu().numpy()
        assert np.array_equal(cases.bits(got_s[k]), cases.bits(ref["samples"][k])), f"{who}: sample rows; {what}"
        assert np.array_equal(got_c[k], ref["cluster_indices"][k]), f"{who}: cluster ids; {what}"
        assert np.array_equal(split.cpu().numpy(), ref["split"]) and np.array_equal(sample.cpu().numpy(), ref["sample"]), f"{who}: trackers; {what}"

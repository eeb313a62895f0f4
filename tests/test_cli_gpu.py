"""The C++ batch driver (`mnv_render`, the VolumeRenderer role of reference
include/renderer/renderer.hpp:9-39 + the CLI flags of src/opts.cpp:17-32 / main.cpp:491-505)
end to end on the GPU: .npz -> N3Tree::open -> VolumeRenderer::set/resize/render -> files,
compared bit for bit with the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

import cases
import hooks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "mega-nerf-viewer_amd", "mnv_render")


def test_mnv_render_cli_matches_oracle(mnv, orc, torch_gpu, tmp_path):
    assert os.path.exists(EXE), "mnv_render not built"
    tree = cases.make_tree(mnv, cases.CASES["sh9_d7_aniso"]["tree"])
    npz = str(tmp_path / "scene.npz")
    tree.save_npz(npz)
    w, h = 200, 144
    center, back = (-3.0, 2.0, 5.0), (-0.45, 0.3, 0.75)
    out = str(tmp_path / "frame")
    cmd = [EXE, npz, "-w", str(w), "-h", str(h), "--fx", "450", "--bg", "0.25", "-s", "2e-4", "-e", "0.02", "-a", "0.5",
           "--center", ",".join(map(str, center)), "--back", ",".join(map(str, back)), "--out", out, "--raw", "--frames", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "HIP gfx950" in r.stdout
    got = np.fromfile(out + "_0000.f32", dtype=np.float32).reshape(h, w, 4)
    again = np.fromfile(out + "_0001.f32", dtype=np.float32).reshape(h, w, 4)
    cam = mnv.Camera(w, h, 450.0).set_pose(center, back)
    opt = mnv.RenderOptions.cli_defaults()
    opt.background_brightness, opt.step_size, opt.stop_thresh, opt.sigma_thresh = 0.25, 2e-4, 0.02, 0.5
    opt.basis_minmax[0], opt.basis_minmax[1] = 0, 8  # VolumeRenderer::set, cuda_renderer.cpp:511-512
    ref = orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt, want_rgba8=True)
    assert np.array_equal(cases.bits(got), cases.bits(ref["rgba"]))
    # frame 2: render() re-runs Camera::_update (cuda_renderer.cpp:79), which re-normalises v_back
    cam2 = mnv.Camera(w, h, 450.0).set_pose(center, tuple(cam.c2w[6:9]))
    ref2 = orc.render(orc.tree_from_view(tree.host_view()), cam2.c, opt)
    assert np.array_equal(cases.bits(again), cases.bits(ref2["rgba"]))
    ppm = open(out + "_0000.ppm", "rb").read()
    header = f"P6\n{w} {h}\n255\n".encode()
    assert ppm.startswith(header)
    rgb = np.frombuffer(ppm[len(header):], np.uint8).reshape(h, w, 3)
    assert np.array_equal(rgb, ref["rgba8"][..., :3])


def test_mnv_render_multi_gpu_mode_with_one_rank(mnv, orc, torch_gpu, tmp_path):
    """`mnv_render --gpus 1`: the multi-GPU mode of the C++ host (one forked process per GPU, interleaved macro-tile partition,
    one batched launch per rank and 64 frames, RCCL gather through mnv_gather_tiles, un-permute on rank 0) run with the one rank a
    one-GPU box allows.  70 orbit frames = two batches through the ring; every file equals the single-GPU path's, byte for byte,
    and frame 0 equals the oracle."""
    tree = cases.make_tree(mnv, cases.CASES["sh9_d7_aniso"]["tree"])
    npz = str(tmp_path / "scene.npz")
    tree.save_npz(npz)
    w, h, frames = 200, 144, 70
    center, back = (-3.0, 2.0, 5.0), (-0.45, 0.3, 0.75)
    common = [EXE, npz, "-w", str(w), "-h", str(h), "--fx", "450", "--bg", "0.25", "--center", ",".join(map(str, center)),
              "--back", ",".join(map(str, back)), "--raw", "--frames", str(frames), "--orbit", "3.5"]
    one, dist = str(tmp_path / "one"), str(tmp_path / "dist")
    r1 = subprocess.run(common + ["--out", one], capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr + r1.stdout
    r2 = subprocess.run(common + ["--out", dist, "--gpus", "1"], capture_output=True, text=True, timeout=600)   # the shipped binary over real RCCL
    assert r2.returncode == 0, r2.stderr + r2.stdout
    assert "x 1 (RCCL" in r2.stdout and "macro tiles" in r2.stdout
    for f in range(frames):
        for ext in ("f32", "ppm"):
            a = open(f"{one}_{f:04d}.{ext}", "rb").read()
            b = open(f"{dist}_{f:04d}.{ext}", "rb").read()
            assert a == b, (f, ext)
    cam = mnv.Camera(w, h, 450.0).set_pose(center, back)
    opt = mnv.RenderOptions.cli_defaults()
    opt.background_brightness = 0.25
    opt.basis_minmax[0], opt.basis_minmax[1] = 0, 8
    ref = orc.render(orc.tree_from_view(tree.host_view()), cam.c, opt)
    got = np.fromfile(dist + "_0000.f32", dtype=np.float32).reshape(h, w, 4)
    assert np.array_equal(cases.bits(got), cases.bits(ref["rgba"]))
    # a rank that fails takes the run down with a non-zero status instead of hanging its peers
    r3 = subprocess.run([EXE, str(tmp_path / "missing.npz"), "--gpus", "1"], capture_output=True, text=True, timeout=120)
    assert r3.returncode != 0 and "rank 0" in r3.stderr
    if torch_gpu.cuda.device_count() == 1:
        # two ranks need two GPUs: rank 1 finds no device (or RCCL refuses the duplicate), rank 0 is taken down with it
        r4 = subprocess.run(common + ["--gpus", "2"], capture_output=True, text=True, timeout=180)
        assert r4.returncode != 0 and "rank" in r4.stderr
    # refinement across ranks needs the networks (test_mnv_render_refinement_across_ranks runs it)
    r5 = subprocess.run(common + ["--gpus", "1", "--use_splitting"], capture_output=True, text=True, timeout=120)
    assert r5.returncode != 0 and "--model_path" in r5.stderr


@pytest.mark.parametrize("world", [2, 3, 8])
def test_mnv_render_multi_gpu_mode_with_several_ranks_on_one_gpu(mnv, torch_gpu, tmp_path, fake_rccl, world):
    """`mnv_render --gpus N` with N > 1 on a one-GPU box: the ranks share the device (MNV_RANKS_SHARE_GPU) and a host-staged stand-in
    takes RCCL's place (MNV_RCCL_LIBRARY=tests/shim/fake_rccl.cpp; RCCL refuses two ranks on one device).  Everything of ours runs as
    it would on N GPUs -- fork per rank, rendezvous through the shared page, communicator per rank, interleaved partition with the
    root-relieving deal, one batched launch per rank, mnv_gather_tiles with world - 1 receives on the root and one send elsewhere, the
    ring of two slots, un-permute, file output -- and every frame equals the single-GPU path's byte for byte."""
    tree = cases.make_tree(mnv, cases.CASES["sh9_d7_aniso"]["tree"])
    npz = str(tmp_path / "scene.npz")
    tree.save_npz(npz)
    w, h, frames = 328, 200, 70          # 70 frames: two batches through the ring; 328 x 200: ragged macro tiles on both edges
    common = [EXE, npz, "-w", str(w), "-h", str(h), "--fx", "450", "--bg", "0.25", "--center", "-3.0,2.0,5.0", "--back", "-0.45,0.3,0.75",
              "--raw", "--frames", str(frames), "--orbit", "3.5"]
    one, dist = str(tmp_path / "one"), str(tmp_path / "dist")
    r1 = subprocess.run(common + ["--out", one], capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr + r1.stdout
    env = dict(os.environ, MNV_RCCL_LIBRARY=fake_rccl, MNV_RANKS_SHARE_GPU="1")
    exe = hooks.HOOKS_EXE  # the build that honours the two variables
    r2 = subprocess.run([exe] + common[1:] + ["--out", dist, "--gpus", str(world), "--reserve_cus", "0"], capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stderr + r2.stdout
    assert f"x {world} (RCCL 29999)" in r2.stdout      # the stand-in's version number: this run did not touch RCCL
    for f in range(frames):
        for ext in ("f32", "ppm"):
            a = open(f"{one}_{f:04d}.{ext}", "rb").read()
            b = open(f"{dist}_{f:04d}.{ext}", "rb").read()
            assert a == b, (f, ext)


@pytest.mark.parametrize("world", [1, 3])
def test_mnv_render_guided_sampling_across_ranks(mnv, torch_gpu, tmp_path, fake_rccl, world):
    """`mnv_render --gpus N --model_path M --use_guided_sampling`: every rank runs the fused guided-sampling kernel (march + networks +
    composite) on its macro tiles (mnv_render_guided_fused_part), the tiles are gathered and un-permuted as for plain frames; the frames
    equal the single-GPU guided frames byte for byte.  world 1 goes through RCCL itself, world 3 shares the GPU over the stand-in."""
    import mlp_cases
    from test_renderer_refine_gpu import make_grid

    tree = cases.make_tree(mnv, cases.CASES["sh9_d7_aniso"]["tree"])
    dd = tree.host_view().data_dim
    npz = str(tmp_path / "scene.npz")
    tree.save_npz(npz)
    desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=dd + 1)
    g = make_grid(mnv)
    model = str(tmp_path / "model.npz")
    np.savez(model, mlp_desc=np.array([6, 4, 2, 0, 0, 0, 64, 2, dd + 1], np.int32), mlp_center=np.zeros(3, np.float32),
             mlp_inv_extent=np.ones(3, np.float32), mlp_params=mlp_cases.make_params(mnv, desc, seed=21), grid_dim=np.array(list(g.grid_dim), np.int64),
             min_position=np.array(list(g.min_position), np.float32), max_position=np.array([g.min_position[i] + g.range[i] for i in range(3)], np.float32))
    w, h, frames = 328, 200, 5
    common = [EXE, npz, "-w", str(w), "-h", str(h), "--fx", "450", "--bg", "0.25", "--center", "-3.0,2.0,5.0", "--back", "-0.45,0.3,0.75", "--raw",
              "--frames", str(frames), "--orbit", "7", "--model_path", model, "--use_guided_sampling", "-z", "24"]
    one, dist = str(tmp_path / "one"), str(tmp_path / "dist")
    r1 = subprocess.run(common + ["--out", one], capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0 and "guided samples" in r1.stdout, r1.stderr + r1.stdout
    env, exe = dict(os.environ), EXE
    if world > 1:
        env.update(MNV_RCCL_LIBRARY=fake_rccl, MNV_RANKS_SHARE_GPU="1")
        exe = hooks.HOOKS_EXE  # the build that honours the two variables
    r2 = subprocess.run([exe] + common[1:] + ["--out", dist, "--gpus", str(world), "--reserve_cus", "0"], capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stderr + r2.stdout
    plain = str(tmp_path / "plain")
    r3 = subprocess.run([a for a in common if a not in ("--use_guided_sampling",)] + ["--out", plain], capture_output=True, text=True, timeout=600)
    assert r3.returncode == 0
    for f in range(frames):
        a = open(f"{one}_{f:04d}.f32", "rb").read()
        assert a == open(f"{dist}_{f:04d}.f32", "rb").read(), f
        assert open(f"{one}_{f:04d}.ppm", "rb").read() == open(f"{dist}_{f:04d}.ppm", "rb").read(), f
        assert a != open(f"{plain}_{f:04d}.f32", "rb").read()     # the networks' colours, not the tree's
    if world == 1:
        # --guided_in_flight: the same frames, three in flight on HIP streams; the sample counts arrive when a frame is written
        flight = str(tmp_path / "flight")
        r4 = subprocess.run(common + ["--out", flight, "--guided_in_flight"], capture_output=True, text=True, timeout=600)
        assert r4.returncode == 0 and "guided samples: in flight" in r4.stdout and "(3 in flight" in r4.stdout, r4.stderr + r4.stdout
        counts = lambda out: sorted(re.findall(r"frame (\d+):.*guided samples (\d+)", out))
        assert counts(r4.stdout) == counts(r1.stdout) and len(counts(r1.stdout)) == frames
        for f in range(frames):
            assert open(f"{one}_{f:04d}.f32", "rb").read() == open(f"{flight}_{f:04d}.f32", "rb").read(), f


@pytest.mark.parametrize("world,guided", [(1, False), (3, False), (3, True), (8, True)])
def test_mnv_render_refinement_across_ranks(mnv, torch_gpu, tmp_path, fake_rccl, world, guided):
    """`mnv_render --gpus N --use_splitting [--use_guided_sampling]` (BASELINE.json configs[4] on several GPUs): the ranks refine one scene in
    lock step -- each marches its macro tiles (pixels, tracker rows, visit marks), the tracker rows and the marks are all-gathered, every
    rank applies the same splits / resamples / prunes to its replica.  Checked against the single-GPU run of the same camera path: every
    frame byte for byte, the refined tree array for array, every rank's replica equal to rank 0's, and the run includes prunes (visit
    marks merged across ranks).  world 1 goes through RCCL itself, the others share the GPU over the transport stand-in."""
    import mlp_cases
    from test_renderer_refine_gpu import check_tree_links, make_grid

    spec = cases.CASES["sh4_d6"]
    tree = cases.make_tree(mnv, spec["tree"])
    cap0, dd = tree.capacity, tree.host_view().data_dim
    npz = str(tmp_path / "scene.npz")
    tree.save_npz(npz)
    desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=dd + 1)
    g = make_grid(mnv)
    model = str(tmp_path / "model.npz")
    np.savez(model, mlp_desc=np.array([6, 4, 2, 0, 0, 0, 64, 2, dd + 1], np.int32), mlp_center=np.zeros(3, np.float32),
             mlp_inv_extent=np.ones(3, np.float32), mlp_params=mlp_cases.make_params(mnv, desc, seed=21), grid_dim=np.array(list(g.grid_dim), np.int64),
             min_position=np.array(list(g.min_position), np.float32), max_position=np.array([g.min_position[i] + g.range[i] for i in range(3)], np.float32))
    w, h, frames = 328, 200, int(os.environ.get("MNV_SOAK_FRAMES", "14"))   # MNV_SOAK_FRAMES: longer runs by hand
    common = [EXE, npz, "-w", str(w), "-h", str(h), "--fx", "700", "--bg", "1.0", "--center", "-3.55,0,3.55", "--model_path", model, "--use_splitting",
              "-x", "64", "-v", "4", "--max_depth", "8", "--max_sample_count", "64", "--seed", "5", "-c", str(cap0 + 330), "--frames", str(frames),
              "--orbit", "4", "--raw"] + (["--use_guided_sampling", "-z", "24"] if guided else [])
    one, dist = str(tmp_path / "one"), str(tmp_path / "dist")
    r1 = subprocess.run(common + ["--out", one, "--save_tree", one + ".npz"], capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr + r1.stdout
    assert "added" in r1.stdout and "pruned" in r1.stdout, r1.stdout   # the path grows the tree until it is full, then prunes
    env, exe = dict(os.environ, MNV_SAVE_EVERY_RANK="1"), EXE
    if world > 1:
        env.update(MNV_RCCL_LIBRARY=fake_rccl, MNV_RANKS_SHARE_GPU="1")
        exe = hooks.HOOKS_EXE  # the build that honours the two variables
    r2 = subprocess.run([exe] + common[1:] + ["--out", dist, "--save_tree", dist + ".npz", "--gpus", str(world)], capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stderr + r2.stdout
    # the same decisions frame by frame (candidates, splits, resamples, prunes, capacities)
    strip = lambda out: [re.sub(r"  guided samples \d+", "", ln) for ln in out.splitlines() if ln.startswith("frame ")]
    assert strip(r1.stdout) == strip(r2.stdout)
    for f in range(frames):
        assert open(f"{one}_{f:04d}.f32", "rb").read() == open(f"{dist}_{f:04d}.f32", "rb").read(), f
        assert open(f"{one}_{f:04d}.ppm", "rb").read() == open(f"{dist}_{f:04d}.ppm", "rb").read(), f
    ta, tb = mnv.N3Tree.open(one + ".npz"), mnv.N3Tree.open(dist + ".npz")
    assert ta.capacity == tb.capacity and ta.capacity != cap0
    for a, b in zip(ta.host_arrays(), tb.host_arrays()):
        assert np.array_equal(a, b)
    check_tree_links(tb.host_arrays()[1], tb.host_arrays()[2], tb.capacity)
    for r in range(1, world):
        tr = mnv.N3Tree.open(f"{dist}.npz.rank{r}.npz")
        for a, b in zip(tr.host_arrays(), tb.host_arrays()):
            assert np.array_equal(a, b)


def test_mnv_render_cli_errors(tmp_path, mnv, torch_gpu):
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stdout
    bad = tmp_path / "bad.npz"
    bad.write_bytes(b"garbage")
    r = subprocess.run([EXE, str(bad), "-w", "8", "-h", "8"], capture_output=True, text=True)
    assert r.returncode == 1 and "mnv_render:" in r.stderr
    # a missing file renders the background only (N == 0), as the reference does (n3tree.cpp:19-22)
    out = str(tmp_path / "bg")
    r = subprocess.run([EXE, str(tmp_path / "missing.npz"), "-w", "16", "-h", "8", "--bg", "0.5", "--out", out, "--raw"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    f = np.fromfile(out + "_0000.f32", dtype=np.float32).reshape(8, 16, 4)
    assert np.all(f[..., :3] == 0.5) and np.all(f[..., 3] == 0)


def test_mnv_render_cli_refinement(mnv, torch_gpu, tmp_path):
    """--model_path + --use_splitting: the CLI runs the same refinement loop as the mnv_renderer_* API (same seed ->
    the same refined tree, written by --save_tree)."""
    import mlp_cases
    from test_renderer_refine_gpu import check_tree_links, make_grid
    spec = cases.CASES["rgba_d5"]
    tree = cases.make_tree(mnv, spec["tree"])
    cap0, dd = tree.capacity, tree.host_view().data_dim
    npz = str(tmp_path / "scene.npz")
    tree.save_npz(npz)
    desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=dd + 1)
    params = mlp_cases.make_params(mnv, desc, seed=21)
    g = make_grid(mnv)
    model = str(tmp_path / "model.npz")
    np.savez(model, mlp_desc=np.array([6, 4, 2, 0, 0, 0, 64, 2, dd + 1], np.int32), mlp_center=np.zeros(3, np.float32),
             mlp_inv_extent=np.ones(3, np.float32), mlp_params=params, grid_dim=np.array(list(g.grid_dim), np.int64),
             min_position=np.array(list(g.min_position), np.float32),
             max_position=np.array([g.min_position[i] + g.range[i] for i in range(3)], np.float32))
    refined = str(tmp_path / "refined.npz")
    w = h = 192
    cmd = [EXE, npz, "-w", str(w), "-h", str(h), "--fx", "900", "--bg", "1.0", "--center", "-3.55,0,3.55", "--model_path", model, "--use_splitting",
           "-x", "64", "-v", "4", "--max_depth", "7", "--max_sample_count", "64", "--seed", "5", "-c", str(cap0 + 2000), "--frames", "3",
           "--save_tree", refined]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "frame 0: capacity" in r.stdout and "added" in r.stdout
    out_tree = mnv.N3Tree.open(refined)
    data, child, parent = out_tree.host_arrays()
    assert out_tree.capacity > cap0
    check_tree_links(child, parent, out_tree.capacity)

    rr = mnv.Renderer()
    rr.resize(w, h)
    tree2 = cases.make_tree(mnv, spec["tree"])
    rr.set(tree2, cap0 + 2000)
    rr.set_model(desc, params, g)
    rr.set_camera((-3.55, 0.0, 3.55), (-0.7071068, 0.0, 0.7071068), fx=900.0)
    rr.set_seed(5)
    o = rr.options
    o.background_brightness, o.step_size, o.stop_thresh, o.sigma_thresh = 1.0, 1e-4, 1e-2, 1e-2
    o.use_splitting, o.split_batch_size, o.samples_per_corner, o.max_depth, o.max_sample_count = True, 64, 4, 7, 64
    for _ in range(3):
        st = rr.render()
    rr.sync_tree()
    d2, c2, p2 = tree2.host_arrays()
    assert st["capacity"] == out_tree.capacity
    assert np.array_equal(c2, child) and np.array_equal(p2, parent) and np.array_equal(d2, data)

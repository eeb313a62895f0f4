"""mnv_render_guided_fused (BASELINE.json configs[4]: the per-sample network fused into the march kernel) against the
four-step path it replaces -- sample march on the accel, cumsum / mask compaction, mnv_query_submodules, CSR composite:
same march, same MFMA sequence, same composite arithmetic, so frames are compared bit for bit.  The four-step path itself is
pinned by tests/test_guided_gpu.py (kernels vs oracle vs the reference's device code) and tests/test_mlp_gpu.py."""
import numpy as np
import pytest

import cases
import mlp_cases
from test_renderer_refine_gpu import make_grid

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[2, 1], ids=["producer_consumer", "one_role"])
def fused_kernel(request, mnv, torch_gpu):
    """Every test of this module runs with each of the two kernels behind mnv_render_guided_fused* (mnv_accel_set_fused_kernel, handed to every
    accel the tests launch on by the binding's harness default, mnv.set_fused_kernel): wavefronts
    specialised into march producers and a network consumer (csrc/mnv_guided_fused2.h), and every wavefront in both roles
    (csrc/mnv_guided_fused.h).  The diagnostics buffer is on so that a spin-wait abandoned by the watchdog fails the test."""
    diag = torch_gpu.zeros(32, dtype=torch_gpu.int64, device="cuda")
    mnv.set_fused_kernel(request.param)
    _FUSED_CHOICE[0] = request.param
    mnv.set_fused_diag(diag if request.param == 2 else None)
    yield request.param
    torch_gpu.cuda.synchronize()
    mnv.set_fused_diag(None)
    mnv.set_fused_kernel(0)
    assert int(diag[15].item()) == 0, "a spin-wait of the producer / consumer kernel ran into its watchdog"


def four_step_frame(mnv, torch, tree, cam, opt, mlp, grid, max_g, dim, tmax_px=None):
    n_px = cam.width * cam.height
    num = torch.zeros(n_px, dtype=torch.int16, device="cuda")
    guided = torch.zeros((n_px, max_g, dim), dtype=torch.float32, device="cuda")
    guided[:, :, 0] = -1
    clusters = torch.zeros((n_px, max_g), dtype=torch.int16, device="cuda")
    mnv.get_samples_from_voxels_accel(tree.accel, cam, opt, num, guided, clusters, grid, tmax_px=tmax_px)
    offsets = torch.cumsum(num, 0)
    flat = guided.view(-1, dim)
    mask = flat[:, 0] >= 0
    valid, valid_clusters = flat[mask], clusters.view(-1)[mask]
    total = valid.shape[0]
    values = torch.zeros((max(total, 1), tree.host_view().data_dim + 1), dtype=torch.float32, device="cuda")
    if total:
        mlp.query(valid_clusters, valid[:, 1:].contiguous(), values, n=total)
    out = torch.empty((cam.height, cam.width, 4), dtype=torch.float32, device="cuda")
    out8 = torch.empty((cam.height, cam.width, 4), dtype=torch.uint8, device="cuda")
    mnv.render_nerf_results(tree.device_view(), cam, opt, values, valid[:, 0].contiguous(), offsets, rgba=out, rgba8=out8)
    torch.cuda.synchronize()
    return out.cpu().numpy(), out8.cpu().numpy(), total


@pytest.mark.parametrize("case,need_viewdir,n_emb,max_g,n_clusters", [
    ("rgba_d5", False, 0, 16, 6),
    ("sh9_d7_aniso", False, 0, 128, 6),
    ("sh4_d6", True, 0, 8, 6),
    ("shell_d7_sh9", True, 3, 128, 6),
    ("sh9_d7_aniso", False, 0, 3, 4),        # quota of 3 samples per ray; 4 networks for 6 grid cells: cells 4, 5 have no sub-module
    ("sh16_d4", False, 0, 32, 6),
])
def test_fused_frame_equals_the_four_step_path(mnv, torch_gpu, case, need_viewdir, n_emb, max_g, n_clusters):
    torch = torch_gpu
    spec = cases.CASES[case]
    tree = cases.make_tree(mnv, spec["tree"])
    v = tree.host_view()
    tree.move_to_device()
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.basis_minmax[0], opt.basis_minmax[1] = 0, max(v.basis_dim - 1, 0)
    opt.max_guided_samples = max_g
    opt.need_viewdir = need_viewdir
    opt.appearance_embedding = 1 if n_emb else -1
    desc = mnv.mlp_desc(n_clusters=n_clusters, pos_octaves=4, dir_octaves=2, need_viewdir=need_viewdir, n_embeddings=n_emb, embedding_dim=8 if n_emb else 0,
                        hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=21))
    grid = make_grid(mnv)
    dim = 4 + (3 if need_viewdir else 0) + (1 if n_emb else 0)
    ref, ref8, total = four_step_frame(mnv, torch, tree, cam, opt, mlp, grid, max_g, dim)
    assert total > 0
    out = torch.full((cam.height, cam.width, 4), float("nan"), dtype=torch.float32, device="cuda")
    out8 = torch.zeros((cam.height, cam.width, 4), dtype=torch.uint8, device="cuda")
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    mnv.render_guided_fused(tree.accel, cam, opt, mlp, grid, rgba=out, rgba8=out8, sample_counter=counter)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert int(counter.item()) == total
    assert np.array_equal(cases.bits(got), cases.bits(ref)), float(np.nanmax(np.abs(got - ref)))
    assert np.array_equal(out8.cpu().numpy(), ref8)
    # the OTHER kernel for this accel alone (pinned: mnv.accel_set_fused_kernel overrides the harness default of the fixture): the same frame
    mnv.accel_set_fused_kernel(tree.accel, 1 if fused_kernel_choice(mnv) == 2 else 2)
    out.fill_(float("nan"))
    mnv.render_guided_fused(tree.accel, cam, opt, mlp, grid, rgba=out)
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(ref))
    mnv.accel_set_fused_kernel(tree.accel, -1)


@pytest.mark.parametrize("case,need_viewdir,max_g", [("sh9_d7_aniso", False, 64), ("sh4_d6", True, 8)])
def test_fused_frame_with_the_depth_image_of_the_live_call(mnv, torch_gpu, case, need_viewdir, max_g):
    """The reference's render loop calls get_samples_from_voxels and render_nerf_results with offscreen == false: every ray stops at the depth
    attachment's t_max (renderer_kernel.cu:354-357); the image under the volume enters render_nerf_results with weight 0.  The fused
    frame with the same limits (mnv_render_guided_fused_track_ex) equals the four-step path with them, bit for bit, and differs from the
    offscreen frame; limits of 1e9 everywhere are the offscreen frame."""
    import guided_cases
    torch = torch_gpu
    spec = cases.CASES[case]
    tree = cases.make_tree(mnv, spec["tree"])
    v = tree.host_view()
    tree.move_to_device()
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.basis_minmax[0], opt.basis_minmax[1] = 0, max(v.basis_dim - 1, 0)
    opt.max_guided_samples, opt.need_viewdir, opt.appearance_embedding = max_g, need_viewdir, -1
    desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, dir_octaves=2, need_viewdir=need_viewdir, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=21))
    grid = make_grid(mnv)
    dim = 4 + (3 if need_viewdir else 0)
    tmax = torch.from_numpy(guided_cases.onscreen_tmax(cam)).cuda()
    ref, ref8, total = four_step_frame(mnv, torch, tree, cam, opt, mlp, grid, max_g, dim, tmax_px=tmax)
    off, _, total_off = four_step_frame(mnv, torch, tree, cam, opt, mlp, grid, max_g, dim)
    assert 0 < total < total_off and int((cases.bits(ref) != cases.bits(off)).any(axis=-1).sum()) > 1000
    out = torch.full((cam.height, cam.width, 4), float("nan"), dtype=torch.float32, device="cuda")
    out8 = torch.zeros((cam.height, cam.width, 4), dtype=torch.uint8, device="cuda")
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    mnv.render_guided_fused(tree.accel, cam, opt, mlp, grid, rgba=out, rgba8=out8, sample_counter=counter, tmax_px=tmax)
    torch.cuda.synchronize()
    assert int(counter.item()) == total
    assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(ref)) and np.array_equal(out8.cpu().numpy(), ref8)
    out.fill_(float("nan"))
    mnv.render_guided_fused(tree.accel, cam, opt, mlp, grid, rgba=out, tmax_px=torch.full((cam.height, cam.width), 1e9, dtype=torch.float32, device="cuda"))
    torch.cuda.synchronize()
    assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(off))


def fused_kernel_choice(mnv):
    """The process-wide choice the module's fixture made for the running test."""
    return _FUSED_CHOICE[0]


_FUSED_CHOICE = [0]


def test_fused_frame_at_cfg2_size(mnv, torch_gpu):
    """The 1.5 M-chunk depth-10 SH9 tree at 1920x1080: about 9 M network evaluations inside the march; ragged tile (a 1000x600
    window) as well."""
    torch = torch_gpu
    tree = cases.make_tree(mnv, cases.CFG2_TREE)
    v = tree.host_view()
    tree.move_to_device()
    opt = mnv.RenderOptions.cli_defaults()
    opt.basis_minmax[1] = 8
    opt.max_guided_samples = 64
    desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, dir_octaves=2, need_viewdir=False, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=21))
    grid = make_grid(mnv)
    for (w, h) in ((1920, 1080), (1000, 600)):
        cam = cases.cfg2_camera(mnv, 2, w, h, 1600.0 * w / 1920)
        ref, ref8, total = four_step_frame(mnv, torch, tree, cam, opt, mlp, grid, 64, 4)
        out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
        counter = torch.zeros(1, dtype=torch.int64, device="cuda")
        mnv.render_guided_fused(tree.accel, cam, opt, mlp, grid, rgba=out, sample_counter=counter)
        torch.cuda.synchronize()
        assert int(counter.item()) == total and (w < 1920 or total > 5_000_000)
        assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(ref))
    assert mnv.accel_fused_faults(tree.accel) == 0


def oracle_frames(mnv, orc, torch, tree, cam, opt, mlp, desc, params, grid, max_g, dim):
    """The guided-sampling frame of cuda_renderer.cpp:107-139 built ON THE CPU from the oracle's pieces: orc.get_samples
    (get_samples_from_voxels, rt_core.cuh:418-576), the compaction of :116-121 (cumsum + mask), the network, orc.render_nerf_results
    (rt_core.cuh:334-416).  Returns (frame with the oracle's own network, frame with the HIP network kernel's values in between, total):
    the first is all-CPU and differs from any matrix-core evaluation by the order of the fp32 sums inside a network layer (which can
    flip a binary16 rounding of a hidden activation, tests/test_mlp_gpu.py); the second isolates everything BUT that order -- march,
    sample order, delta z, cluster choice, composite -- and must equal the fused kernel's frame bit for bit."""
    v = tree.host_view()
    ot = orc.tree_from_view(v)
    s = orc.get_samples(ot, cam.c, opt, grid, dim)
    flat = s["samples"].reshape(-1, dim)
    mask = flat[:, 0] >= 0
    valid, valid_clusters = np.ascontiguousarray(flat[mask]), np.ascontiguousarray(s["cluster_indices"].reshape(-1)[mask])
    offsets = np.cumsum(s["num_samples"].astype(np.int64))
    total = int(valid.shape[0])
    z = np.ascontiguousarray(valid[:, 0])
    cpu_values = orc.mlp_forward(desc, params, valid_clusters, valid[:, 1:])
    all_cpu = orc.render_nerf_results(ot, cam.c, opt, cpu_values, z, offsets)["rgba"]
    d_values = torch.zeros((max(total, 1), v.data_dim + 1), dtype=torch.float32, device="cuda")
    if total:
        mlp.query(torch.from_numpy(valid_clusters).cuda(), torch.from_numpy(np.ascontiguousarray(valid[:, 1:])).cuda(), d_values, n=total)
    torch.cuda.synchronize()
    hybrid = orc.render_nerf_results(ot, cam.c, opt, d_values.cpu().numpy()[:max(total, 1)], z, offsets)["rgba"]
    return all_cpu, hybrid, total


# Measured on an MI355X (round 4), all-CPU frame against the fused kernel's -- max |d| / share of pixels above the north star's 1e-4:
# rgba_d5 2.4e-3 / 0.0091, sh9_d7_aniso 3.8e-4 / 0.0001, sh4_d6 1.7e-4 / 0.0003, cfg2 at 480x270 8.2e-3 / 0.0014 (587 k samples).  Every one
# of those pixels is a sample where the order of the fp32 sums inside v_mfma_f32_16x16x32_f16 flipped the binary16 rounding of a hidden
# activation (tests/test_mlp_gpu.py bounds that per output); the hybrid frame -- the oracle's march, compaction and composite around
# the HIP network kernel's values -- is bit-identical in all four cases.
ORACLE_FRAME_TOL = 2e-2


@pytest.mark.parametrize("case,need_viewdir,n_emb,max_g,size", [
    ("rgba_d5", False, 0, 16, None),
    ("sh9_d7_aniso", False, 0, 32, None),
    ("sh4_d6", True, 3, 8, None),
    ("cfg2", False, 0, 32, (480, 270)),
])
def test_fused_frame_against_the_oracle(mnv, orc, torch_gpu, case, need_viewdir, n_emb, max_g, size):
    """mnv_render_guided_fused against the ORACLE's frame directly (not against another HIP path): the CPU chain orc.get_samples ->
    orc.mlp_forward -> orc.render_nerf_results, which is what cuda_renderer.cpp:107-139 computes.  Bit-equality where the definition
    allows it (everything but the summation order inside a network layer: the hybrid frame), the stated tolerance for the all-CPU frame."""
    torch = torch_gpu
    if case == "cfg2":
        tree = cases.make_tree(mnv, cases.CFG2_TREE)
        cam = cases.cfg2_camera(mnv, 3, size[0], size[1], 1600.0 * size[0] / 1920)
        opt = mnv.RenderOptions.cli_defaults()
    else:
        spec = cases.CASES[case]
        tree = cases.make_tree(mnv, spec["tree"])
        cam = cases.make_camera(mnv, spec["camera"])
        opt = cases.make_options(mnv, spec["options"])
    v = tree.host_view()
    tree.move_to_device()
    opt.basis_minmax[0], opt.basis_minmax[1] = 0, max(v.basis_dim - 1, 0)
    opt.max_guided_samples = max_g
    opt.need_viewdir = need_viewdir
    opt.appearance_embedding = 1 if n_emb else -1
    desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, dir_octaves=2, need_viewdir=need_viewdir, n_embeddings=n_emb, embedding_dim=8 if n_emb else 0,
                        hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    params = mlp_cases.make_params(mnv, desc, seed=21)
    mlp = mnv.Mlp(desc, params)
    grid = make_grid(mnv)
    dim = 4 + (3 if need_viewdir else 0) + (1 if n_emb else 0)
    all_cpu, hybrid, total = oracle_frames(mnv, orc, torch, tree, cam, opt, mlp, desc, params, grid, max_g, dim)
    assert total > 0
    out = torch.full((cam.height, cam.width, 4), float("nan"), dtype=torch.float32, device="cuda")
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    mnv.render_guided_fused(tree.accel, cam, opt, mlp, grid, rgba=out, sample_counter=counter)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert int(counter.item()) == total                          # the oracle's march and the fused march emit the same samples
    assert np.array_equal(cases.bits(got), cases.bits(hybrid)), float(np.nanmax(np.abs(got - hybrid)))
    d = np.abs(got - all_cpu)
    above = float((d.max(axis=-1) > 1e-4).mean())
    print(f"fused vs all-CPU oracle frame [{case}]: max |d| {d.max():.3e}, pixels above 1e-4: {above:.5f}, samples {total}")
    assert np.isfinite(got).all() and d.max() < ORACLE_FRAME_TOL and above < 0.02, (float(d.max()), above)
    assert float(np.abs(all_cpu[..., :3] - opt.background_brightness).max()) > 0.05  # the frame is not empty


def test_many_frames_in_a_row_stay_bit_identical(mnv, torch_gpu, fused_kernel):
    """The same frames over and over (8 poses x 80 launches at 1920x1080 for the producer / consumer kernel -- 640 frames, the length
    that first showed the defect of LAB_NOTEBOOK.md, "the rare wrong denominator"; 8 x 12 for the one-role kernel --, about 9 M samples
    each): which samples share a window of the
    network, which weight slot they find and when their owners composite them changes from launch to launch, the picture must not.
    (tools/fused_stress.py is the long form of this; LAB_NOTEBOOK.md, "the rare wrong denominator", is why it exists.)"""
    torch = torch_gpu
    tree = cases.make_tree(mnv, cases.CFG2_TREE)
    v = tree.host_view()
    tree.move_to_device()
    opt = mnv.RenderOptions.cli_defaults()
    opt.basis_minmax[1] = 8
    opt.max_guided_samples = 32
    desc = mnv.mlp_desc(n_clusters=8, pos_octaves=4, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=4))
    grid = make_grid(mnv)
    w, h = 1920, 1080
    out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    bad = []
    for pose in range(8):
        cam = cases.cfg2_camera(mnv, pose, w, h, 1600.0)
        ref, _, total = four_step_frame(mnv, torch, tree, cam, opt, mlp, grid, 32, 4)
        ref_bits = torch.from_numpy(cases.bits(ref).view(np.int32)).cuda()
        for rep in range(80 if fused_kernel == 2 else 12):
            out.fill_(float("nan"))
            mnv.render_guided_fused(tree.accel, cam, opt, mlp, grid, rgba=out)
            n_bad = int((out.view(torch.int32) != ref_bits).any(dim=-1).sum().item())
            if n_bad:
                bad.append((pose, rep, n_bad))
    assert not bad, f"frames that differ from the four-step path (pose, launch, pixels): {bad}"
    assert mnv.accel_fused_faults(tree.accel) == 0  # the always-on count of abandoned spin-waits (include/mnv.h: mnv_accel_fused_faults)


def test_fused_frame_rejects_what_it_does_not_cover(mnv, torch_gpu):
    torch = torch_gpu
    spec = cases.CASES["rgba_d5"]
    tree = cases.make_tree(mnv, spec["tree"])
    tree.move_to_device()
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    out = torch.empty((cam.height, cam.width, 4), dtype=torch.float32, device="cuda")
    wide = mnv.mlp_desc(n_clusters=2, pos_octaves=4, dir_octaves=2, need_viewdir=False, hidden_width=128, hidden_layers=2, out_dim=5)
    with pytest.raises(mnv.MnvError):
        mnv.render_guided_fused(tree.accel, cam, opt, mnv.Mlp(wide, mlp_cases.make_params(mnv, wide, seed=1)), make_grid(mnv), rgba=out)
    ok = mnv.mlp_desc(n_clusters=2, pos_octaves=4, dir_octaves=2, need_viewdir=False, hidden_width=64, hidden_layers=2, out_dim=5)
    m = mnv.Mlp(ok, mlp_cases.make_params(mnv, ok, seed=1))
    opt.render_depth = True
    with pytest.raises(mnv.MnvError):
        mnv.render_guided_fused(tree.accel, cam, opt, m, make_grid(mnv), rgba=out)
    opt.render_depth = False
    bad = mnv.mlp_desc(n_clusters=2, pos_octaves=4, dir_octaves=2, need_viewdir=False, hidden_width=64, hidden_layers=2, out_dim=7)
    with pytest.raises(mnv.MnvError):
        mnv.render_guided_fused(tree.accel, cam, opt, mnv.Mlp(bad, mlp_cases.make_params(mnv, bad, seed=1)), make_grid(mnv), rgba=out)


def test_volume_renderer_uses_the_fused_kernel_and_the_four_step_path_agrees(mnv, torch_gpu):
    """VolumeRenderer::render with use_guided_sampling: the fused kernel when only the picture is wanted (stats.fused), the
    four-step path when switched off or when the frame also feeds refinement; same frame either way."""
    from test_renderer_refine_gpu import setup

    frames = {}
    for fused in (True, False):
        r, tree, desc, params, cam_spec = setup(mnv, "sh9_d7_aniso", 4000, use_guided_sampling=True, max_guided_samples=24)
        r.set_fused_guided(fused)
        st = r.render()
        assert st["fused"] == int(fused) and st["used_accel"] == 1 and st["guided_samples"] > 0
        frames[fused] = (r.download(), st["guided_samples"])
    assert frames[True][1] == frames[False][1]
    assert np.array_equal(cases.bits(frames[True][0]), cases.bits(frames[False][0]))
    # with splitting on as well (configs[4]) the fused kernel also writes the trackers: the split step that follows finds candidates
    logs = {}
    for fused in (True, False):
        r, tree, desc, params, cam_spec = setup(mnv, "sh9_d7_aniso", 4000, use_guided_sampling=True, use_splitting=True, max_guided_samples=24, max_depth=9)
        r.set_fused_guided(fused)
        st = r.render()
        assert st["fused"] == int(fused) and st["guided_samples"] > 0 and st["split_candidates"] > 0 and st["added"] > 0
        logs[fused] = (st["split_candidates"], st["added"], st["capacity"], r.download())
    assert logs[True][:3] == logs[False][:3] and np.array_equal(cases.bits(logs[True][3]), cases.bits(logs[False][3]))


@pytest.mark.parametrize("case,max_g", [("sh4_d6", 16), ("sh9_d7_aniso", 4), ("rgba_d5", 128), ("shell_d7_sh9", 32)])
def test_fused_frame_with_trackers_and_visit_marks(mnv, torch_gpu, case, max_g):
    """BASELINE.json configs[4] has refinement AND guided sampling on: mnv_render_guided_fused_track writes, besides the picture,
    the tracker rows and visit marks that get_samples_from_voxels produces (rt_core.cuh:475-507,561-574,132-134) -- equal, element
    for element, to the sample march on the accel (itself pinned to the oracle and the reference's device code); the picture equals
    the four-step path's.  max_g = 4: rays keep marching after their quota, for the trackers' sake."""
    torch = torch_gpu
    spec = cases.CASES[case]
    tree = cases.make_tree(mnv, spec["tree"])
    v = tree.host_view()
    tree.move_to_device(need_parent=True, need_sample_counts=True)
    dv = tree.device_view()
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.basis_minmax[0], opt.basis_minmax[1] = 0, max(v.basis_dim - 1, 0)
    opt.max_guided_samples, opt.max_depth, opt.max_sample_count = max_g, 5, 9
    sc = np.full((v.capacity, 8), 8, np.int16)
    sc[::3] = 12
    sc_dev = torch.from_numpy(sc).cuda()
    desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, dir_octaves=2, need_viewdir=False, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=21))
    grid = make_grid(mnv)
    h, w = cam.height, cam.width
    n_px = h * w
    # the four-step path's first step with trackers and marks
    num = torch.zeros(n_px, dtype=torch.int16, device="cuda")
    guided = torch.zeros((n_px, max_g, 4), dtype=torch.float32, device="cuda")
    clusters = torch.zeros((n_px, max_g), dtype=torch.int16, device="cuda")
    split0 = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
    sample0 = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
    visited0 = torch.zeros(v.capacity, dtype=torch.int32, device="cuda")
    mnv.get_samples_from_voxels_accel_visit(tree.accel, cam, opt, visited0, dv.parent, num, guided, clusters, grid, split_track=split0,
                                            sample_track=sample0, sample_counts=sc_dev)
    ref, ref8, total = four_step_frame(mnv, torch, tree, cam, opt, mlp, grid, max_g, 4)
    # one kernel
    out = torch.full((h, w, 4), float("nan"), dtype=torch.float32, device="cuda")
    split = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
    sample = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
    visited = torch.zeros(v.capacity, dtype=torch.int32, device="cuda")
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    mnv.render_guided_fused(tree.accel, cam, opt, mlp, grid, rgba=out, sample_counter=counter, split_track=split, sample_track=sample,
                            sample_counts=sc_dev, visited=visited, parent=dv.parent)
    torch.cuda.synchronize()
    assert int(counter[0].item()) == total
    assert np.array_equal(cases.bits(out.cpu().numpy()), cases.bits(ref))
    assert torch.equal(split, split0) and torch.equal(sample, sample0)
    assert torch.equal(visited, visited0) and int(visited.sum()) > 1
    assert (split[..., 1] >= 0).any()


def test_fused_frame_of_a_sub_rectangle(mnv, torch_gpu):
    """`tile` = a window of the image (ragged 8x8 tiles at its right and bottom edges): the fused kernel writes exactly the crop of the
    full frame, and nothing outside its [h][w] buffer."""
    torch = torch_gpu
    spec = cases.CASES["sh9_d7_aniso"]
    tree = cases.make_tree(mnv, spec["tree"])
    v = tree.host_view()
    tree.move_to_device()
    cam = cases.make_camera(mnv, spec["camera"])
    opt = cases.make_options(mnv, spec["options"])
    opt.basis_minmax[0], opt.basis_minmax[1] = 0, 8
    opt.max_guided_samples = 16
    desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, dir_octaves=2, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=21))
    grid = make_grid(mnv)
    full = torch.empty((cam.height, cam.width, 4), dtype=torch.float32, device="cuda")
    mnv.render_guided_fused(tree.accel, cam, opt, mlp, grid, rgba=full)
    x0, y0, w, h = 37, 21, 101, 67
    guard = torch.full((h + 2, w, 4), float("nan"), dtype=torch.float32, device="cuda")
    win = guard[1:h + 1]
    mnv.render_guided_fused(tree.accel, cam, opt, mlp, grid, tile=(x0, y0, w, h), rgba=win)
    torch.cuda.synchronize()
    assert torch.equal(win.view(torch.int32), full[y0:y0 + h, x0:x0 + w].contiguous().view(torch.int32))
    assert torch.isnan(guard[0]).all() and torch.isnan(guard[h + 1]).all()


def test_fused_frames_in_flight_on_several_streams(mnv, torch_gpu):
    """mnv_render_guided_fused is re-entrant across HIP streams like the plain march (per-launch queue heads and camera blocks live in the
    accel's slot ring): twelve frames of the cfg2 tree, three in flight at a time, each equal to the frame rendered alone, the shared sample
    counter equal to the sum, no fault."""
    torch = torch_gpu
    tree = cases.make_tree(mnv, cases.CFG2_TREE)
    v = tree.host_view()
    tree.move_to_device()
    opt = mnv.RenderOptions.cli_defaults()
    opt.basis_minmax[1] = 8
    opt.max_guided_samples = 32
    desc = mnv.mlp_desc(n_clusters=6, pos_octaves=4, dir_octaves=2, need_viewdir=False, hidden_width=64, hidden_layers=2, out_dim=v.data_dim + 1)
    mlp = mnv.Mlp(desc, mlp_cases.make_params(mnv, desc, seed=21))
    grid = make_grid(mnv)
    w, h, n = 960, 540, 12
    cams = [cases.cfg2_camera(mnv, p, w, h, 800.0) for p in range(n)]
    alone = torch.empty((n, h, w, 4), dtype=torch.float32, device="cuda")
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    for i, cam in enumerate(cams):
        mnv.render_guided_fused(tree.accel, cam, opt, mlp, grid, rgba=alone[i], sample_counter=counter)
    torch.cuda.synchronize()
    total = int(counter.item())
    counter.zero_()
    streams = [torch.cuda.Stream() for _ in range(3)]
    torch.cuda.synchronize()
    flight = torch.full((n, h, w, 4), float("nan"), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for lap in range(2):
        for i, cam in enumerate(cams):
            mnv.render_guided_fused(tree.accel, cam, opt, mlp, grid, rgba=flight[i], sample_counter=counter, stream=streams[i % 3].cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(flight.view(torch.int32), alone.view(torch.int32))
    assert int(counter.item()) == 2 * total and total > 1_000_000
    assert mnv.accel_fused_faults(tree.accel) == 0


def test_volume_renderer_guided_frames_in_flight(mnv, torch_gpu, fused_kernel):
    """VolumeRenderer::guided_in_flight: guided-sampling frames that change nothing rotate over the frame slots; every frame and its sample
    count equal what the one-at-a-time renderer gives for the same camera, and a frame that cannot overlap (splitting switched on) falls back
    to slot 0 with its count in the stats."""
    from test_renderer_refine_gpu import setup

    poses = [((-3.0 + 0.1 * k, 2.0, 5.0 - 0.1 * k), (-0.45, 0.3, 0.75)) for k in range(7)]
    serial = []
    r, tree, desc, params, cam_spec = setup(mnv, "sh9_d7_aniso", 400000, use_guided_sampling=True, max_guided_samples=24)
    for center, back in poses:
        r.set_camera(center, back, fx=cam_spec["fx"])
        st = r.render()
        assert st["fused"] == 1 and st["guided_samples"] > 0 and r.last_slot() == 0
        serial.append((r.download(), st["guided_samples"]))
    r, tree, desc, params, cam_spec = setup(mnv, "sh9_d7_aniso", 400000, use_guided_sampling=True, max_guided_samples=24)
    r.set_guided_in_flight(True)
    r.set_frames_in_flight(3)
    slots = []
    for center, back in poses:
        r.set_camera(center, back, fx=cam_spec["fx"])
        st = r.render()
        assert st["fused"] == 1 and st["used_accel"] == 1 and st["guided_samples"] == -1
        slots.append(r.last_slot())
        if len(slots) >= 3:  # frame k is read once frames k + 1 and k + 2 have been issued
            k = len(slots) - 3
            assert r.slot_guided_samples(slots[k]) == serial[k][1]
            assert np.array_equal(cases.bits(r.download_slot(slots[k])), cases.bits(serial[k][0])), k
    assert slots[:4] == [1, 2, 0, 1]
    for k in (len(poses) - 2, len(poses) - 1):
        assert r.slot_guided_samples(slots[k]) == serial[k][1]
        assert np.array_equal(cases.bits(r.download_slot(slots[k])), cases.bits(serial[k][0])), k
    # a frame that edits the tree does not overlap
    r.options.use_splitting, r.options.max_depth = True, 9
    st = r.render()
    assert r.last_slot() == 0 and st["guided_samples"] > 0 and st["split_candidates"] > 0
    assert r.slot_guided_samples(0) == st["guided_samples"]

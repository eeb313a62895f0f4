"""The four remaining launchers of include/cuda/renderer_kernel.hpp -- render_nerf_results (:12-21), add_children_and_generate_samples
(:54-63), generate_samples (:65-73), adjust_parents_and_children (:75-79) -- with their ORIGINAL parameter lists, served by libmnv.so
through include/mnv_reference_binding.hpp compiled INSIDE a build of the reference (oracle/Makefile.ref -> oracle/_ref): the reference's own
loader, N3Tree (libtorch tensors on the device) and Camera (glm) are the arguments, exactly as src/renderer/cuda_renderer.cpp:138,255,
306,358 pass them.  Expected arrays: what the reference's own kernels wrote on gfx950 (tests/golden/ref_refine_kernels.npz,
ref_guided_nerf_results_*.npz) and -- live -- the reference's kernels run in the same process on the same inputs.
(render_voxels and get_samples_from_voxels, the other two of the six: test_parity_gpu.py::test_reference_binding_is_a_drop_in,
test_onscreen_gpu.py.)"""
import os

import numpy as np
import pytest

import cases
import guided_cases
import refine_kernel_cases as rk
from test_parity_gpu import require_live_reference

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("variant", list(rk.VARIANTS))
def test_add_children_and_generate_samples_through_the_nine_parameter_binding(mnv, torch_gpu, tmp_path, variant):
    mnv_ref = require_live_reference()
    z = np.load(os.path.join(GOLD, "ref_refine_kernels.npz"))
    g = rk.grid(mnv)
    tree, opt, dim, parent_nodes, visited, samples = rk.add_children_inputs(mnv, variant)
    path = str(tmp_path / "t.npz")
    tree.save_npz(path)
    cap = tree.capacity
    got = mnv_ref.add_children_npz(path, opt, cap + rk.N_NEW, parent_nodes, samples, visited, g, dropin=True)
    live = mnv_ref.add_children_npz(path, opt, cap + rk.N_NEW, parent_nodes, samples, visited, g, dropin=False)
    pre = f"add_children/{variant}/"
    for k in ("child", "parent", "visited", "clusters"):
        # rows >= capacity + n of child / parent are uninitialised device memory in both (move_to_device allocates, n3tree.cpp:208-233)
        n = (cap + rk.N_NEW) if k in ("child", "parent", "visited") else None
        assert np.array_equal(got[k][:n], z[pre + k][:n]), k
        assert np.array_equal(got[k][:n], live[k][:n]), k
    assert np.array_equal(cases.bits(got["samples"]), cases.bits(z[pre + "samples"]))
    assert np.array_equal(cases.bits(got["samples"]), cases.bits(live["samples"]))


@pytest.mark.parametrize("variant", list(rk.VARIANTS))
def test_generate_samples_through_the_eight_parameter_binding(mnv, torch_gpu, tmp_path, variant):
    mnv_ref = require_live_reference()
    z = np.load(os.path.join(GOLD, "ref_refine_kernels.npz"))
    g = rk.grid(mnv)
    tree, opt, dim, nodes, samples = rk.generate_samples_inputs(mnv, variant)
    path = str(tmp_path / "t.npz")
    tree.save_npz(path)
    got = mnv_ref.generate_samples_npz(path, opt, nodes, samples, g, dropin=True)
    live = mnv_ref.generate_samples_npz(path, opt, nodes, samples, g, dropin=False)
    pre = f"generate_samples/{variant}/"
    assert np.array_equal(cases.bits(got["samples"]), cases.bits(z[pre + "samples"])) and np.array_equal(got["clusters"], z[pre + "clusters"])
    assert np.array_equal(cases.bits(got["samples"]), cases.bits(live["samples"])) and np.array_equal(got["clusters"], live["clusters"])


def test_adjust_parents_and_children_through_the_four_parameter_binding(mnv, orc, torch_gpu, tmp_path):
    mnv_ref = require_live_reference()
    z = np.load(os.path.join(GOLD, "ref_refine_kernels.npz"))
    tree, to_delete, shifts = rk.adjust_inputs(mnv, orc)
    path = str(tmp_path / "t.npz")
    tree.save_npz(path)
    got = mnv_ref.adjust_parents_npz(path, tree.capacity, 1, to_delete, shifts, dropin=True)
    live = mnv_ref.adjust_parents_npz(path, tree.capacity, 1, to_delete, shifts, dropin=False)
    keep = to_delete == 0   # rows of deleted chunks are dropped by the compaction that follows (cuda_renderer.cpp:360-376)
    assert int(to_delete.sum()) > 0 and int(keep.sum()) > 1
    assert np.array_equal(got["child"], z["adjust_parents/child"]) and np.array_equal(got["child"], live["child"])
    assert np.array_equal(got["parent"][keep], z["adjust_parents/parent"][keep]) and np.array_equal(got["parent"][keep], live["parent"][keep])


@pytest.mark.parametrize("case", ["sh4_d6", "rgba_d5"])
def test_render_nerf_results_through_the_nine_parameter_binding(mnv, orc, torch_gpu, tmp_path, case):
    """The launcher writes RGBA8 through the image surface (renderer_kernel.cu:237): the binding's image equals the C ABI's RGBA8 output and the
    oracle's byte for byte, equals the u8 pack of the reference's own float frame (golden) up to the rare pixel whose float sits within the
    oracle / reference last-bit difference of a u8 boundary, and does not depend on what the image held before or on `offscreen`
    (alpha starts at 1, renderer_kernel.cu:316)."""
    torch = torch_gpu
    mnv_ref = require_live_reference()
    g = np.load(os.path.join(GOLD, f"ref_guided_nerf_results_{case}.npz"))
    tree, cam, opt, values, zv, offsets = guided_cases.nerf_results_setup(mnv, case)
    path = str(tmp_path / "t.npz")
    tree.save_npz(path)
    got = mnv_ref.render_nerf_results_dropin_npz(path, cam.c, opt, values, zv, offsets)
    rng = np.random.default_rng(3)
    under = rng.integers(0, 256, got.shape, dtype=np.uint8)
    got2 = mnv_ref.render_nerf_results_dropin_npz(path, cam.c, opt, values, zv, offsets, image=under, offscreen=False)
    assert np.array_equal(got, got2)
    want = orc.render_nerf_results(orc.tree_from_view(tree.host_view()), cam.c, opt, values, zv, offsets, want_rgba8=True)
    assert np.array_equal(got, want["rgba8"])
    out8 = torch.zeros((cam.height, cam.width, 4), dtype=torch.uint8, device="cuda")
    mnv.render_nerf_results(tree.host_view(), cam, opt, torch.from_numpy(values).cuda(), torch.from_numpy(zv).cuda(), torch.from_numpy(offsets).cuda(),
                            rgba8=out8)
    torch.cuda.synchronize()
    assert np.array_equal(got, out8.cpu().numpy())
    ref8 = np.clip(g["rgba"][..., :3] * 255.0, 0, 255).astype(np.uint8)   # truncation, as uint8_t(out * 255) does for values in range
    diff = (got[..., :3].astype(np.int32) - ref8.astype(np.int32))
    assert np.abs(diff).max() <= 1 and int((diff != 0).sum()) <= 4, (int(np.abs(diff).max()), int((diff != 0).sum()))
    assert (got[..., 3] == 255).all()

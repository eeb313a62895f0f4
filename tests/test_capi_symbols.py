"""The C-ABI library loads without a GPU and exports every symbol include/mnv.h declares; the ctypes
binding covers all of them; POD structs have the layout the header promises."""
import ctypes as C
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "mnv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mnv_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(mnv):
    names = header_functions()
    assert len(names) >= 25
    out = subprocess.run(["nm", "-D", "--defined-only", mnv.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (mnv_[a-z0-9_]+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, missing
    assert set(names) == set(mnv._SIGNATURES), set(names) ^ set(mnv._SIGNATURES)


# the drop-in core of include/mnv.h: one entry point per function / method of the reference's interface for this path
CORE = """mnv_render_voxels mnv_render_voxels_ex mnv_get_samples_from_voxels mnv_get_samples_from_voxels_ex mnv_render_nerf_results
mnv_add_children_and_generate_samples mnv_generate_samples mnv_adjust_parents_and_children
mnv_renderer_create mnv_renderer_destroy mnv_renderer_set mnv_renderer_load_model mnv_renderer_resize mnv_renderer_options mnv_renderer_set_camera
mnv_renderer_render mnv_renderer_download
mnv_n3tree_open mnv_n3tree_free mnv_n3tree_move_to_device mnv_n3tree_host_view mnv_n3tree_device_view
mnv_data_format_parse mnv_data_format_to_string mnv_camera_init mnv_camera_set_pose mnv_camera_drag
mnv_default_render_options mnv_cli_render_options mnv_last_error""".split()


def test_the_drop_in_core_is_thirty_symbols_and_no_call_depends_on_process_wide_switches(mnv):
    """The header's map (reference function -> core entry point) names 30 symbols, all exported; the only process-wide setter left in the ABI
    is the opt-in tree cache (colour math, fused kernel, diagnostics are per accel; launch timing is the caller's; the lookup-table threshold
    of mnv_render_voxels exists in the test-hook build alone)."""
    names = header_functions()
    assert len(CORE) == 30 and all(n in names for n in CORE)
    header = open(os.path.join(ROOT, "include", "mnv.h")).read()
    for n in CORE:
        assert n in header or n.replace("mnv_renderer", "") in header   # (the map abbreviates the renderer's methods)
    setters = [n for n in names if re.match(r"mnv_set_", n)]
    assert setters == ["mnv_set_tree_cache"], setters
    out = subprocess.run(["nm", "-D", "--defined-only", mnv.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "mnv_hook_" not in out
    hooks = subprocess.run(["nm", "-D", "--defined-only", os.path.join(os.path.dirname(mnv.LIB_PATH), "testhooks", "libmnv.so")], capture_output=True, text=True).stdout
    assert "mnv_hook_set_ref_table_min_rays" in hooks or os.environ.get("MNV_LIB_PATH")


def test_no_torch_or_oracle_in_the_abi(mnv):
    out = subprocess.run(["ldd", mnv.LIB_PATH], capture_output=True, text=True).stdout
    assert "torch" not in out and "oracle" not in out and "libamdhip64" in out


def test_struct_layouts_and_defaults(mnv):
    assert C.sizeof(mnv.RenderOptions) == 104          # reference include/render_options.hpp: 104 B by value
    assert C.sizeof(mnv.CameraStruct) == 72 and C.sizeof(mnv.Rect) == 16 and C.sizeof(mnv.Partition) == 20
    o = mnv.RenderOptions.defaults()                    # struct defaults, render_options.hpp:9-56
    assert (o.step_size, o.sigma_thresh, o.stop_thresh, o.background_brightness) == (C.c_float(1e-4).value, C.c_float(1e-2).value, C.c_float(1e-2).value, 1.0)
    assert list(o.render_bbox) == [0, 0, 0, 1, 1, 1] and list(o.basis_minmax) == [0, 24]
    assert (o.max_depth, o.samples_per_corner, o.split_batch_size, o.nerf_batch_size, o.max_sample_count) == (16, 8, 4192, 1024, 256)
    assert (o.appearance_embedding, o.max_guided_samples, o.grid_max_depth) == (-1, 128, 4)
    c = mnv.RenderOptions.cli_defaults()                # src/opts.cpp:17-32
    assert c.background_brightness == 0.0 and c.split_batch_size == 4096 and c.nerf_batch_size == 4096
    assert mnv.lib().mnv_version() == 1


def test_device_entry_points_fail_loudly_without_a_gpu(mnv):
    """No CPU fallback: on a box without a HIP device the product path reports an error."""
    import pytest
    if mnv.device_count() > 0:
        pytest.skip("a GPU is present")
    t = mnv.N3Tree.synth_random(depth=2, basis_dim=1)
    with pytest.raises(mnv.MnvError) as e:
        t.move_to_device()
    assert e.value.code == mnv.MNV_E_NO_DEVICE


def test_comm_entry_points_validate_without_a_gpu(mnv):
    """mnv_comm_* / mnv_gather_tiles reject bad arguments before they touch RCCL or a device."""
    lib = mnv.lib()
    assert lib.mnv_gather_tiles(None, None, None, 16, 0, None) == mnv.MNV_E_INVALID
    assert lib.mnv_comm_get_unique_id(None) == mnv.MNV_E_INVALID
    h = C.c_void_p()
    assert lib.mnv_comm_init_rank(None, 2, 0, C.byref(h)) == mnv.MNV_E_INVALID
    buf = C.create_string_buffer(128)
    assert lib.mnv_comm_init_rank(buf, 2, 2, C.byref(h)) == mnv.MNV_E_INVALID      # rank outside the world
    assert lib.mnv_comm_rank(None) == -1 and lib.mnv_comm_world(None) == 0
    lib.mnv_comm_destroy(None)


def test_the_shipped_binaries_carry_no_test_hooks(mnv):
    """MNV_RCCL_LIBRARY (a stand-in for RCCL's transport) and MNV_RANKS_SHARE_GPU (every rank on one device) exist only in the
    -DMNV_TEST_HOOKS build under testhooks/ that the rehearsal tests load (tests/hooks.py): the shipped library cannot be pointed at
    an arbitrary shared object through the environment."""
    import hooks

    def mentions(path, name):
        return name.encode() in open(path, "rb").read()

    exe = os.path.join(ROOT, "mega-nerf-viewer_amd", "mnv_render")
    shipped = mnv.LIB_PATH if not os.environ.get("MNV_LIB_PATH") else os.path.join(ROOT, "mega-nerf-viewer_amd", "libmnv.so")
    assert not mentions(shipped, "MNV_RCCL_LIBRARY")
    assert not mentions(exe, "MNV_RANKS_SHARE_GPU")
    assert mentions(hooks.HOOKS_LIB, "MNV_RCCL_LIBRARY") and mentions(hooks.HOOKS_EXE, "MNV_RANKS_SHARE_GPU")
    # ... nor be steered by a stray measurement knob in a user's shell (MNV_STATS, MNV_BLOCKS_PER_CU, MNV_ABLATE ...): the shipped
    # library and binary name NO environment variable of their own; every knob of csrc/mnv_knobs.h lives in the test-hook build
    import re

    def env_names(path):
        return sorted(set(m.decode() for m in re.findall(rb"MNV_[A-Z][A-Z0-9_]{2,}", open(path, "rb").read())))

    allowed = {"MNV_LIB_PATH", "MNV_MAX_BATCH"}  # (the second is a macro named in an error message)
    assert set(env_names(shipped)) <= allowed, env_names(shipped)
    assert set(env_names(exe)) <= allowed, env_names(exe)
    knobs = re.findall(r"KNOB_([A-Z0-9_]+),", open(os.path.join(ROOT, "mega-nerf-viewer_amd", "csrc", "mnv_knobs.h")).read())
    assert len(knobs) >= 15 and all(("MNV_" + k) in env_names(hooks.HOOKS_LIB) for k in knobs)


def test_no_packed_fp32_instructions_in_the_code_objects(mnv):
    """gfx950: a VOP3P packed-FP32 add whose low half takes the high dword of a source (op_sel) sporadically read that operand as 0.0
    in lanes 48-63 beside co-resident MFMA wavefronts -- the cause of the "rare wrong denominator" of guided_fused2_kernel (DESIGN.md
    5.4, LAB_NOTEBOOK.md).  The build switches the instructions off (Makefile: NOPK); this holds the shipped code objects to it."""
    import glob
    import shutil
    import tempfile

    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        import pytest
        pytest.skip("no llvm-objdump on this machine")
    lib = os.path.join(ROOT, "mega-nerf-viewer_amd", "libmnv.so")
    with tempfile.TemporaryDirectory() as d:
        so = os.path.join(d, "libmnv.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", so], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = sorted(glob.glob(so + ".*gfx950*"))
        assert len(objs) >= 5, objs  # one code object per .hip translation unit
        n_insts, hits = 0, []
        for co in objs:
            dis = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
            n_insts += dis.count("\n")
            hits += re.findall(r"\bv_pk_(?:add|mul|fma|mov)_[fb]32\b[^\n]*", dis)
    assert n_insts > 100_000 and not hits, hits[:5]


def test_the_library_was_built_from_the_sources_in_this_tree(mnv):
    """mnv_source_sha (compiled in by the Makefile over its source lists) == the same hash taken over the tree by the Python side
    (shipped_source_sha): a source file the Makefile compiles but the Python mirror does not list (or the other way round) would make
    smoke() call every library stale -- as happened when csrc/mnv_knobs.cpp was added -- and an edited source without a rebuild fails here
    instead of passing tests with old code."""
    assert mnv.built_source_sha() == mnv.shipped_source_sha(), "libmnv.so is stale (run make) or the two source lists differ"


def test_live_reference_build_carries_all_six_launcher_bindings():
    """oracle/_ref (the reference's own device code built for gfx950, with include/mnv_reference_binding.hpp compiled in) exports the entry
    points through which the tests call the six original-signature launchers; skipped where the library was never built (no /root/reference)."""
    import pytest

    ref = os.path.join(ROOT, "oracle", "_ref", "libmnv_ref_gfx950.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/libmnv_ref_gfx950.so not built (needs /root/reference at build time)")
    out = subprocess.run(["nm", "-D", "--defined-only", ref], capture_output=True, text=True, check=True).stdout
    for name in ("ref_dropin_render_npz", "ref_dropin_onscreen_npz", "ref_get_samples_onscreen_npz", "ref_render_nerf_results_dropin_npz",
                 "ref_add_children_dropin_npz", "ref_generate_samples_dropin_npz", "ref_adjust_parents_dropin_npz"):
        assert f" T {name}" in out, name
    # ... and it resolves libmnv.so's entry points of those launchers dynamically (it carries no copy of the product)
    und = subprocess.run(["nm", "-D", "--undefined-only", ref], capture_output=True, text=True, check=True).stdout
    for name in ("mnv_render_voxels_ex", "mnv_get_samples_from_voxels_ex", "mnv_render_nerf_results", "mnv_add_children_and_generate_samples",
                 "mnv_generate_samples", "mnv_adjust_parents_and_children"):
        assert f" U {name}" in und, name

"""ctypes binding of libmnv.so (include/mnv.h) for the test-suite, bench.py and the
multi-GPU tile driver.

This is plumbing over the C ABI, not a second implementation: every render call goes
through ``mnv_render_voxels`` / ``mnv_render_voxels_accel`` into the hand-written HIP
kernels.  There is no CPU or PyTorch fallback -- if ``libmnv.so`` is missing the import of
the library raises, and a render call on a machine without a HIP device returns the
library's error (``MnvError``).

Reference surface mirrored here (cmusatyalab/mega-nerf-viewer):
  * ``N3Tree``           include/n3tree/n3tree.hpp:17-69
  * ``Camera``           include/camera.hpp:12-87 (pose model + intrinsics)
  * ``RenderOptions``    include/render_options.hpp:9-56
  * ``render_voxels``    include/cuda/renderer_kernel.hpp:23-34
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MNV_LIB_PATH") or os.path.join(_HERE, "libmnv.so")   # MNV_LIB_PATH: A/B builds of the library (tools/build_variant.sh)

MNV_OK = 0
MNV_E_INVALID = -1
MNV_E_UNSUPPORTED = -2
MNV_E_NO_DEVICE = -3
MNV_E_IO = -4
MNV_E_NO_RCCL = -5
MNV_E_RCCL = -6
MNV_E_FAULT = -7
FORMAT_RGBA = 0
FORMAT_SH = 1
BASIS_MAX = 25


class MnvError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mnv error {code}: {msg}")
        self.code = code


class TreeView(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("child", C.c_void_p),
        ("parent", C.c_void_p),
        ("sample_counts", C.c_void_p),
        ("offset", C.c_float * 3),
        ("scale", C.c_float * 3),
        ("N", C.c_int32),
        ("data_dim", C.c_int32),
        ("format", C.c_int32),
        ("basis_dim", C.c_int32),
        ("capacity", C.c_int32),
    ]


class CameraStruct(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("fx", C.c_float),
        ("fy", C.c_float),
        ("cx", C.c_float),
        ("cy", C.c_float),
        ("c2w", C.c_float * 12),
    ]


class RenderOptions(C.Structure):
    """viewer::RenderOptions, field for field (reference include/render_options.hpp:9-56)."""

    _fields_ = [
        ("step_size", C.c_float),
        ("sigma_thresh", C.c_float),
        ("stop_thresh", C.c_float),
        ("background_brightness", C.c_float),
        ("render_bbox", C.c_float * 6),
        ("basis_minmax", C.c_int32 * 2),
        ("rot_dirs", C.c_float * 3),
        ("show_grid", C.c_bool),
        ("grid_max_depth", C.c_int32),
        ("render_depth", C.c_bool),
        ("use_splitting", C.c_bool),
        ("use_guided_sampling", C.c_bool),
        ("max_depth", C.c_int32),
        ("samples_per_corner", C.c_int32),
        ("split_batch_size", C.c_int32),
        ("nerf_batch_size", C.c_int32),
        ("max_sample_count", C.c_int32),
        ("need_viewdir", C.c_bool),
        ("appearance_embedding", C.c_int32),
        ("max_guided_samples", C.c_int32),
    ]

    @classmethod
    def defaults(cls) -> "RenderOptions":
        o = cls()
        lib().mnv_default_render_options(C.byref(o))
        return o

    @classmethod
    def cli_defaults(cls) -> "RenderOptions":
        o = cls()
        lib().mnv_cli_render_options(C.byref(o))
        return o


class Rect(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("w", C.c_int32), ("h", C.c_int32)]


class FrameInputs(C.Structure):
    """mnv_frame_inputs: the per-pixel arrays of the reference's offscreen == false call shape (device pointers, either may be NULL)."""
    _fields_ = [("tmax_px", C.c_void_p), ("rgba8_init", C.c_void_p)]


class Partition(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("tile_w", C.c_int32), ("tile_h", C.c_int32), ("root_period", C.c_int32)]


class ClusterGrid(C.Structure):
    _fields_ = [("grid_dim", C.c_int32 * 2), ("min_position", C.c_float * 3), ("range", C.c_float * 3)]


class TreeEdit(C.Structure):
    _fields_ = [("child", C.c_void_p), ("parent", C.c_void_p), ("offset", C.c_float * 3), ("scale", C.c_float * 3),
                ("N", C.c_int32), ("capacity", C.c_int32)]


class MlpDesc(C.Structure):
    _fields_ = [("n_clusters", C.c_int32), ("pos_octaves", C.c_int32), ("dir_octaves", C.c_int32), ("need_viewdir", C.c_int32),
                ("n_embeddings", C.c_int32), ("embedding_dim", C.c_int32), ("hidden_width", C.c_int32), ("hidden_layers", C.c_int32),
                ("out_dim", C.c_int32), ("center", C.c_float * 3), ("inv_extent", C.c_float * 3)]


class RendererStats(C.Structure):
    _fields_ = [("track_visit", C.c_int32), ("used_accel", C.c_int32), ("full", C.c_int32), ("split_candidates", C.c_int32),
                ("added", C.c_int32), ("sample_candidates", C.c_int32), ("resampled", C.c_int32), ("pruned", C.c_int32),
                ("guided_samples", C.c_int64), ("capacity", C.c_int64), ("fused", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class SynthRandomParams(C.Structure):
    _fields_ = [
        ("depth", C.c_int32),
        ("format", C.c_int32),
        ("basis_dim", C.c_int32),
        ("refine_prob", C.c_float),
        ("empty_prob", C.c_float),
        ("sigma_max", C.c_float),
        ("coef_sd", C.c_float),
        ("offset", C.c_float * 3),
        ("scale", C.c_float * 3),
        ("seed", C.c_uint64),
    ]


class SynthTerrainParams(C.Structure):
    _fields_ = [("depth", C.c_int32), ("basis_dim", C.c_int32), ("bricks_y", C.c_int32), ("bricks_z", C.c_int32),
                ("noise_cells", C.c_int32), ("base", C.c_float), ("amplitude", C.c_float), ("thickness", C.c_float),
                ("sigma_lo", C.c_float), ("sigma_hi", C.c_float), ("offset", C.c_float * 3), ("scale", C.c_float * 3),
                ("seed", C.c_uint64)]


class SynthShellParams(C.Structure):
    _fields_ = [
        ("depth", C.c_int32),
        ("basis_dim", C.c_int32),
        ("radius", C.c_float),
        ("half_thickness", C.c_float),
        ("sigma_lo", C.c_float),
        ("sigma_hi", C.c_float),
        ("offset", C.c_float * 3),
        ("scale", C.c_float * 3),
        ("seed", C.c_uint64),
    ]


# every symbol include/mnv.h declares (tests/test_capi_symbols.py checks the export list)
_SIGNATURES = {
    "mnv_version": (C.c_int, []),
    "mnv_last_error": (C.c_char_p, []),
    "mnv_source_sha": (C.c_char_p, []),
    "mnv_device_count": (C.c_int, []),
    "mnv_default_render_options": (None, [C.POINTER(RenderOptions)]),
    "mnv_cli_render_options": (None, [C.POINTER(RenderOptions)]),
    "mnv_camera_init": (None, [C.POINTER(CameraStruct), C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float]),
    "mnv_camera_set_pose": (None, [C.POINTER(CameraStruct), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mnv_camera_drag": (None, [C.POINTER(CameraStruct), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float,
                               C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]),
    "mnv_render_voxels": (C.c_int, [C.POINTER(TreeView), C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mnv_render_voxels_ex": (C.c_int, [C.POINTER(TreeView), C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect, C.POINTER(FrameInputs),
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mnv_render_voxels_accel_ex": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect, C.POINTER(FrameInputs),
                                             C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_accel_create": (C.c_int, [C.POINTER(TreeView), C.c_void_p, C.POINTER(C.c_void_p)]),
    "mnv_accel_create_reserved": (C.c_int, [C.POINTER(TreeView), C.c_int64, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mnv_accel_refresh": (C.c_int, [C.c_void_p, C.POINTER(TreeView), C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "mnv_accel_rebuild": (C.c_int, [C.c_void_p, C.POINTER(TreeView), C.c_void_p]),
    "mnv_accel_destroy": (None, [C.c_void_p]),
    "mnv_accel_device_bytes": (C.c_size_t, [C.c_void_p]),
    "mnv_accel_grid2_level": (C.c_int32, [C.c_void_p]),
    "mnv_accel_brick_levels": (C.c_int32, [C.c_void_p]),
    "mnv_accel_lookup_coverage": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mnv_accel_set_cu_budget": (C.c_int, [C.c_void_p, C.c_int32]),
    "mnv_stream_create_reserved": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]),
    "mnv_stream_destroy": (C.c_int, [C.c_void_p]),
    "mnv_render_voxels_accel": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_partition_local_tiles": (C.c_int32, [Rect, Partition]),
    "mnv_render_voxels_accel_part": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect, Partition,
                                               C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_accel_set_colour_math": (C.c_int, [C.c_void_p, C.c_int]),
    "mnv_accel_set_fused_kernel": (C.c_int, [C.c_void_p, C.c_int]),
    "mnv_accel_set_fused_diag": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mnv_accel_fused_faults": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "mnv_set_tree_cache": (None, [C.c_int]),
    "mnv_tree_invalidate": (None, [C.c_void_p]),
    "mnv_assemble_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, Partition, C.c_int32, C.c_int32, C.c_void_p]),
    "mnv_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "mnv_comm_init_rank": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "mnv_comm_rank": (C.c_int32, [C.c_void_p]),
    "mnv_comm_world": (C.c_int32, [C.c_void_p]),
    "mnv_comm_rccl_version": (C.c_int32, []),
    "mnv_gather_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "mnv_comm_destroy": (None, [C.c_void_p]),
    "mnv_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mnv_merge_visit_marks": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mnv_render_voxels_accel_visit_part": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect, Partition, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_render_guided_fused_track_part": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect, Partition, C.c_void_p,
                                                     C.POINTER(ClusterGrid), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_render_voxels_accel_batch": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.c_int32, C.POINTER(RenderOptions), Rect,
                                                Partition, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_render_voxels_accel_track": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_get_samples_from_voxels": (C.c_int, [C.POINTER(TreeView), C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int32,
                                              C.c_void_p, C.POINTER(ClusterGrid), C.c_void_p]),
    "mnv_get_samples_from_voxels_ex": (C.c_int, [C.POINTER(TreeView), C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect, C.POINTER(FrameInputs),
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int32,
                                                 C.c_void_p, C.POINTER(ClusterGrid), C.c_void_p]),
    "mnv_get_samples_from_voxels_accel_visit_ex": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect, C.POINTER(FrameInputs),
                                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                             C.c_void_p, C.POINTER(ClusterGrid), C.c_void_p]),
    "mnv_render_guided_fused_track_ex": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect, C.POINTER(FrameInputs), C.c_void_p,
                                                   C.POINTER(ClusterGrid), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p]),
    "mnv_get_samples_from_voxels_accel": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(ClusterGrid), C.c_void_p]),
    "mnv_render_voxels_accel_visit": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_get_samples_from_voxels_accel_visit": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect, C.c_void_p, C.c_void_p,
                                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                                          C.POINTER(ClusterGrid), C.c_void_p]),
    "mnv_render_guided_fused": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect, C.c_void_p, C.POINTER(ClusterGrid),
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_render_guided_fused_track": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect, C.c_void_p, C.POINTER(ClusterGrid),
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_render_guided_fused_part": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect, Partition, C.c_void_p,
                                               C.POINTER(ClusterGrid), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_render_nerf_results": (C.c_int, [C.POINTER(TreeView), C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect,
                                          C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_add_children_and_generate_samples": (C.c_int, [C.POINTER(TreeEdit), C.POINTER(RenderOptions), C.c_void_p, C.c_int32, C.c_void_p,
                                                        C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(ClusterGrid), C.c_void_p]),
    "mnv_generate_samples": (C.c_int, [C.POINTER(TreeEdit), C.POINTER(RenderOptions), C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.POINTER(ClusterGrid), C.c_void_p]),
    "mnv_adjust_parents_and_children": (C.c_int, [C.POINTER(TreeEdit), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_select_split_candidates": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]),
    "mnv_select_sample_candidates": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]),
    "mnv_apply_split_results": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "mnv_apply_sample_results": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "mnv_prune_tree": (C.c_int, [C.POINTER(TreeEdit), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int32), C.c_void_p]),
    "mnv_prune_tree_accel": (C.c_int, [C.POINTER(TreeEdit), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32), C.c_void_p]),
    "mnv_mlp_param_count": (C.c_size_t, [C.POINTER(MlpDesc)]),
    "mnv_mlp_create": (C.c_int, [C.POINTER(MlpDesc), C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mnv_mlp_destroy": (None, [C.c_void_p]),
    "mnv_query_submodules": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "mnv_fill_uniform": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p]),
    "mnv_compact_guided_samples": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
    "mnv_renderer_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "mnv_renderer_destroy": (None, [C.c_void_p]),
    "mnv_renderer_set": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "mnv_renderer_load_model": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mnv_renderer_set_model": (C.c_int, [C.c_void_p, C.POINTER(MlpDesc), C.c_void_p, C.c_size_t, C.POINTER(ClusterGrid)]),
    "mnv_renderer_resize": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "mnv_renderer_options": (C.POINTER(RenderOptions), [C.c_void_p]),
    "mnv_renderer_set_camera": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mnv_renderer_set_seed": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int32]),
    "mnv_renderer_render": (C.c_int, [C.c_void_p, C.POINTER(RendererStats)]),
    "mnv_renderer_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_renderer_sync_tree": (C.c_int, [C.c_void_p]),
    "mnv_renderer_set_frames_in_flight": (C.c_int, [C.c_void_p, C.c_int32]),
    "mnv_renderer_set_guided_in_flight": (C.c_int, [C.c_void_p, C.c_int]),
    "mnv_renderer_slot_guided_samples": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    "mnv_renderer_last_slot": (C.c_int32, [C.c_void_p]),
    "mnv_renderer_set_fused_guided": (C.c_int, [C.c_void_p, C.c_int]),
    "mnv_renderer_set_frame_inputs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_render_voxels_accel_visit_ex": (C.c_int, [C.c_void_p, C.POINTER(CameraStruct), C.POINTER(RenderOptions), Rect, C.POINTER(FrameInputs), C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mnv_renderer_set_ranks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "mnv_renderer_download_slot": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mnv_n3tree_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "mnv_n3tree_from_arrays": (C.c_int, [C.POINTER(TreeView), C.POINTER(C.c_void_p)]),
    "mnv_n3tree_free": (None, [C.c_void_p]),
    "mnv_n3tree_host_view": (C.c_int, [C.c_void_p, C.POINTER(TreeView)]),
    "mnv_n3tree_move_to_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "mnv_n3tree_device_view": (C.c_int, [C.c_void_p, C.POINTER(TreeView)]),
    "mnv_n3tree_accel": (C.c_void_p, [C.c_void_p]),
    "mnv_n3tree_save_npz": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mnv_data_format_parse": (None, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mnv_data_format_to_string": (C.c_int, [C.c_int32, C.c_int32, C.c_char_p, C.c_size_t]),
    "mnv_synth_random_tree": (C.c_int, [C.POINTER(SynthRandomParams), C.POINTER(C.c_void_p)]),
    "mnv_synth_shell_tree": (C.c_int, [C.POINTER(SynthShellParams), C.POINTER(C.c_void_p)]),
    "mnv_synth_terrain_tree": (C.c_int, [C.POINTER(SynthTerrainParams), C.POINTER(C.c_void_p)]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load libmnv.so (built by ``__graft_entry__.build()`` / ``make``); raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'). There is no fallback path.")
        # One HIP runtime per process: torch ships its own libamdhip64.so.7 and must load it first;
        # libmnv.so (linked against the same SONAME) then binds to that copy.  Loading ours first
        # leaves two half-initialised runtimes and hipGetDeviceCount() == 0.
        import torch  # noqa: F401

        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


_hooks_lib: Optional[C.CDLL] = None
HOOKS_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "testhooks", "libmnv.so")


def hooks_lib() -> C.CDLL:
    """The test-hook build of the library (testhooks/libmnv.so: the same object code but for the knobs / transport units) as a SECOND binding in
    this process -- for the one switch the shipped library does not have: mnv_hook_set_ref_table_min_rays, with which the tests run
    mnv_render_voxels' per-launch lookup table at any frame size.  (When MNV_LIB_PATH already selects the hooks build it is that library.)"""
    global _hooks_lib
    if _hooks_lib is None:
        lib()   # torch's HIP runtime first (see lib())
        if os.path.abspath(LIB_PATH) == os.path.abspath(HOOKS_LIB_PATH):
            h = lib()
        else:
            if not os.path.exists(HOOKS_LIB_PATH):
                raise ImportError(f"{HOOKS_LIB_PATH} is missing (make builds it beside libmnv.so)")
            h = C.CDLL(HOOKS_LIB_PATH)
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(h, name)
                fn.restype = res
                fn.argtypes = args
        h.mnv_hook_set_ref_table_min_rays.restype = None
        h.mnv_hook_set_ref_table_min_rays.argtypes = [C.c_longlong]
        _hooks_lib = h
    return _hooks_lib


def built_source_sha() -> str:
    """Hash of the sources the loaded libmnv.so was built from (mnv_source_sha)."""
    return lib().mnv_source_sha().decode()


def shipped_source_sha() -> str:
    """The same hash over the sources in this tree (the Makefile's SRC_SHA: sorted relative names, contents concatenated)."""
    import glob
    import hashlib

    here = os.path.dirname(os.path.abspath(__file__))
    names = []
    for pat in ("csrc/*.hip", "csrc/*.h", "csrc/*.cpp", "host/*.cpp", "host/*.hpp", "../include/*.h"):
        names += [os.path.relpath(f, here) for f in glob.glob(os.path.join(here, pat))]
    names = sorted(set(names) - {os.path.join("csrc", "mnv_build_info.cpp")})   # (the unit the hash is compiled INTO is not part of it)
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(here, n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _check(rc: int, h=None) -> None:
    if rc != 0:
        raise MnvError(rc, (h or lib()).mnv_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    return int(lib().mnv_device_count())


def parse_data_format(s: str):
    f, b = C.c_int32(), C.c_int32()
    lib().mnv_data_format_parse(s.encode(), C.byref(f), C.byref(b))
    return f.value, b.value


def data_format_to_string(fmt: int, basis_dim: int) -> str:
    buf = C.create_string_buffer(64)
    _check(lib().mnv_data_format_to_string(fmt, basis_dim, buf, 64))
    return buf.value.decode()


def _f3(v: Sequence[float]):
    return (C.c_float * 3)(*[float(x) for x in v])


class Camera:
    """Pose model + intrinsics of viewer::Camera (reference src/camera.cpp:29-130)."""

    def __init__(self, width=256, height=256, fx=1111.0, fy=-1.0, cx=-1.0, cy=-1.0):
        self.c = CameraStruct()
        lib().mnv_camera_init(C.byref(self.c), width, height, fx, fy, cx, cy)

    def set_pose(self, center, v_back, v_world_up=(0.0, 0.0, 1.0)) -> "Camera":
        lib().mnv_camera_set_pose(C.byref(self.c), _f3(center), _f3(v_back), _f3(v_world_up))
        return self

    @property
    def width(self):
        return self.c.width

    @property
    def height(self):
        return self.c.height

    @property
    def c2w(self) -> np.ndarray:
        return np.array(list(self.c.c2w), dtype=np.float32)


def camera_drag(cam: "Camera", center, v_back, v_world_up, origin, movement_speed, is_pan, about_origin, start_xy, end_xy):
    """Camera::begin_drag / drag_update / end_drag (+ the next frame's _update) on a camera at the given pose: -> (center, v_back, origin, c2w[12])."""
    c, b, u, o = _f3(center), _f3(v_back), _f3(v_world_up), _f3(origin)
    lib().mnv_camera_drag(C.byref(cam.c), c, b, u, o, float(movement_speed), int(is_pan), int(about_origin), float(start_xy[0]), float(start_xy[1]),
                          float(end_xy[0]), float(end_xy[1]))
    return np.float32(list(c)), np.float32(list(b)), np.float32(list(o)), np.float32(list(cam.c.c2w))


def orbit_camera(width, height, fx, radius, azimuth_deg, elevation_deg, fy=-1.0) -> Camera:
    """Camera on a sphere of `radius` around the world origin looking at the origin
    (SURVEY.md 8(d) cfg2 poses).  Pure float64 host trigonometry -> float32 pose vectors."""
    az, el = np.deg2rad(azimuth_deg), np.deg2rad(elevation_deg)
    back = np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
    center = (radius * back).astype(np.float32)
    return Camera(width, height, fx, fy).set_pose(center, back.astype(np.float32))


class N3Tree:
    """Handle to the C++ viewer::N3Tree (host arrays + optional device copy + accel)."""

    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)

    def __del__(self):
        try:
            if self._h:
                lib().mnv_n3tree_free(self._h)
                self._h = None
        except Exception:
            pass

    @classmethod
    def open(cls, path: str) -> "N3Tree":
        h = C.c_void_p()
        _check(lib().mnv_n3tree_open(os.fsencode(path), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_arrays(cls, data: np.ndarray, child: np.ndarray, *, data_format: str, offset=(0.5, 0.5, 0.5),
                    scale=(0.5, 0.5, 0.5), parent: Optional[np.ndarray] = None) -> "N3Tree":
        data = np.ascontiguousarray(data)
        child = np.ascontiguousarray(child, dtype=np.int32)
        if data.dtype == np.float16:
            data = data.view(np.uint16)
        assert data.dtype == np.uint16
        cap = child.shape[0]
        v = TreeView()
        v.data = data.ctypes.data
        v.child = child.ctypes.data
        if parent is not None:
            parent = np.ascontiguousarray(parent, dtype=np.int32)
            v.parent = parent.ctypes.data
        v.offset = _f3(offset)
        v.scale = _f3(scale)
        v.N = 2
        v.data_dim = data.size // (cap * 8)
        v.format, v.basis_dim = parse_data_format(data_format)
        v.capacity = cap
        h = C.c_void_p()
        _check(lib().mnv_n3tree_from_arrays(C.byref(v), C.byref(h)))
        return cls(h.value)

    @classmethod
    def synth_random(cls, depth=4, basis_dim=1, fmt=FORMAT_SH, refine_prob=0.6, empty_prob=0.5, sigma_max=30.0,
                     coef_sd=1.5, offset=(0.5, 0.5, 0.5), scale=(0.5, 0.5, 0.5), seed=0) -> "N3Tree":
        p = SynthRandomParams(depth, fmt, basis_dim, refine_prob, empty_prob, sigma_max, coef_sd, _f3(offset), _f3(scale), seed)
        h = C.c_void_p()
        _check(lib().mnv_synth_random_tree(C.byref(p), C.byref(h)))
        return cls(h.value)

    @classmethod
    def synth_shell(cls, depth=10, basis_dim=9, radius=0.35, half_thickness=1.5 / 1024, sigma_lo=50.0, sigma_hi=400.0,
                    offset=(0.5, 0.5, 0.5), scale=(0.5, 0.5, 0.5), seed=0) -> "N3Tree":
        p = SynthShellParams(depth, basis_dim, radius, half_thickness, sigma_lo, sigma_hi, _f3(offset), _f3(scale), seed)
        h = C.c_void_p()
        _check(lib().mnv_synth_shell_tree(C.byref(p), C.byref(h)))
        return cls(h.value)

    @classmethod
    def synth_terrain(cls, depth=10, basis_dim=9, bricks_y=4, bricks_z=2, noise_cells=6, base=0.25, amplitude=0.35,
                      thickness=1.5 / 1024, sigma_lo=50.0, sigma_hi=400.0, offset=(0.5, 0.5, 0.5), scale=(0.5, 0.125, 0.125),
                      seed=0) -> "N3Tree":
        p = SynthTerrainParams(depth, basis_dim, bricks_y, bricks_z, noise_cells, base, amplitude, thickness, sigma_lo, sigma_hi,
                               _f3(offset), _f3(scale), seed)
        h = C.c_void_p()
        _check(lib().mnv_synth_terrain_tree(C.byref(p), C.byref(h)))
        return cls(h.value)

    def host_view(self) -> TreeView:
        v = TreeView()
        _check(lib().mnv_n3tree_host_view(self._h, C.byref(v)))
        return v

    def device_view(self) -> TreeView:
        v = TreeView()
        _check(lib().mnv_n3tree_device_view(self._h, C.byref(v)))
        return v

    def host_arrays(self):
        """(data uint16 [cap,8,data_dim], child int32 [cap,8], parent int32 [cap]) views of the C++ arrays."""
        v = self.host_view()
        cap, dd = v.capacity, v.data_dim
        data = np.ctypeslib.as_array(C.cast(v.data, C.POINTER(C.c_uint16)), shape=(cap, 8, dd))
        child = np.ctypeslib.as_array(C.cast(v.child, C.POINTER(C.c_int32)), shape=(cap, 8))
        parent = np.ctypeslib.as_array(C.cast(v.parent, C.POINTER(C.c_int32)), shape=(cap,))
        return data, child, parent

    @property
    def capacity(self) -> int:
        return self.host_view().capacity

    @property
    def data_format(self) -> str:
        v = self.host_view()
        return data_format_to_string(v.format, v.basis_dim)

    def move_to_device(self, max_capacity: int = 0, need_parent=False, need_sample_counts=False, stream: int = 0):
        _check(lib().mnv_n3tree_move_to_device(self._h, max_capacity, int(need_parent), int(need_sample_counts), C.c_void_p(stream)))
        return self

    @property
    def accel(self) -> int:
        a = lib().mnv_n3tree_accel(self._h)
        if not a:
            raise MnvError(MNV_E_INVALID, "tree has no accel (call move_to_device first)")
        return a

    def save_npz(self, path: str) -> None:
        _check(lib().mnv_n3tree_save_npz(self._h, os.fsencode(path)))


def _ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (or a raw int / None)."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def _frame_inputs(tmax_px, rgba8_init, pixels: int):
    """mnv_frame_inputs of the reference's offscreen == false call shape, or None when neither array is given."""
    if tmax_px is None and rgba8_init is None:
        return None
    import torch

    if tmax_px is not None and (not tmax_px.is_cuda or not tmax_px.is_contiguous() or tmax_px.dtype != torch.float32 or tmax_px.numel() < pixels):
        raise MnvError(MNV_E_INVALID, f"tmax_px must be a contiguous float32 device tensor with at least {pixels} elements")
    _check_out("rgba8_init", rgba8_init, pixels, "u8")
    return FrameInputs(tmax_px.data_ptr() if tmax_px is not None else None, rgba8_init.data_ptr() if rgba8_init is not None else None)


def render_voxels(tree_view: TreeView, cam: Camera, opt: RenderOptions, tile=None, rgba=None, rgba8=None,
                  split_track=None, sample_track=None, visited=None, track_visit=False, stream: int = 0, tmax_px=None, rgba8_init=None,
                  table_min_rays: Optional[int] = None) -> None:
    """viewer::render_voxels on reference-layout device arrays (asynchronous on `stream`).  tmax_px / rgba8_init: the depth attachment and
    the image under the volume of the reference's offscreen == false call shape (indexed like the outputs; rgba8_init may be rgba8).
    table_min_rays (tests only): run the call on the test-hook build with its per-launch lookup table forced on (0) / off (-1) / from that
    many rays -- the shipped library has no such switch (65536 rays)."""
    if tile is None:
        tile = (0, 0, cam.width, cam.height)
    h = lib()
    if table_min_rays is not None:
        h = hooks_lib()
        h.mnv_hook_set_ref_table_min_rays(int(table_min_rays))
    inputs = _frame_inputs(tmax_px, rgba8_init, tile[2] * tile[3])
    try:
        with _timed(stream):
            if inputs is not None:
                _check(h.mnv_render_voxels_ex(C.byref(tree_view), C.byref(cam.c), C.byref(opt), Rect(*tile), C.byref(inputs), _ptr(rgba), _ptr(rgba8),
                                              _ptr(split_track), _ptr(sample_track), _ptr(visited), int(track_visit), C.c_void_p(stream)), h)
            else:
                _check(h.mnv_render_voxels(C.byref(tree_view), C.byref(cam.c), C.byref(opt), Rect(*tile), _ptr(rgba), _ptr(rgba8),
                                           _ptr(split_track), _ptr(sample_track), _ptr(visited), int(track_visit), C.c_void_p(stream)), h)
    finally:
        if table_min_rays is not None:
            h.mnv_hook_set_ref_table_min_rays(1 << 16)


def accel_create(tree_view: TreeView, max_capacity: int = 0, stream: int = 0) -> int:
    """Packed accel of a device tree view with room for max_capacity chunks (0: the tree's capacity); free with accel_destroy."""
    h = C.c_void_p()
    _check(lib().mnv_accel_create_reserved(C.byref(tree_view), max(max_capacity, tree_view.capacity), C.c_void_p(stream), C.byref(h)))
    return h.value


def accel_rebuild(accel: int, tree_view: TreeView, stream: int = 0) -> None:
    """Rebuild the accel in place from the tree (after a prune renumbered the chunks)."""
    _check(lib().mnv_accel_rebuild(C.c_void_p(accel), C.byref(tree_view), C.c_void_p(stream)))


def accel_destroy(accel: int) -> None:
    _pinned_fused_kernel.pop(accel, None)
    lib().mnv_accel_destroy(C.c_void_p(accel))


def accel_refresh(accel: int, tree_view: TreeView, old_capacity: int, changed_nodes=None, stream: int = 0) -> None:
    """Patch the accel after chunks [old_capacity, tree_view.capacity) were appended and / or the rows of changed_nodes
    (device int32 [n][2]) were rewritten."""
    n = 0 if changed_nodes is None else int(changed_nodes.shape[0])
    _check(lib().mnv_accel_refresh(C.c_void_p(accel), C.byref(tree_view), old_capacity, _ptr(changed_nodes), n, C.c_void_p(stream)))


def _check_out(name: str, t, pixels: int, dtype_name: str) -> None:
    """An output tensor must be a contiguous device tensor of the right element type with room for `pixels` RGBA pixels: the library
    writes through the raw pointer."""
    if t is None or isinstance(t, int):
        return
    import torch

    want = torch.float32 if dtype_name == "f32" else torch.uint8
    if not t.is_cuda or not t.is_contiguous() or t.dtype != want or t.numel() < pixels * 4:
        raise MnvError(MNV_E_INVALID, f"{name} must be a contiguous {want} device tensor with at least {pixels} RGBA pixels "
                                      f"(got {tuple(t.shape)} {t.dtype}, device {t.device})")


def _pixels(tile, n_frames: int, part) -> int:
    """Pixels the launch may write: frames x tile, or frames x j_max macro tiles for a partitioned launch."""
    _, _, w, h = tile
    if part is None or not (part[1] > 1 or (part[1] == 1 and part[2] > 0)):
        return n_frames * w * h
    rank, world, tw, th = part[:4]
    period = part[4] if len(part) > 4 else 0
    j_max = max(partition_local_tiles(tile, r, world, tw, th, period) for r in range(world))
    return n_frames * j_max * tw * th


def render_voxels_accel(accel: int, cam: Camera, opt: RenderOptions, tile=None, rgba=None, rgba8=None, stream: int = 0, tmax_px=None,
                        rgba8_init=None) -> None:
    """The tuned march on the packed layout (asynchronous on `stream`); tmax_px / rgba8_init as in render_voxels."""
    if tile is None:
        tile = (0, 0, cam.width, cam.height)
    _check_out("rgba", rgba, tile[2] * tile[3], "f32")
    _check_out("rgba8", rgba8, tile[2] * tile[3], "u8")
    inputs = _frame_inputs(tmax_px, rgba8_init, tile[2] * tile[3])
    with _timed(stream):
        if inputs is not None:
            _check(lib().mnv_render_voxels_accel_ex(C.c_void_p(accel), C.byref(cam.c), C.byref(opt), Rect(*tile), C.byref(inputs), _ptr(rgba),
                                                    _ptr(rgba8), C.c_void_p(stream)))
            return
        _check(lib().mnv_render_voxels_accel(C.c_void_p(accel), C.byref(cam.c), C.byref(opt), Rect(*tile), _ptr(rgba), _ptr(rgba8),
                                             C.c_void_p(stream)))


def render_voxels_accel_track(accel: int, cam: Camera, opt: RenderOptions, tile=None, rgba=None, rgba8=None,
                              split_track=None, sample_track=None, sample_counts=None, stream: int = 0) -> None:
    """The tuned march with the refinement trackers (rows as render_voxels writes them)."""
    if tile is None:
        tile = (0, 0, cam.width, cam.height)
    with _timed(stream):
        _check(lib().mnv_render_voxels_accel_track(C.c_void_p(accel), C.byref(cam.c), C.byref(opt), Rect(*tile), _ptr(rgba),
                                                   _ptr(rgba8), _ptr(split_track), _ptr(sample_track), _ptr(sample_counts),
                                                   C.c_void_p(stream)))


def render_voxels_accel_visit(accel: int, cam: Camera, opt: RenderOptions, visited, parent, tile=None, rgba=None, rgba8=None, split_track=None,
                              sample_track=None, sample_counts=None, stream: int = 0, part=None, tmax_px=None, rgba8_init=None) -> None:
    """Tracker march on the packed accel that also leaves the reference's visit marks (march marks leaf chunks, closure adds ancestors).
    part = (rank, world, tile_w, tile_h[, root_period]): only that rank's macro tiles, pixels and tracker rows in compact tile order.
    tmax_px / rgba8_init: the reference's offscreen == false inputs (whole-frame launches only)."""
    if tile is None:
        tile = (0, 0, cam.width, cam.height)
    inputs = _frame_inputs(tmax_px, rgba8_init, tile[2] * tile[3])
    if inputs is not None:
        if part is not None:
            raise MnvError(MNV_E_INVALID, "frame inputs are for whole-frame launches")
        _check(lib().mnv_render_voxels_accel_visit_ex(C.c_void_p(accel), C.byref(cam.c), C.byref(opt), Rect(*tile), C.byref(inputs), _ptr(rgba), _ptr(rgba8),
                                                      _ptr(split_track), _ptr(sample_track), _ptr(sample_counts), _ptr(visited), _ptr(parent),
                                                      C.c_void_p(stream)))
        return
    if part is not None:
        px = _pixels(tile, 1, part)
        _check_out("rgba", rgba, px, "f32")
        _check_out("rgba8", rgba8, px, "u8")
        _check(lib().mnv_render_voxels_accel_visit_part(C.c_void_p(accel), C.byref(cam.c), C.byref(opt), Rect(*tile), Partition(*part), _ptr(rgba), _ptr(rgba8),
                                                        _ptr(split_track), _ptr(sample_track), _ptr(sample_counts), _ptr(visited), _ptr(parent),
                                                        C.c_void_p(stream)))
        return
    _check(lib().mnv_render_voxels_accel_visit(C.c_void_p(accel), C.byref(cam.c), C.byref(opt), Rect(*tile), _ptr(rgba), _ptr(rgba8), _ptr(split_track),
                                               _ptr(sample_track), _ptr(sample_counts), _ptr(visited), _ptr(parent), C.c_void_p(stream)))


def get_samples_from_voxels_accel_visit(accel: int, cam: Camera, opt: RenderOptions, visited, parent, num_samples, samples, cluster_indices,
                                        grid: ClusterGrid, split_track=None, sample_track=None, sample_counts=None, tile=None, stream: int = 0,
                                        tmax_px=None) -> None:
    """tmax_px: the depth attachment of the reference's offscreen == false call (every ray stops there), indexed like the pixels."""
    if tile is None:
        tile = (0, 0, cam.width, cam.height)
    inputs = _frame_inputs(tmax_px, None, tile[2] * tile[3])
    if inputs is not None:
        _check(lib().mnv_get_samples_from_voxels_accel_visit_ex(C.c_void_p(accel), C.byref(cam.c), C.byref(opt), Rect(*tile), C.byref(inputs),
                                                                _ptr(split_track), _ptr(sample_track), _ptr(sample_counts), _ptr(visited), _ptr(parent),
                                                                _ptr(num_samples), _ptr(samples), int(samples.shape[-1]), _ptr(cluster_indices),
                                                                C.byref(grid), C.c_void_p(stream)))
        return
    _check(lib().mnv_get_samples_from_voxels_accel_visit(C.c_void_p(accel), C.byref(cam.c), C.byref(opt), Rect(*tile), _ptr(split_track), _ptr(sample_track),
                                                         _ptr(sample_counts), _ptr(visited), _ptr(parent), _ptr(num_samples), _ptr(samples),
                                                         int(samples.shape[-1]), _ptr(cluster_indices), C.byref(grid), C.c_void_p(stream)))


def accel_set_colour_math(accel: int, mode: int) -> None:
    """Colour math of ONE accel: 0 exact (bit-identical to the oracle, the default), 1 hardware exp2 / rcp in the colour sigmoid (colours move
    ~1e-7).  There is no process-wide switch."""
    _check(lib().mnv_accel_set_colour_math(C.c_void_p(accel), int(mode)))


# The C library holds the choice of the fused guided-sampling kernel and its diagnostics buffer PER ACCEL (mnv_accel_set_fused_kernel /
# mnv_accel_set_fused_diag).  The two module-level setters below are a convenience of this harness for tests and tools whose trees are
# created deep inside helpers: render_guided_fused* (and Renderer.render) hand the harness's current choice to the accel they launch on,
# unless accel_set_fused_kernel pinned that accel.
_harness_fused_kernel: Optional[int] = None
_harness_fused_diag = None
_harness_diag_used = False
_pinned_fused_kernel: dict = {}


def accel_set_fused_kernel(accel: int, version: int) -> None:
    """Fused guided-sampling kernel of ONE accel: 0 = by network size (default), 1 = one-role wavefronts, 2 = producer / consumer wavefronts;
    negative = un-pin (the harness default of set_fused_kernel applies again)."""
    if version < 0:
        _pinned_fused_kernel.pop(accel, None)
        _check(lib().mnv_accel_set_fused_kernel(C.c_void_p(accel), int(_harness_fused_kernel or 0)))
    else:
        _pinned_fused_kernel[accel] = int(version)
        _check(lib().mnv_accel_set_fused_kernel(C.c_void_p(accel), int(version)))


def accel_set_fused_diag(accel: int, words32=None) -> None:
    """Diagnostics buffer of the fused kernels on ONE accel: a device int64 tensor of at least 32 words, or None.  The library keeps only the
    address: the caller keeps the tensor alive while it is set."""
    if words32 is not None and words32.numel() < 32:
        raise ValueError("the diagnostics buffer holds 32 words")
    _check(lib().mnv_accel_set_fused_diag(C.c_void_p(accel), _ptr(words32)))


def set_fused_kernel(version: int) -> None:
    """Harness default for the accels render_guided_fused* launch on (see above): 0 = by network size, 1, 2."""
    global _harness_fused_kernel
    _harness_fused_kernel = int(version) if version in (1, 2) else None


def set_fused_diag(words32=None) -> None:
    """Harness default: the diagnostics buffer handed to every accel render_guided_fused* launches on from now on (None: none).  The tensor
    is held here, so that a caller that drops its own reference cannot leave the kernels adding into freed memory."""
    global _harness_fused_diag, _harness_diag_used
    if words32 is not None and words32.numel() < 32:
        raise ValueError("the diagnostics buffer holds 32 words")
    _harness_fused_diag = words32
    _harness_diag_used = True


def _fused_defaults(accel: int) -> None:
    if accel not in _pinned_fused_kernel and (_harness_fused_kernel is not None or _harness_diag_used):
        _check(lib().mnv_accel_set_fused_kernel(C.c_void_p(accel), int(_harness_fused_kernel or 0)))
    if _harness_diag_used:
        _check(lib().mnv_accel_set_fused_diag(C.c_void_p(accel), _ptr(_harness_fused_diag)))


def accel_fused_faults(accel: int) -> int:
    """Spin-waits of the producer / consumer kernel that its watchdog abandoned since the accel was created (0 on a healthy build);
    waits for the device."""
    n = C.c_uint32(0)
    _check(lib().mnv_accel_fused_faults(C.c_void_p(accel), C.byref(n)))
    return int(n.value)


def set_tree_cache(enable: bool) -> None:
    """render_voxels (reference layout) keeps the packed re-layout of every tree it has seen (include/mnv.h: mnv_set_tree_cache);
    after editing a tree's arrays in place call tree_invalidate()."""
    lib().mnv_set_tree_cache(1 if enable else 0)


def tree_invalidate(child=None) -> None:
    """Forget the cached re-layout of the tree whose `child` array is this device tensor / address (None: all of them)."""
    if child is None:
        lib().mnv_tree_invalidate(None)
    else:
        lib().mnv_tree_invalidate(C.c_void_p(child if isinstance(child, int) else child.data_ptr()))


def assemble_tiles(gathered, frames, width: int, height: int, world: int, tile_w: int, tile_h: int, n_frames: int = 1, stream: int = 0,
                   root_period: int = 0) -> None:
    """Rank 0's un-permute of the gathered tile buffers into frames (device tensors, RGBA8 or float RGBA)."""
    bpp = gathered.element_size() * 4
    _check(lib().mnv_assemble_tiles(_ptr(gathered), _ptr(frames), width, height, Partition(0, world, tile_w, tile_h, root_period), n_frames, bpp,
                                    C.c_void_p(stream)))


def partition_local_tiles(tile, rank: int, world: int, tile_w: int, tile_h: int, root_period: int = 0) -> int:
    return int(lib().mnv_partition_local_tiles(Rect(*tile), Partition(rank, world, tile_w, tile_h, root_period)))


def render_voxels_accel_part(accel: int, cam: Camera, opt: RenderOptions, rank: int, world: int, tile_w: int, tile_h: int,
                             tile=None, rgba=None, rgba8=None, stream: int = 0, root_period: int = 0) -> None:
    """Render the macro tiles of `rank` (m % world == rank, or the root-relieving deal of mnv_partition.root_period) of `tile` into a
    compact local-tile-major buffer [local_tiles][tile_h][tile_w][4] (see mnv_partition in include/mnv.h)."""
    if tile is None:
        tile = (0, 0, cam.width, cam.height)
    n_local = partition_local_tiles(tile, rank, world, tile_w, tile_h, root_period)
    _check_out("rgba", rgba, n_local * tile_w * tile_h, "f32")
    _check_out("rgba8", rgba8, n_local * tile_w * tile_h, "u8")
    with _timed(stream):
        _check(lib().mnv_render_voxels_accel_part(C.c_void_p(accel), C.byref(cam.c), C.byref(opt), Rect(*tile),
                                                  Partition(rank, world, tile_w, tile_h, root_period), _ptr(rgba), _ptr(rgba8), C.c_void_p(stream)))


def get_samples_from_voxels(tree_view: TreeView, cam: Camera, opt: RenderOptions, num_samples, samples, cluster_indices,
                            grid: ClusterGrid, split_track=None, sample_track=None, visited=None, track_visit=False,
                            tile=None, stream: int = 0, tmax_px=None) -> None:
    """viewer::get_samples_from_voxels (reference include/cuda/renderer_kernel.hpp:36-52).  tmax_px: the depth attachment of its
    offscreen == false call (renderer_kernel.cu:354-357), a float32 device tensor indexed like the pixels."""
    if tile is None:
        tile = (0, 0, cam.width, cam.height)
    inputs = _frame_inputs(tmax_px, None, tile[2] * tile[3])
    if inputs is not None:
        _check(lib().mnv_get_samples_from_voxels_ex(C.byref(tree_view), C.byref(cam.c), C.byref(opt), Rect(*tile), C.byref(inputs), _ptr(split_track),
                                                    _ptr(sample_track), _ptr(visited), int(track_visit), _ptr(num_samples), _ptr(samples),
                                                    int(samples.shape[-1]), _ptr(cluster_indices), C.byref(grid), C.c_void_p(stream)))
        return
    _check(lib().mnv_get_samples_from_voxels(C.byref(tree_view), C.byref(cam.c), C.byref(opt), Rect(*tile), _ptr(split_track),
                                             _ptr(sample_track), _ptr(visited), int(track_visit), _ptr(num_samples), _ptr(samples),
                                             int(samples.shape[-1]), _ptr(cluster_indices), C.byref(grid), C.c_void_p(stream)))


def get_samples_from_voxels_accel(accel: int, cam: Camera, opt: RenderOptions, num_samples, samples, cluster_indices, grid: ClusterGrid,
                                  split_track=None, sample_track=None, sample_counts=None, tile=None, stream: int = 0, tmax_px=None) -> None:
    """get_samples_from_voxels on the packed accel (no visit marks)."""
    if tile is None:
        tile = (0, 0, cam.width, cam.height)
    if tmax_px is not None:
        get_samples_from_voxels_accel_visit(accel, cam, opt, None, None, num_samples, samples, cluster_indices, grid, split_track, sample_track,
                                            sample_counts, tile, stream, tmax_px)
        return
    _check(lib().mnv_get_samples_from_voxels_accel(C.c_void_p(accel), C.byref(cam.c), C.byref(opt), Rect(*tile), _ptr(split_track),
                                                   _ptr(sample_track), _ptr(sample_counts), _ptr(num_samples), _ptr(samples),
                                                   int(samples.shape[-1]), _ptr(cluster_indices), C.byref(grid), C.c_void_p(stream)))


def render_nerf_results(tree_view: TreeView, cam: Camera, opt: RenderOptions, sample_values, z_vals, offsets, rgba=None,
                        rgba8=None, tile=None, stream: int = 0) -> None:
    """viewer::render_nerf_results (reference include/cuda/renderer_kernel.hpp:12-21)."""
    if tile is None:
        tile = (0, 0, cam.width, cam.height)
    _check(lib().mnv_render_nerf_results(C.byref(tree_view), C.byref(cam.c), C.byref(opt), Rect(*tile), _ptr(sample_values),
                                         int(sample_values.shape[-1]), _ptr(z_vals), _ptr(offsets), _ptr(rgba), _ptr(rgba8),
                                         C.c_void_p(stream)))


def render_guided_fused(accel: int, cam: Camera, opt: RenderOptions, mlp: "Mlp", grid: ClusterGrid, tile=None, rgba=None, rgba8=None,
                        sample_counter=None, split_track=None, sample_track=None, sample_counts=None, visited=None, parent=None,
                        stream: int = 0, tmax_px=None) -> None:
    """The guided-sampling frame as one kernel (mnv_render_guided_fused[_track]).  `sample_counter`: optional device int64 tensor,
    incremented by the number of network evaluations; trackers / visit marks as in render_voxels_accel_visit.  tmax_px: the depth
    attachment of the reference's offscreen == false frame (the image under the volume has weight 0 in that frame: not an input)."""
    if tile is None:
        tile = (0, 0, cam.width, cam.height)
    _fused_defaults(accel)
    inputs = _frame_inputs(tmax_px, None, tile[2] * tile[3])
    if inputs is not None:
        _check(lib().mnv_render_guided_fused_track_ex(C.c_void_p(accel), C.byref(cam.c), C.byref(opt), Rect(*tile), C.byref(inputs), mlp._h, C.byref(grid),
                                                      _ptr(rgba), _ptr(rgba8), _ptr(split_track), _ptr(sample_track), _ptr(sample_counts), _ptr(visited),
                                                      _ptr(parent), _ptr(sample_counter), C.c_void_p(stream)))
        return
    _check(lib().mnv_render_guided_fused_track(C.c_void_p(accel), C.byref(cam.c), C.byref(opt), Rect(*tile), mlp._h, C.byref(grid), _ptr(rgba),
                                               _ptr(rgba8), _ptr(split_track), _ptr(sample_track), _ptr(sample_counts), _ptr(visited), _ptr(parent),
                                               _ptr(sample_counter), C.c_void_p(stream)))


def render_guided_fused_part(accel: int, cam: Camera, opt: RenderOptions, mlp: "Mlp", grid: ClusterGrid, part, tile=None, rgba=None, rgba8=None,
                             sample_counter=None, stream: int = 0, split_track=None, sample_track=None, sample_counts=None, visited=None,
                             parent=None) -> None:
    """One rank's macro tiles of the fused guided-sampling frame (part = (rank, world, tile_w, tile_h[, root_period])); with trackers /
    visit marks: mnv_render_guided_fused_track_part (rows in the compact tile order of the pixels)."""
    if tile is None:
        tile = (0, 0, cam.width, cam.height)
    px = _pixels(tile, 1, part)
    _check_out("rgba", rgba, px, "f32")
    _check_out("rgba8", rgba8, px, "u8")
    _fused_defaults(accel)
    if split_track is not None or sample_track is not None or visited is not None:
        _check(lib().mnv_render_guided_fused_track_part(C.c_void_p(accel), C.byref(cam.c), C.byref(opt), Rect(*tile), Partition(*part), mlp._h, C.byref(grid),
                                                        _ptr(rgba), _ptr(rgba8), _ptr(split_track), _ptr(sample_track), _ptr(sample_counts), _ptr(visited),
                                                        _ptr(parent), _ptr(sample_counter), C.c_void_p(stream)))
        return
    _check(lib().mnv_render_guided_fused_part(C.c_void_p(accel), C.byref(cam.c), C.byref(opt), Rect(*tile), Partition(*part), mlp._h, C.byref(grid),
                                              _ptr(rgba), _ptr(rgba8), _ptr(sample_counter), C.c_void_p(stream)))


def tree_edit(child, parent, offset, scale, capacity: int) -> TreeEdit:
    e = TreeEdit()
    e.child, e.parent = _ptr(child), _ptr(parent)
    e.offset, e.scale = _f3(offset), _f3(scale)
    e.N, e.capacity = 2, capacity
    return e


def add_children_and_generate_samples(edit: TreeEdit, opt: RenderOptions, parent_nodes, samples, cluster_indices, visited,
                                      grid: ClusterGrid, stream: int = 0) -> None:
    _check(lib().mnv_add_children_and_generate_samples(C.byref(edit), C.byref(opt), _ptr(parent_nodes), int(parent_nodes.shape[0]),
                                                       _ptr(samples), int(samples.shape[-1]), _ptr(cluster_indices), _ptr(visited),
                                                       C.byref(grid), C.c_void_p(stream)))


def generate_samples(edit: TreeEdit, opt: RenderOptions, nodes, samples, cluster_indices, grid: ClusterGrid, stream: int = 0) -> None:
    _check(lib().mnv_generate_samples(C.byref(edit), C.byref(opt), _ptr(nodes), int(nodes.shape[0]), _ptr(samples),
                                      int(samples.shape[-1]), _ptr(cluster_indices), C.byref(grid), C.c_void_p(stream)))


def adjust_parents_and_children(edit: TreeEdit, first_shift_index: int, to_delete, index_shifts, stream: int = 0) -> None:
    _check(lib().mnv_adjust_parents_and_children(C.byref(edit), first_shift_index, _ptr(to_delete), _ptr(index_shifts), C.c_void_p(stream)))


def select_split_candidates(split_track, max_out: int, nodes_out, stream: int = 0):
    """expand_voxels' vote on a [n][3] device tracker -> (pairs written to nodes_out, qualifying candidates)."""
    n_out, n_cand = C.c_int32(0), C.c_int32(0)
    _check(lib().mnv_select_split_candidates(_ptr(split_track), split_track.numel() // 3, max_out, _ptr(nodes_out), C.byref(n_out),
                                             C.byref(n_cand), C.c_void_p(stream)))
    return n_out.value, n_cand.value


def select_sample_candidates(sample_track, max_out: int, nodes_out, stream: int = 0):
    n_out, n_cand = C.c_int32(0), C.c_int32(0)
    _check(lib().mnv_select_sample_candidates(_ptr(sample_track), sample_track.numel() // 3, max_out, _ptr(nodes_out), C.byref(n_out),
                                              C.byref(n_cand), C.c_void_p(stream)))
    return n_out.value, n_cand.value


def apply_split_results(data, sample_counts, capacity: int, num_parents: int, results, samples_per_corner: int, data_dim: int,
                        stream: int = 0) -> None:
    _check(lib().mnv_apply_split_results(_ptr(data), _ptr(sample_counts), capacity, num_parents, _ptr(results), results.shape[-1],
                                         samples_per_corner, data_dim, C.c_void_p(stream)))


def apply_sample_results(data, sample_counts, nodes, results, samples_per_corner: int, data_dim: int, stream: int = 0) -> None:
    _check(lib().mnv_apply_sample_results(_ptr(data), _ptr(sample_counts), _ptr(nodes), nodes.shape[0], _ptr(results), results.shape[-1],
                                          samples_per_corner, data_dim, C.c_void_p(stream)))


def prune_tree(edit: TreeEdit, data, data_dim: int, sample_counts, visited, max_capacity: int, stream: int = 0, accel: int = 0):
    """-> (new_capacity, num_deleted).  `accel`: the tree's packed accel follows the prune in place (mnv_prune_tree_accel)."""
    new_cap, n_del = C.c_int32(0), C.c_int32(0)
    _check(lib().mnv_prune_tree_accel(C.byref(edit), _ptr(data), data_dim, _ptr(sample_counts), _ptr(visited), max_capacity,
                                      C.c_void_p(accel) if accel else None, C.byref(new_cap), C.byref(n_del), C.c_void_p(stream)))
    return new_cap.value, n_del.value


def mlp_desc(n_clusters=1, pos_octaves=4, dir_octaves=2, need_viewdir=False, n_embeddings=0, embedding_dim=0, hidden_width=64,
             hidden_layers=2, out_dim=5, center=(0.0, 0.0, 0.0), inv_extent=(1.0, 1.0, 1.0)) -> MlpDesc:
    d = MlpDesc(n_clusters, pos_octaves, dir_octaves, int(need_viewdir), n_embeddings, embedding_dim, hidden_width, hidden_layers, out_dim)
    d.center, d.inv_extent = _f3(center), _f3(inv_extent)
    return d


class Mlp:
    """Per-sample sub-module network (include/mnv.h: mnv_mlp_*).  `params`: numpy float16/uint16
    [n_clusters * param_count], cluster-major, in the order documented in the header."""

    def __init__(self, desc: MlpDesc, params, stream: int = 0):
        self.desc = desc
        p = np.ascontiguousarray(params).view(np.uint16).reshape(-1)
        h = C.c_void_p()
        _check(lib().mnv_mlp_create(C.byref(desc), p.ctypes.data, p.size, C.c_void_p(stream), C.byref(h)))
        self._h = h

    @staticmethod
    def param_count(desc: MlpDesc) -> int:
        return int(lib().mnv_mlp_param_count(C.byref(desc)))

    def query(self, cluster_indices, samples, results, n=None, stream: int = 0) -> None:
        """cluster_indices int16 [n], samples float32 [n][cols], results float32 [n][>= out_dim] (device tensors)."""
        n = int(samples.shape[0]) if n is None else n
        _check(lib().mnv_query_submodules(self._h, _ptr(cluster_indices), _ptr(samples), samples.stride(0), n, _ptr(results),
                                          results.stride(0), C.c_void_p(stream)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().mnv_mlp_destroy(self._h)
                self._h = None
        except Exception:  # interpreter shutdown: module globals may be gone
            pass


def fill_uniform(out, seed: int, stream: int = 0) -> None:
    _check(lib().mnv_fill_uniform(_ptr(out), out.numel(), seed & 0xFFFFFFFFFFFFFFFF, C.c_void_p(stream)))


def compact_guided_samples(num_samples, samples, cluster_indices, offsets, z_vals=None, rows=None, clusters=None, stream: int = 0) -> int:
    """cumsum + packing of the emitted guided samples; returns the total.  With the outputs None only `offsets` is filled."""
    n_rays, max_g, dim = samples.shape
    total = C.c_int64(0)
    _check(lib().mnv_compact_guided_samples(_ptr(num_samples), _ptr(samples), _ptr(cluster_indices), n_rays, max_g, dim, _ptr(offsets),
                                            _ptr(z_vals), _ptr(rows), _ptr(clusters), 0 if z_vals is None else z_vals.shape[0],
                                            C.byref(total), C.c_void_p(stream)))
    return total.value


class Renderer:
    """viewer::VolumeRenderer behind the C ABI (mnv_renderer_*): owns camera + options + stream, runs the
    refinement loop when a model is set."""

    def __init__(self):
        h = C.c_void_p()
        _check(lib().mnv_renderer_create(C.byref(h)))
        self._h = h
        self._tree = None
        self.width = self.height = 0

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().mnv_renderer_destroy(self._h)
                self._h = None
        except Exception:  # interpreter shutdown: module globals may be gone
            pass

    @property
    def options(self) -> RenderOptions:
        return lib().mnv_renderer_options(self._h).contents

    def set(self, tree: "N3Tree", max_tree_capacity: int) -> None:
        _check(lib().mnv_renderer_set(self._h, tree._h, max_tree_capacity))
        self._tree = tree  # keep alive

    def load_model(self, path: str) -> None:
        _check(lib().mnv_renderer_load_model(self._h, os.fsencode(path)))

    def set_model(self, desc: MlpDesc, params, grid: ClusterGrid) -> None:
        p = np.ascontiguousarray(params).view(np.uint16).reshape(-1)
        _check(lib().mnv_renderer_set_model(self._h, C.byref(desc), p.ctypes.data, p.size, C.byref(grid)))

    def resize(self, width: int, height: int) -> None:
        _check(lib().mnv_renderer_resize(self._h, width, height))
        self.width, self.height = width, height

    def set_camera(self, center, back, up=(0.0, 0.0, 1.0), fx: float = -1.0, fy: float = -1.0) -> None:
        _check(lib().mnv_renderer_set_camera(self._h, fx, fy, _f3(center), _f3(back), _f3(up)))

    def set_seed(self, seed: int, accel_rebuild_after: int = -1) -> None:
        _check(lib().mnv_renderer_set_seed(self._h, seed, accel_rebuild_after))

    def render(self) -> dict:
        st = RendererStats()
        if self._tree is not None and (_harness_fused_kernel is not None or _harness_diag_used):
            a = lib().mnv_n3tree_accel(self._tree._h)
            if a:
                _fused_defaults(a)
        _check(lib().mnv_renderer_render(self._h, C.byref(st)))
        return st.as_dict()

    def download(self, want_rgba8=False):
        rgba = np.empty((self.height, self.width, 4), np.float32)
        rgba8 = np.empty((self.height, self.width, 4), np.uint8) if want_rgba8 else None
        _check(lib().mnv_renderer_download(self._h, rgba.ctypes.data, rgba8.ctypes.data if want_rgba8 else None))
        return (rgba, rgba8) if want_rgba8 else rgba

    def sync_tree(self) -> None:
        _check(lib().mnv_renderer_sync_tree(self._h))

    def set_frames_in_flight(self, count: int) -> None:
        _check(lib().mnv_renderer_set_frames_in_flight(self._h, int(count)))

    def set_guided_in_flight(self, on: bool) -> None:
        """Guided-sampling frames that change nothing rotate over the frame slots like plain frames (default off); their sample count:
        slot_guided_samples(last_slot())."""
        _check(lib().mnv_renderer_set_guided_in_flight(self._h, 1 if on else 0))

    def slot_guided_samples(self, slot: int) -> int:
        n = C.c_int64(0)
        _check(lib().mnv_renderer_slot_guided_samples(self._h, int(slot), C.byref(n)))
        return int(n.value)

    def set_fused_guided(self, enable: bool) -> None:
        _check(lib().mnv_renderer_set_fused_guided(self._h, int(enable)))

    def set_frame_inputs(self, tmax_px=None, rgba8_init=None) -> None:
        """The depth image ([height][width] float32) and the image under the volume ([height][width][4] uint8) of the reference's
        offscreen == false frames: device tensors the caller keeps alive while frames are rendered; None, None = the offline renderer."""
        px = self.width * self.height
        inputs = _frame_inputs(tmax_px, rgba8_init, px)
        self._inputs = (tmax_px, rgba8_init)  # keep the tensors alive
        _check(lib().mnv_renderer_set_frame_inputs(self._h, C.c_void_p(inputs.tmax_px if inputs is not None else None),
                                                   C.c_void_p(inputs.rgba8_init if inputs is not None else None)))

    def set_ranks(self, comm, tile_w: int = 64, tile_h: int = 24) -> None:
        """Several ranks refine one scene in lock step (VolumeRenderer::set_ranks); comm: mnv.Comm or None."""
        self._comm = comm  # keep the communicator alive as long as the renderer uses it
        _check(lib().mnv_renderer_set_ranks(self._h, C.c_void_p(comm.handle) if comm is not None else None, int(tile_w), int(tile_h)))

    def last_slot(self) -> int:
        return int(lib().mnv_renderer_last_slot(self._h))

    def download_slot(self, slot: int, want_rgba8=False):
        rgba = np.empty((self.height, self.width, 4), np.float32)
        rgba8 = np.empty((self.height, self.width, 4), np.uint8) if want_rgba8 else None
        _check(lib().mnv_renderer_download_slot(self._h, int(slot), rgba.ctypes.data, rgba8.ctypes.data if want_rgba8 else None))
        return (rgba, rgba8) if want_rgba8 else rgba


MAX_BATCH = 64


def render_voxels_accel_batch(accel: int, cams, opt: RenderOptions, tile=None, part=None, rgba=None, rgba8=None,
                              stream: int = 0) -> None:
    """Several frames (`cams`: list of Camera, same image size) in one launch; frame f is written at
    rgba[f].  `part` = (rank, world, tile_w, tile_h[, root_period]) selects the interleaved macro-tile partition."""
    n = len(cams)
    if n < 1 or n > MAX_BATCH:
        raise MnvError(MNV_E_INVALID, f"need 1 .. {MAX_BATCH} cameras, got {n}")
    arr = (CameraStruct * n)(*[c.c for c in cams])
    if tile is None:
        tile = (0, 0, cams[0].width, cams[0].height)
    p = Partition(0, 1, 0, 0) if part is None else Partition(*part)
    px = _pixels(tile, n, part)
    _check_out("rgba", rgba, px, "f32")
    _check_out("rgba8", rgba8, px, "u8")
    with _timed(stream):
        _check(lib().mnv_render_voxels_accel_batch(C.c_void_p(accel), arr, n, C.byref(opt), Rect(*tile), p, _ptr(rgba), _ptr(rgba8),
                                                   C.c_void_p(stream)))


def accel_info(accel: int, coverage: bool = False) -> dict:
    """Device bytes of the packed layout, the level of its second lookup grid, the levels below it answered by inline cell words / brick records;
    coverage=True adds mnv_accel_lookup_coverage (a pass over the grid: waits for the device)."""
    d = {"device_bytes": int(lib().mnv_accel_device_bytes(accel)), "grid2_level": int(lib().mnv_accel_grid2_level(accel)),
         "brick_levels": int(lib().mnv_accel_brick_levels(C.c_void_p(accel)))}
    if coverage:
        out = (C.c_int64 * 4)()
        _check(lib().mnv_accel_lookup_coverage(C.c_void_p(accel), out))
        d.update(nonleaf_cells=int(out[0]), inline_cells=int(out[1]), inline_lost_to_chunk_field=int(out[2]), record_chunks=int(out[3]))
    return d


def accel_set_cu_budget(accel: int, num_cus: int) -> None:
    """Compute units the tuned kernel fills (0 = all); for launches on a CU-masked stream."""
    _check(lib().mnv_accel_set_cu_budget(accel, int(num_cus)))


def stream_create_reserved(reserve_cus: int):
    """(stream handle, enabled compute units): a HIP stream that leaves `reserve_cus` units free for other streams' kernels."""
    h, n = C.c_void_p(), C.c_int32()
    _check(lib().mnv_stream_create_reserved(int(reserve_cus), C.byref(h), C.byref(n)))
    return h.value, n.value


def stream_destroy(stream: int) -> None:
    _check(lib().mnv_stream_destroy(C.c_void_p(stream)))


COMM_ID_BYTES = 128


def comm_get_unique_id() -> bytes:
    """The 128-byte RCCL id one rank draws and hands to the others (any side channel)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(lib().mnv_comm_get_unique_id(buf))
    return buf.raw


class Comm:
    """RCCL communicator of the tile gather (mnv_comm_*): one process per GPU, bound to the current device."""

    def __init__(self, unique_id: bytes, world: int, rank: int):
        if len(unique_id) != COMM_ID_BYTES:
            raise MnvError(MNV_E_INVALID, "unique id must be 128 bytes")
        h = C.c_void_p()
        _check(lib().mnv_comm_init_rank(C.create_string_buffer(unique_id, COMM_ID_BYTES), int(world), int(rank), C.byref(h)))
        self.handle, self.world, self.rank = h.value, int(world), int(rank)

    def gather_tiles(self, local, gathered, root: int = 0, stream: int = 0) -> None:
        """local: contiguous device tensor; gathered (root only): contiguous device tensor [world, *local.shape]."""
        nbytes = local.numel() * local.element_size()
        if not local.is_contiguous() or not local.is_cuda:
            raise MnvError(MNV_E_INVALID, "local must be a contiguous device tensor")
        if self.rank == root:
            if gathered is None or not gathered.is_contiguous() or not gathered.is_cuda or gathered.numel() * gathered.element_size() != nbytes * self.world:
                raise MnvError(MNV_E_INVALID, "gathered must be a contiguous device tensor of world x local bytes")
        _check(lib().mnv_gather_tiles(C.c_void_p(self.handle), _ptr(local), _ptr(gathered) if gathered is not None else None, nbytes, int(root),
                                      C.c_void_p(stream)))

    def allgather(self, table, stream: int = 0) -> None:
        """table: contiguous device tensor [world, ...]; this rank has written table[rank]; afterwards every rank holds every block."""
        if not table.is_contiguous() or not table.is_cuda or table.shape[0] != self.world:
            raise MnvError(MNV_E_INVALID, "table must be a contiguous device tensor [world, ...]")
        _check(lib().mnv_allgather(C.c_void_p(self.handle), _ptr(table), table.numel() * table.element_size() // self.world, C.c_void_p(stream)))

    def close(self) -> None:
        if getattr(self, "handle", None):
            lib().mnv_comm_destroy(C.c_void_p(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_visit_marks(table, visited, stream: int = 0) -> None:
    """visited[c] = max over ranks of table[r][c] (int32 device tensors [world, capacity] and [capacity])."""
    _check(lib().mnv_merge_visit_marks(_ptr(table), int(table.shape[0]), int(table.shape[1]), _ptr(visited), C.c_void_p(stream)))


def rccl_version() -> int:
    return int(lib().mnv_comm_rccl_version())


# ---- device times of the launches made through this binding: HIP events on the launch stream, recorded HERE (the library records nothing itself)
class _Timing:
    on = False
    pairs: list = []
    hip = None


def _hip():
    if _Timing.hip is None:
        lib()
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        h = C.CDLL(path or "libamdhip64.so")
        for name, args in (("hipEventCreate", [C.POINTER(C.c_void_p)]), ("hipEventRecord", [C.c_void_p, C.c_void_p]), ("hipEventSynchronize", [C.c_void_p]),
                           ("hipEventElapsedTime", [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]), ("hipEventDestroy", [C.c_void_p])):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = C.c_int, args
        _Timing.hip = h
    return _Timing.hip


class _timed:
    """with _timed(stream): <one launch> -- two events around it when set_timing(True) is active"""

    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        self.e0 = None
        if _Timing.on:
            h = _hip()
            self.e0, self.e1 = C.c_void_p(), C.c_void_p()
            if h.hipEventCreate(C.byref(self.e0)) or h.hipEventCreate(C.byref(self.e1)):
                raise MnvError(MNV_E_FAULT, "hipEventCreate failed")
            h.hipEventRecord(self.e0, C.c_void_p(self.stream))
        return self

    def __exit__(self, exc_type, exc, tb):
        if self.e0 is not None:
            _hip().hipEventRecord(self.e1, C.c_void_p(self.stream))
            _Timing.pairs.append((self.e0, self.e1))
        return False


def set_timing(enable: bool) -> None:
    """Start / stop timing the march launches made through render_voxels / render_voxels_accel[_track|_part|_batch] (HIP events on their streams)."""
    h = _hip()
    for e0, e1 in _Timing.pairs:
        h.hipEventDestroy(e0)
        h.hipEventDestroy(e1)
    _Timing.pairs = []
    _Timing.on = bool(enable)


def take_timing():
    """(sum of the device times in ms, launches) since the last call; waits for the launches."""
    h = _hip()
    total, n = 0.0, 0
    for e0, e1 in _Timing.pairs:
        if h.hipEventSynchronize(e1):
            raise MnvError(MNV_E_FAULT, "hipEventSynchronize failed")
        ms = C.c_float()
        if h.hipEventElapsedTime(C.byref(ms), e0, e1):
            raise MnvError(MNV_E_FAULT, "hipEventElapsedTime failed")
        total += ms.value
        n += 1
        h.hipEventDestroy(e0)
        h.hipEventDestroy(e1)
    _Timing.pairs = []
    return total, n

"""ctypes binding of oracle/_ref/libmnv_ref_gfx950*.so: the REFERENCE's own device code
(include/cuda/rt_core.cuh) + its own N3Tree loader, built for gfx950 by oracle/Makefile.ref.
TEST INFRASTRUCTURE ONLY; needs a GPU.  See oracle/ref_driver.hip."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path(contract=False):
    return os.path.join(_HERE, "_ref", "libmnv_ref_gfx950_contract.so" if contract else "libmnv_ref_gfx950.so")


def available(contract=False):
    return os.path.exists(lib_path(contract))


_libs = {}


def lib(contract=False):
    if contract not in _libs:
        import torch  # noqa: F401  (the reference build links libtorch_hip; load torch's runtime first)
        h = C.CDLL(lib_path(contract))
        h.ref_render_npz.restype = C.c_int
        h.ref_render_npz.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.POINTER(C.c_float), C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        h.ref_render_options_size.restype = C.c_int
        _libs[contract] = h
    return _libs[contract]


def render_npz(npz_path, cam_struct, opt_struct, n_probe=4096, contract=False):
    """Full-frame float RGBA [h, w, 4] from the reference's device code, plus loader probes."""
    h = lib(contract)
    assert h.ref_render_options_size() == C.sizeof(opt_struct), "RenderOptions layout differs from the reference's"
    w, ht = cam_struct.width, cam_struct.height
    rgba = np.empty((ht, w, 4), np.float32)
    dp, cp, pp = np.zeros(n_probe, np.uint16), np.zeros(n_probe, np.int32), np.zeros(n_probe, np.int32)
    meta = (C.c_int * 5)()
    c2w = (C.c_float * 12)(*list(cam_struct.c2w))
    rc = h.ref_render_npz(os.fsencode(npz_path), w, ht, cam_struct.fx, cam_struct.fy, cam_struct.cx, cam_struct.cy, c2w,
                          C.byref(opt_struct), C.sizeof(opt_struct), rgba.ctypes.data, dp.ctypes.data, cp.ctypes.data,
                          pp.ctypes.data, n_probe, meta)
    if rc != 0:
        raise RuntimeError(f"ref_render_npz failed with {rc}")
    return dict(rgba=rgba, data_probe=dp, child_probe=cp, parent_probe=pp, meta=list(meta))


def render_onscreen_npz(npz_path, cam_struct, opt_struct, tmax_px=None, rgba8_init=None):
    """The reference's march in its live call shape (offscreen == false): per-pixel t_max [h][w] float32 and / or the image under the
    volume [h][w][4] uint8.  -> float RGBA [h][w][4]."""
    h = lib()
    h.ref_render_onscreen_npz.restype = C.c_int
    w, ht = cam_struct.width, cam_struct.height
    rgba = np.empty((ht, w, 4), np.float32)
    if tmax_px is not None:
        tmax_px = np.ascontiguousarray(tmax_px, np.float32)
    if rgba8_init is not None:
        rgba8_init = np.ascontiguousarray(rgba8_init, np.uint8)
    ptr = lambda a: C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)  # noqa: E731
    wd, hh, fx, fy, cx, cy, c2w = _cam_args(cam_struct)
    rc = h.ref_render_onscreen_npz(os.fsencode(npz_path), C.c_int(wd), C.c_int(hh), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), c2w,
                                   C.byref(opt_struct), C.c_int(C.sizeof(opt_struct)), ptr(tmax_px), ptr(rgba8_init), ptr(rgba))
    if rc != 0:
        raise RuntimeError(f"ref_render_onscreen_npz failed with {rc}")
    return rgba


def dropin_onscreen_npz(npz_path, cam_spec, opt_struct, image, depth, path=0, offscreen=False):
    """The reference's eleven-parameter render_voxels call through include/mnv_reference_binding.hpp: `image` [h][w][4] uint8 is read
    (what is under the volume) and returned overwritten, `depth` [h][w] float32 is the depth attachment."""
    h = lib()
    h.ref_dropin_onscreen_npz.restype = C.c_int
    image = np.ascontiguousarray(image, np.uint8).copy()
    depth = np.ascontiguousarray(depth, np.float32)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])  # noqa: E731
    rc = h.ref_dropin_onscreen_npz(os.fsencode(npz_path), C.c_int(cam_spec["width"]), C.c_int(cam_spec["height"]), C.c_float(cam_spec["fx"]),
                                   C.c_float(cam_spec.get("fy", -1.0)), C.c_float(cam_spec.get("cx", -1.0)), C.c_float(cam_spec.get("cy", -1.0)),
                                   f3(cam_spec["center"]), f3(cam_spec["back"]), f3(cam_spec.get("up", (0.0, 0.0, 1.0))), C.byref(opt_struct),
                                   C.c_int(C.sizeof(opt_struct)), C.c_int(path), C.c_int(1 if offscreen else 0), C.c_void_p(image.ctypes.data),
                                   C.c_void_p(depth.ctypes.data))
    if rc != 0:
        raise RuntimeError(f"ref_dropin_onscreen_npz failed with {rc}")
    return image


def render_track_npz(npz_path, cam_struct, opt_struct, capacity, sample_counts=None, track_visit=True, tmax_px=None, rgba8_init=None):
    """The reference's march with its trackers: dict(rgba [h,w,4], split [h,w,3], sample [h,w,3], visited [capacity]).  tmax_px / rgba8_init: the
    two surfaces of the render loop's call (offscreen == false, cuda_renderer.cpp:141-142)."""
    h = lib()
    w, ht = cam_struct.width, cam_struct.height
    rgba = np.empty((ht, w, 4), np.float32)
    split, sample = np.empty((ht, w, 3), np.float32), np.empty((ht, w, 3), np.float32)
    visited = np.zeros(capacity, np.int32)
    sc = None if sample_counts is None else np.ascontiguousarray(sample_counts, np.int16)
    h.ref_render_track_onscreen_npz.restype = C.c_int
    tm = None if tmax_px is None else np.ascontiguousarray(tmax_px, np.float32)
    im = None if rgba8_init is None else np.ascontiguousarray(rgba8_init, np.uint8)
    assert tm is None or tm.shape == (ht, w)
    assert im is None or im.shape == (ht, w, 4)
    wd, hh, fx, fy, cx, cy, c2w = _cam_args(cam_struct)
    rc = h.ref_render_track_onscreen_npz(os.fsencode(npz_path), C.c_int(wd), C.c_int(hh), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), c2w,
                                         C.byref(opt_struct), C.c_int(C.sizeof(opt_struct)),
                                         C.c_void_p(sc.ctypes.data) if sc is not None else C.c_void_p(0), C.c_int(int(track_visit)),
                                         C.c_void_p(tm.ctypes.data if tm is not None else 0), C.c_void_p(im.ctypes.data if im is not None else 0),
                                         C.c_void_p(rgba.ctypes.data), C.c_void_p(split.ctypes.data), C.c_void_p(sample.ctypes.data),
                                         C.c_void_p(visited.ctypes.data))
    if rc != 0:
        raise RuntimeError(f"ref_render_track_npz failed with {rc}")
    return dict(rgba=rgba, split=split, sample=sample, visited=visited)


def _cam_args(cam_struct):
    return (cam_struct.width, cam_struct.height, cam_struct.fx, cam_struct.fy, cam_struct.cx, cam_struct.cy,
            (C.c_float * 12)(*list(cam_struct.c2w)))


def get_samples_npz(npz_path, cam_struct, opt_struct, grid_struct, samples_dim, tmax_px=None, dropin=False):
    """The reference's get_samples_trace_ray (rt_core.cuh:418-576) on the device, full frame; tmax_px [h][w]: the depth attachment the
    kernel reads when offscreen == false (renderer_kernel.cu:354-357).  Also returns the two tracker arrays.  dropin: the same call served by
    libmnv.so through the sixteen-parameter binding of include/mnv_reference_binding.hpp instead of the reference's device code."""
    h = lib()
    h.ref_get_samples_onscreen_npz.restype = C.c_int
    n, mg = cam_struct.width * cam_struct.height, opt_struct.max_guided_samples
    num = np.zeros(n, np.int16)
    samples = np.empty((n, mg, samples_dim), np.float32)
    clusters = np.empty((n, mg), np.int16)
    split, sample = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
    if tmax_px is not None:
        tmax_px = np.ascontiguousarray(tmax_px, np.float32)
        assert tmax_px.shape == (cam_struct.height, cam_struct.width)
    w, ht, fx, fy, cx, cy, c2w = _cam_args(cam_struct)
    rc = h.ref_get_samples_onscreen_npz(os.fsencode(npz_path), C.c_int(w), C.c_int(ht), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), c2w,
                                        C.byref(opt_struct), C.c_int(C.sizeof(opt_struct)), (C.c_int32 * 2)(*list(grid_struct.grid_dim)),
                                        (C.c_float * 3)(*list(grid_struct.min_position)), (C.c_float * 3)(*list(grid_struct.range)),
                                        C.c_int(samples_dim), C.c_void_p(tmax_px.ctypes.data if tmax_px is not None else 0), C.c_void_p(num.ctypes.data),
                                        C.c_void_p(samples.ctypes.data), C.c_void_p(clusters.ctypes.data), C.c_void_p(split.ctypes.data),
                                        C.c_void_p(sample.ctypes.data), C.c_int(int(dropin)))
    if rc != 0:
        raise RuntimeError(f"ref_get_samples_onscreen_npz failed with {rc}")
    return dict(num_samples=num, samples=samples, cluster_indices=clusters, split=split, sample=sample)


def render_nerf_results_npz(npz_path, cam_struct, opt_struct, sample_values, z_vals, offsets):
    """The reference's composite_nerf_results (rt_core.cuh:334-416) on the device, full frame."""
    h = lib()
    h.ref_render_nerf_results_npz.restype = C.c_int
    sample_values = np.ascontiguousarray(sample_values, np.float32)
    z_vals = np.ascontiguousarray(z_vals, np.float32)
    offsets = np.ascontiguousarray(offsets, np.int64)
    rgba = np.empty((cam_struct.height, cam_struct.width, 4), np.float32)
    w, ht, fx, fy, cx, cy, c2w = _cam_args(cam_struct)
    rc = h.ref_render_nerf_results_npz(os.fsencode(npz_path), C.c_int(w), C.c_int(ht), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                                       c2w, C.byref(opt_struct), C.c_int(C.sizeof(opt_struct)), C.c_void_p(sample_values.ctypes.data),
                                       C.c_int64(sample_values.shape[0]), C.c_int(sample_values.shape[1]), C.c_void_p(z_vals.ctypes.data),
                                       C.c_void_p(offsets.ctypes.data), C.c_void_p(rgba.ctypes.data))
    if rc != 0:
        raise RuntimeError(f"ref_render_nerf_results_npz failed with {rc}")
    return rgba


def render_nerf_results_dropin_npz(npz_path, cam_struct, opt_struct, sample_values, z_vals, offsets, image=None, offscreen=True):
    """libmnv.so through the nine-parameter viewer::render_nerf_results of include/mnv_reference_binding.hpp on the reference's own N3Tree / Camera /
    tensors: the RGBA8 frame [h][w][4] its launcher writes through the image surface.  image: what the surface holds before the call."""
    h = lib()
    h.ref_render_nerf_results_dropin_npz.restype = C.c_int
    sample_values = np.ascontiguousarray(sample_values, np.float32)
    z_vals = np.ascontiguousarray(z_vals, np.float32)
    offsets = np.ascontiguousarray(offsets, np.int64)
    out = np.zeros((cam_struct.height, cam_struct.width, 4), np.uint8) if image is None else np.ascontiguousarray(image, np.uint8).copy()
    w, ht, fx, fy, cx, cy, c2w = _cam_args(cam_struct)
    rc = h.ref_render_nerf_results_dropin_npz(os.fsencode(npz_path), C.c_int(w), C.c_int(ht), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                                              c2w, C.byref(opt_struct), C.c_int(C.sizeof(opt_struct)), C.c_void_p(sample_values.ctypes.data),
                                              C.c_int64(sample_values.shape[0]), C.c_int(sample_values.shape[1]), C.c_void_p(z_vals.ctypes.data),
                                              C.c_void_p(offsets.ctypes.data), C.c_int(int(offscreen)), C.c_void_p(out.ctypes.data))
    if rc != 0:
        raise RuntimeError(f"ref_render_nerf_results_dropin_npz failed with {rc}")
    return out


def _grid_args(grid_struct):
    return ((C.c_int32 * 2)(*list(grid_struct.grid_dim)), (C.c_float * 3)(*list(grid_struct.min_position)), (C.c_float * 3)(*list(grid_struct.range)))


def add_children_npz(npz_path, opt_struct, max_capacity, parent_nodes, samples, visited, grid_struct, dropin=False):
    """The reference's add_children_and_generate_samples_kernel.  samples [n*8][spc][dim] uniform numbers in; returns
    dict(samples, clusters, visited, child [max_capacity][8], parent [max_capacity]).  dropin: the same call served by libmnv.so through the
    nine-parameter viewer::add_children_and_generate_samples of include/mnv_reference_binding.hpp on the reference's own N3Tree and tensors."""
    h = lib()
    h.ref_add_children_dropin_npz.restype = C.c_int
    parent_nodes = np.ascontiguousarray(parent_nodes, np.int32)
    samples = np.ascontiguousarray(samples, np.float32).copy()
    visited = np.ascontiguousarray(visited, np.int32).copy()
    n = parent_nodes.shape[0]
    clusters = np.empty(samples.shape[:2], np.int16)
    child, parent = np.empty((max_capacity, 8), np.int32), np.empty(max_capacity, np.int32)
    gd, mp, rg = _grid_args(grid_struct)
    rc = h.ref_add_children_dropin_npz(os.fsencode(npz_path), C.byref(opt_struct), C.c_int(C.sizeof(opt_struct)), C.c_int(max_capacity),
                                       C.c_void_p(parent_nodes.ctypes.data), C.c_int(n), C.c_void_p(samples.ctypes.data), C.c_int(samples.shape[-1]),
                                       C.c_void_p(clusters.ctypes.data), C.c_void_p(visited.ctypes.data), gd, mp, rg, C.c_void_p(child.ctypes.data),
                                       C.c_void_p(parent.ctypes.data), C.c_int(int(dropin)))
    if rc != 0:
        raise RuntimeError(f"ref_add_children_npz failed with {rc}")
    return dict(samples=samples, clusters=clusters, visited=visited, child=child, parent=parent)


def generate_samples_npz(npz_path, opt_struct, nodes, samples, grid_struct, dropin=False):
    """The reference's generate_samples_kernel; dropin: libmnv.so through the eight-parameter viewer::generate_samples binding."""
    h = lib()
    h.ref_generate_samples_dropin_npz.restype = C.c_int
    nodes = np.ascontiguousarray(nodes, np.int32)
    samples = np.ascontiguousarray(samples, np.float32).copy()
    clusters = np.empty(samples.shape[:2], np.int16)
    gd, mp, rg = _grid_args(grid_struct)
    rc = h.ref_generate_samples_dropin_npz(os.fsencode(npz_path), C.byref(opt_struct), C.c_int(C.sizeof(opt_struct)), C.c_void_p(nodes.ctypes.data),
                                           C.c_int(nodes.shape[0]), C.c_void_p(samples.ctypes.data), C.c_int(samples.shape[-1]),
                                           C.c_void_p(clusters.ctypes.data), gd, mp, rg, C.c_int(int(dropin)))
    if rc != 0:
        raise RuntimeError(f"ref_generate_samples_npz failed with {rc}")
    return dict(samples=samples, clusters=clusters)


def adjust_parents_npz(npz_path, capacity, first_shift_index, to_delete, index_shifts, dropin=False):
    """The reference's adjust_parents_and_children_kernel; dropin: libmnv.so through the four-parameter viewer::adjust_parents_and_children binding."""
    h = lib()
    h.ref_adjust_parents_dropin_npz.restype = C.c_int
    to_delete = np.ascontiguousarray(to_delete, np.uint8)
    index_shifts = np.ascontiguousarray(index_shifts, np.int32)
    child, parent = np.empty((capacity, 8), np.int32), np.empty(capacity, np.int32)
    rc = h.ref_adjust_parents_dropin_npz(os.fsencode(npz_path), C.c_int(first_shift_index), C.c_void_p(to_delete.ctypes.data),
                                         C.c_void_p(index_shifts.ctypes.data), C.c_void_p(child.ctypes.data), C.c_void_p(parent.ctypes.data),
                                         C.c_int(int(dropin)))
    if rc != 0:
        raise RuntimeError(f"ref_adjust_parents_npz failed with {rc}")
    return dict(child=child, parent=parent)


def camera_pose(width, height, fx, fy, cx, cy, center, back, up, updates=1):
    """The reference's Camera ctor + _update (src/camera.cpp:29-82, glm): -> (c2w float32 [12], (fx, fy, cx, cy))."""
    h = lib()
    h.ref_camera_pose.restype = C.c_int
    out, intr = (C.c_float * 12)(), (C.c_float * 4)()
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])  # noqa: E731
    rc = h.ref_camera_pose(C.c_int(width), C.c_int(height), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), f3(center), f3(back),
                           f3(up), C.c_int(updates), out, intr)
    if rc != 0:
        raise RuntimeError(f"ref_camera_pose failed with {rc}")
    return np.array(list(out), np.float32), tuple(float(x) for x in intr)


def camera_drag(width, height, fx, center, back, up, origin, movement_speed, is_pan, about_origin, start_xy, end_xy):
    """The reference's Camera drag helpers: -> (center, v_back, origin, c2w[12]) after the drag and the next _update."""
    h = lib()
    h.ref_camera_drag.restype = C.c_int
    pose = (C.c_float * 9)(*[float(x) for x in list(center) + list(back) + list(origin)])
    out = (C.c_float * 12)()
    rc = h.ref_camera_drag(C.c_int(width), C.c_int(height), C.c_float(fx), (C.c_float * 3)(*[float(x) for x in up]), C.c_float(movement_speed),
                           C.c_int(int(is_pan)), C.c_int(int(about_origin)), C.c_float(start_xy[0]), C.c_float(start_xy[1]), C.c_float(end_xy[0]),
                           C.c_float(end_xy[1]), pose, out)
    if rc != 0:
        raise RuntimeError(f"ref_camera_drag failed with {rc}")
    p = np.array(list(pose), np.float32)
    return p[0:3], p[3:6], p[6:9], np.array(list(out), np.float32)


def dropin_render_npz(npz_path, cam_spec, opt_struct, capacity, path=0, want_trackers=False):
    """The reference's loader / N3Tree / Camera feeding libmnv.so through include/mnv_reference_binding.hpp.
    cam_spec: dict(width, height, fx, fy, cx, cy, center, back, up).  path 0 = mnv_render_voxels, 1 = packed accel."""
    h = lib()
    h.ref_dropin_render_npz.restype = C.c_int
    w, ht = cam_spec["width"], cam_spec["height"]
    rgba, rgba8 = np.empty((ht, w, 4), np.float32), np.empty((ht, w, 4), np.uint8)
    split = np.empty((ht, w, 3), np.float32) if want_trackers else None
    sample = np.empty((ht, w, 3), np.float32) if want_trackers else None
    visited = np.zeros(capacity, np.int32) if want_trackers else None
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])  # noqa: E731
    ptr = lambda a: C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)  # noqa: E731
    rc = h.ref_dropin_render_npz(os.fsencode(npz_path), C.c_int(w), C.c_int(ht), C.c_float(cam_spec["fx"]), C.c_float(cam_spec.get("fy", -1.0)),
                                 C.c_float(cam_spec.get("cx", -1.0)), C.c_float(cam_spec.get("cy", -1.0)), f3(cam_spec["center"]), f3(cam_spec["back"]),
                                 f3(cam_spec.get("up", (0.0, 0.0, 1.0))), C.byref(opt_struct), C.c_int(C.sizeof(opt_struct)), C.c_int(path), ptr(rgba),
                                 ptr(rgba8), ptr(split), ptr(sample), ptr(visited))
    if rc != 0:
        raise RuntimeError(f"ref_dropin_render_npz failed with {rc}")
    return dict(rgba=rgba, rgba8=rgba8, split=split, sample=sample, visited=visited)

"""ctypes binding of the CPU parity oracle (oracle/libmnv_oracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never from the product package.  See mnv_oracle.h for what it restates."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmnv_oracle.so")


class OrcTree(C.Structure):
    _fields_ = [("data", C.c_void_p), ("child", C.c_void_p), ("sample_counts", C.c_void_p),
                ("offset", C.c_float * 3), ("scale", C.c_float * 3), ("N", C.c_int32), ("data_dim", C.c_int32),
                ("format", C.c_int32), ("basis_dim", C.c_int32), ("capacity", C.c_int32)]


class OrcCamera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("c2w", C.c_float * 12)]


class OrcOptions(C.Structure):
    _fields_ = [("step_size", C.c_float), ("sigma_thresh", C.c_float), ("stop_thresh", C.c_float),
                ("background_brightness", C.c_float), ("render_bbox", C.c_float * 6), ("basis_minmax", C.c_int32 * 2),
                ("rot_dirs", C.c_float * 3), ("show_grid", C.c_bool), ("grid_max_depth", C.c_int32),
                ("render_depth", C.c_bool), ("use_splitting", C.c_bool), ("use_guided_sampling", C.c_bool),
                ("max_depth", C.c_int32), ("samples_per_corner", C.c_int32), ("split_batch_size", C.c_int32),
                ("nerf_batch_size", C.c_int32), ("max_sample_count", C.c_int32), ("need_viewdir", C.c_bool),
                ("appearance_embedding", C.c_int32), ("max_guided_samples", C.c_int32)]


class OrcCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "rays_in_bbox", "rays_hit", "steps", "levels", "hits", "early_stops", "max_steps")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class OrcClusterGrid(C.Structure):
    _fields_ = [("grid_dim", C.c_int32 * 2), ("min_position", C.c_float * 3), ("range", C.c_float * 3)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} missing: run `make -C oracle` (or __graft_entry__.build())")
        h = C.CDLL(LIB_PATH)
        h.orc_expf.restype = C.c_float
        h.orc_expf.argtypes = [C.c_float]
        h.orc_half_to_float.restype = C.c_float
        h.orc_half_to_float.argtypes = [C.c_uint16]
        h.orc_float_to_half.restype = C.c_uint16
        h.orc_float_to_half.argtypes = [C.c_float]
        h.orc_sh_basis.restype = None
        h.orc_sh_basis.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        h.orc_camera_pose.restype = None
        h.orc_camera_pose.argtypes = [C.POINTER(C.c_float)] * 4
        h.orc_default_options.restype = None
        h.orc_default_options.argtypes = [C.POINTER(OrcOptions)]
        h.orc_algorithmic_bytes.restype = C.c_uint64
        h.orc_algorithmic_bytes.argtypes = [C.POINTER(OrcCounters), C.c_int32, C.c_int32]
        h.orc_num_threads.restype = C.c_int
        h.orc_render_voxels.restype = C.c_int
        h.orc_render_voxels.argtypes = [C.POINTER(OrcTree), C.POINTER(OrcCamera), C.POINTER(OrcOptions),
                                        C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.POINTER(OrcCounters), C.c_int]
        h.orc_render_voxels_ex.restype = C.c_int
        h.orc_render_voxels_ex.argtypes = [C.POINTER(OrcTree), C.POINTER(OrcCamera), C.POINTER(OrcOptions),
                                           C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p, C.POINTER(OrcCounters), C.c_int]
        h.orc_tree_copy_first_touch.restype = C.c_int
        h.orc_tree_copy_first_touch.argtypes = [C.POINTER(OrcTree), C.POINTER(OrcTree), C.c_int]
        h.orc_tree_free_copy.restype = None
        h.orc_tree_free_copy.argtypes = [C.POINTER(OrcTree)]
        h.orc_get_samples_from_voxels.restype = C.c_int
        h.orc_get_samples_from_voxels.argtypes = [C.POINTER(OrcTree), C.POINTER(OrcCamera), C.POINTER(OrcOptions), C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                                  C.POINTER(OrcClusterGrid), C.c_int]
        h.orc_render_nerf_results.restype = C.c_int
        h.orc_render_nerf_results.argtypes = [C.POINTER(OrcTree), C.POINTER(OrcCamera), C.POINTER(OrcOptions), C.c_void_p, C.c_int32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        h.orc_add_children_and_generate_samples.restype = C.c_int
        h.orc_generate_samples.restype = C.c_int
        h.orc_adjust_parents_and_children.restype = C.c_int
        _lib = h
    return _lib


def _copy_struct(dst, src):
    """Field-wise copy between layout-compatible ctypes structs (product struct -> oracle struct)."""
    for name, _ in dst._fields_:
        setattr(dst, name, getattr(src, name))
    return dst


def tree_from_view(view, sample_counts=None) -> OrcTree:
    """Build an OrcTree from a *host* mnv TreeView (pointers are borrowed)."""
    t = OrcTree()
    t.data, t.child = view.data, view.child
    t.sample_counts = sample_counts.ctypes.data if sample_counts is not None else None
    for i in range(3):
        t.offset[i], t.scale[i] = view.offset[i], view.scale[i]
    t.N, t.data_dim, t.format, t.basis_dim, t.capacity = view.N, view.data_dim, view.format, view.basis_dim, view.capacity
    return t


_default_threads = None


def _threads(n_threads):
    """n_threads <= 0: every hardware thread this process can really run -- under a cgroup CPU quota not the visible ones (256 threads
    under a quota of 16 CPUs are throttled in turn and finish later than 16)."""
    global _default_threads
    if n_threads and n_threads > 0:
        return n_threads
    if _default_threads is None:
        import math
        import os

        n = min(os.cpu_count() or 1, _usable_cpus())
        q = cpu_quota()
        if q is not None:
            n = min(n, max(1, int(math.ceil(q))))
        _default_threads = max(1, n)
    return _default_threads


def copy_first_touch(tree: OrcTree, n_threads=0) -> OrcTree:
    """A copy of the tree whose pages the marching threads touched first (NUMA spread on a big host); release with free_copy."""
    out = OrcTree()
    if lib().orc_tree_copy_first_touch(C.byref(tree), C.byref(out), _threads(n_threads)) != 0:
        raise MemoryError("orc_tree_copy_first_touch")
    return out


def free_copy(tree: OrcTree) -> None:
    lib().orc_tree_free_copy(C.byref(tree))


def _usable_cpus():
    """CPUs in the process's scheduler affinity.  A host program that pins OpenMP threads (OMP_PROC_BIND) must count them BEFORE an OpenMP
    runtime loads -- afterwards the main thread is bound to one place and its own mask says 1 or 2 -- and hand the count over in
    MNV_ORACLE_CPUS (bench.py, tools/cpu_ladder.py)."""
    import os

    try:
        return max(1, int(os.environ["MNV_ORACLE_CPUS"]))
    except (KeyError, ValueError):
        pass
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def cpu_quota():
    """CPUs' worth of time this process may use: the cgroup CPU quota (v2 cpu.max, v1 cfs_quota_us / cfs_period_us) or None when there
    is none.  A container with 256 visible hardware threads and a quota of 16 CPUs runs 128 threads no faster than 16 -- slower: they
    are throttled in turn."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def baseline_threads():
    """(threads for the CPU baseline, physical cores, hardware threads, quota): one thread per physical core this process can actually run --
    the smaller of the physical cores, the scheduler affinity and the cgroup CPU quota."""
    import math
    import os

    phys, hw = physical_cores()
    n = min(phys or hw, _usable_cpus())
    quota = cpu_quota()
    if quota is not None:
        n = min(n, max(1, int(math.floor(quota + 1e-9))))
    return max(1, n), phys, hw, quota


def physical_cores():
    """(physical cores, hardware threads) of this host from the sysfs topology (distinct sibling sets); (None, threads) when unreadable."""
    import glob
    import os

    threads = os.cpu_count() or 1
    try:
        groups = set()
        for f in glob.glob("/sys/devices/system/cpu/cpu[0-9]*/topology/thread_siblings_list"):
            groups.add(open(f).read().strip())
        return (len(groups) or None), threads
    except OSError:
        return None, threads


def render(tree: OrcTree, cam_struct, opt_struct, tile=None, *, want_rgba8=False, want_trackers=False,
           want_steps=False, visited=None, track_visit=False, n_threads=0, tmax_px=None, rgba8_init=None):
    """Render a tile with the oracle.  Returns dict(rgba, rgba8, split, sample, steps, counters).  tmax_px [h][w] float32 / rgba8_init
    [h][w][4] uint8: the per-pixel inputs of the reference's offscreen == false call shape (renderer_kernel.cu:230-234,277-280)."""
    cam = _copy_struct(OrcCamera(), cam_struct)
    opt = _copy_struct(OrcOptions(), opt_struct)
    if tile is None:
        tile = (0, 0, cam.width, cam.height)
    x0, y0, w, h = tile
    rgba = np.empty((h, w, 4), np.float32)
    rgba8 = np.empty((h, w, 4), np.uint8) if want_rgba8 else None
    split = np.full((h, w, 3), -1, np.float32) if want_trackers else None
    sample = np.full((h, w, 3), -1, np.float32) if want_trackers else None
    steps = np.empty((h, w), np.int32) if want_steps else None
    ctr = OrcCounters()
    p = lambda a: a.ctypes.data if a is not None else None
    if tmax_px is not None:
        tmax_px = np.ascontiguousarray(tmax_px, np.float32)
        assert tmax_px.size == w * h
    if rgba8_init is not None:
        rgba8_init = np.ascontiguousarray(rgba8_init, np.uint8)
        assert rgba8_init.size == w * h * 4
    rc = lib().orc_render_voxels_ex(C.byref(tree), C.byref(cam), C.byref(opt), x0, y0, w, h, p(tmax_px), p(rgba8_init), p(rgba), p(rgba8),
                                    p(split), p(sample), p(visited), int(track_visit), p(steps), C.byref(ctr), _threads(n_threads))
    if rc != 0:
        raise RuntimeError("orc_render_voxels: invalid arguments")
    return dict(rgba=rgba, rgba8=rgba8, split=split, sample=sample, steps=steps, counters=ctr)


def algorithmic_bytes(ctr: OrcCounters, fmt: int, basis_dim: int) -> int:
    return int(lib().orc_algorithmic_bytes(C.byref(ctr), fmt, basis_dim))


def get_samples(tree: OrcTree, cam_struct, opt_struct, grid_struct, samples_dim, visited=None, track_visit=False, n_threads=0, tmax_px=None):
    """orc_get_samples_from_voxels(_ex) on a full frame.  Buffers are initialised the way the reference's
    host code does (num_samples = 0, samples column 0 = -1, trackers = -1).  tmax_px [h][w]: the depth attachment of offscreen == false."""
    cam = _copy_struct(OrcCamera(), cam_struct)
    opt = _copy_struct(OrcOptions(), opt_struct)
    grid = _copy_struct(OrcClusterGrid(), grid_struct)
    n, mg = cam.width * cam.height, opt.max_guided_samples
    num = np.zeros(n, np.int16)
    samples = np.full((n, mg, samples_dim), -1, np.float32)
    clusters = np.full((n, mg), -1, np.int16)
    split = np.full((n, 3), -1, np.float32)
    sample = np.full((n, 3), -1, np.float32)
    if tmax_px is not None:
        tmax_px = np.ascontiguousarray(tmax_px, np.float32)
        assert tmax_px.shape == (cam.height, cam.width)
    rc = lib().orc_get_samples_from_voxels_ex(C.byref(tree), C.byref(cam), C.byref(opt), C.c_void_p(tmax_px.ctypes.data if tmax_px is not None else 0),
                                              C.c_void_p(split.ctypes.data), C.c_void_p(sample.ctypes.data),
                                              C.c_void_p(visited.ctypes.data if visited is not None else 0), int(track_visit), C.c_void_p(num.ctypes.data),
                                              C.c_void_p(samples.ctypes.data), samples_dim, C.c_void_p(clusters.ctypes.data), C.byref(grid), _threads(n_threads))
    if rc != 0:
        raise RuntimeError("orc_get_samples_from_voxels: invalid arguments")
    return dict(num_samples=num, samples=samples, cluster_indices=clusters, split=split, sample=sample)


def render_nerf_results(tree: OrcTree, cam_struct, opt_struct, sample_values, z_vals, offsets, want_rgba8=False, n_threads=0):
    cam = _copy_struct(OrcCamera(), cam_struct)
    opt = _copy_struct(OrcOptions(), opt_struct)
    sample_values = np.ascontiguousarray(sample_values, np.float32)
    z_vals = np.ascontiguousarray(z_vals, np.float32)
    offsets = np.ascontiguousarray(offsets, np.int64)
    rgba = np.empty((cam.height, cam.width, 4), np.float32)
    rgba8 = np.empty((cam.height, cam.width, 4), np.uint8) if want_rgba8 else None
    rc = lib().orc_render_nerf_results(C.byref(tree), C.byref(cam), C.byref(opt), sample_values.ctypes.data, sample_values.shape[-1],
                                       z_vals.ctypes.data, offsets.ctypes.data, rgba.ctypes.data,
                                       rgba8.ctypes.data if want_rgba8 else None, _threads(n_threads))
    if rc != 0:
        raise RuntimeError("orc_render_nerf_results: invalid arguments")
    return dict(rgba=rgba, rgba8=rgba8)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _raw(*arrays):
    """The C side reads raw memory: every array must be C-contiguous."""
    for a in arrays:
        assert a.flags["C_CONTIGUOUS"], "pass C-contiguous arrays (np.ascontiguousarray)"


def add_children_and_generate_samples(child, parent, offset, scale, capacity, opt_struct, parent_nodes, samples, clusters, visited, grid_struct):
    """In-place on the numpy arrays (child int32 [max_cap,8], parent int32 [max_cap], samples float32, ...)."""
    opt = _copy_struct(OrcOptions(), opt_struct)
    grid = _copy_struct(OrcClusterGrid(), grid_struct)
    _raw(child, parent, parent_nodes, samples, clusters, visited)
    rc = lib().orc_add_children_and_generate_samples(C.c_void_p(child.ctypes.data), C.c_void_p(parent.ctypes.data), _f3(offset), _f3(scale),
                                                     C.c_int32(capacity), C.byref(opt), C.c_void_p(parent_nodes.ctypes.data),
                                                     C.c_int32(parent_nodes.shape[0]), C.c_void_p(samples.ctypes.data), C.c_int32(samples.shape[-1]),
                                                     C.c_void_p(clusters.ctypes.data), C.c_void_p(visited.ctypes.data), C.byref(grid))
    assert rc == 0


def generate_samples(parent, offset, scale, opt_struct, nodes, samples, clusters, grid_struct):
    opt = _copy_struct(OrcOptions(), opt_struct)
    grid = _copy_struct(OrcClusterGrid(), grid_struct)
    _raw(parent, nodes, samples, clusters)
    rc = lib().orc_generate_samples(C.c_void_p(parent.ctypes.data), _f3(offset), _f3(scale), C.byref(opt), C.c_void_p(nodes.ctypes.data),
                                    C.c_int32(nodes.shape[0]), C.c_void_p(samples.ctypes.data), C.c_int32(samples.shape[-1]),
                                    C.c_void_p(clusters.ctypes.data), C.byref(grid))
    assert rc == 0


def adjust_parents_and_children(child, parent, capacity, first_shift_index, to_delete, index_shifts):
    _raw(child, parent, to_delete, index_shifts)
    rc = lib().orc_adjust_parents_and_children(C.c_void_p(child.ctypes.data), C.c_void_p(parent.ctypes.data), C.c_int32(capacity),
                                               C.c_int32(first_shift_index), C.c_void_p(to_delete.ctypes.data), C.c_void_p(index_shifts.ctypes.data))
    assert rc == 0


class OrcMlpDesc(C.Structure):
    _fields_ = [("n_clusters", C.c_int32), ("pos_octaves", C.c_int32), ("dir_octaves", C.c_int32), ("need_viewdir", C.c_int32),
                ("n_embeddings", C.c_int32), ("embedding_dim", C.c_int32), ("hidden_width", C.c_int32), ("hidden_layers", C.c_int32),
                ("out_dim", C.c_int32), ("center", C.c_float * 3), ("inv_extent", C.c_float * 3)]


def mlp_forward(desc_struct, params, cluster_indices, samples, out_cols=None):
    """The build's own MLP on the CPU (parity unpinned, see mnv_oracle.h).  Returns float32 [n][out_cols]."""
    d = _copy_struct(OrcMlpDesc(), desc_struct)
    params = np.ascontiguousarray(params).view(np.uint16).reshape(-1)
    samples = np.ascontiguousarray(samples, np.float32)
    cluster_indices = np.ascontiguousarray(cluster_indices, np.int16)
    n = samples.shape[0]
    out_cols = d.out_dim if out_cols is None else out_cols
    out = np.zeros((n, out_cols), np.float32)
    rc = lib().orc_mlp_forward(C.byref(d), C.c_void_p(params.ctypes.data), C.c_void_p(cluster_indices.ctypes.data),
                               C.c_void_p(samples.ctypes.data), C.c_int32(samples.shape[1]), C.c_int64(n),
                               C.c_void_p(out.ctypes.data), C.c_int32(out_cols))
    assert rc == 0
    return out

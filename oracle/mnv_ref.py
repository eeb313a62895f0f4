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

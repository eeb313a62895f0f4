"""The -DMNV_TEST_HOOKS build of the library and the batch renderer (mega-nerf-viewer_amd/testhooks/, built by the same Makefile from the same
sources): the only binaries that honour MNV_RCCL_LIBRARY (a stand-in for RCCL's transport, tests/shim/fake_rccl.cpp) and
MNV_RANKS_SHARE_GPU (every rank on one device).  The rehearsals of the world > 1 paths on a one-GPU box load these; the shipped libmnv.so
and mnv_render ignore both variables (tests/test_capi_symbols.py)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOOKS_DIR = os.path.join(ROOT, "mega-nerf-viewer_amd", "testhooks")
HOOKS_LIB = os.path.join(HOOKS_DIR, "libmnv.so")
HOOKS_EXE = os.path.join(HOOKS_DIR, "mnv_render")


def hooks_env(env=None, **extra):
    """A copy of `env` (default: os.environ) in which Python's binding loads the hooks library."""
    e = dict(os.environ if env is None else env)
    e["MNV_LIB_PATH"] = HOOKS_LIB
    e.update(extra)
    return e

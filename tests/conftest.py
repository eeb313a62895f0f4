import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The live reference build (oracle/_ref: the reference's own device code for gfx950) travels with the repository.  When it is
    # there at collection, the tests that run against it must run: a library that is present but does not load, or that vanishes
    # on the way, FAILS them instead of skipping (tests/test_parity_gpu.py: require_live_reference).
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libmnv_ref_gfx950.so")):
        os.environ.setdefault("MNV_REQUIRE_LIVE_REF", "1")


def pytest_report_header(config):
    """Say which binaries the run uses: a missing live reference build or a stale libmnv.so is a reported condition, not a
    silent skip (tests that need oracle/_ref still skip individually; the committed goldens hold the pin without it)."""
    try:
        import __graft_entry__ as g

        st = g.build_state()
        return [f"mnv build state: libmnv.so: {st['libmnv']}; oracle: {st['oracle']}",
                f"mnv live reference build: {st['live_reference_build']}"]
    except Exception as e:  # the header must never break collection
        return [f"mnv build state: unavailable ({e})"]


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    ref = os.path.join(ROOT, "oracle", "_ref", "libmnv_ref_gfx950.so")
    terminalreporter.write_line(f"mnv live reference build: {'present' if os.path.exists(ref) else 'ABSENT (tests against oracle/_ref were skipped; goldens from it are committed)'}")


@pytest.fixture(scope="session")
def mnv():
    """The product binding; builds libmnv.so / the oracle on first use if they are missing."""
    import __graft_entry__ as g

    g.build_if_missing()
    import mega_nerf_viewer_amd as m

    m.lib()
    return m


@pytest.fixture(scope="session")
def orc(mnv):
    import mnv_oracle

    mnv_oracle.lib()
    return mnv_oracle


@pytest.fixture(scope="session")
def torch_gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no HIP device is visible")
    return torch


@pytest.fixture(scope="session")
def fake_rccl(tmp_path_factory):
    """tests/shim/fake_rccl.cpp built into a shared library: a host-staged stand-in for the RCCL entry points libmnv.so binds, so that the
    world > 1 paths can run with several ranks on one GPU (MNV_RCCL_LIBRARY).  Test infrastructure; RCCL's own transport is not covered."""
    import subprocess

    out = str(tmp_path_factory.mktemp("shim") / "libfake_rccl.so")
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "shim", "fake_rccl.cpp"),
           "-o", out, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out

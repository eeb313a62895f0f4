"""The .npz loader (own ZIP / deflate / .npy parser + N3Tree::open) under AddressSanitizer and UBSan on the CPU: mutated files must be
accepted or rejected with an exception, without a single sanitizer report.  (GPU sanitizers are not available on this pool.)"""
import os
import shutil
import subprocess

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "mega-nerf-viewer_amd", "host")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_npz_loader_survives_mutated_files_under_asan_ubsan(mnv, tmp_path):
    exe = str(tmp_path / "npz_fuzz")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + HOST, "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "san", "npz_fuzz.cpp"), os.path.join(ROOT, "tests", "san", "link_stubs.cpp"),
           os.path.join(HOST, "npz.cpp"), os.path.join(HOST, "n3tree.cpp"), os.path.join(HOST, "data_format.cpp"),
           "-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-lz", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    tree = cases.make_tree(mnv, cases.CASES["sh4_d6"]["tree"])
    plain = str(tmp_path / "plain.npz")
    tree.save_npz(plain)
    data, child, parent = tree.host_arrays()
    comp = str(tmp_path / "comp.npz")   # deflate members, written by numpy
    np.savez_compressed(comp, data=data.reshape(data.shape[0], 2, 2, 2, -1), child=child.reshape(-1, 2, 2, 2),
                        invradius3=np.float32([0.5, 0.5, 0.5]), offset=np.float32([0.5, 0.5, 0.5]), data_format=np.array("SH4"))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for src, n in ((plain, 600), (comp, 300)):
        r = subprocess.run([exe, src, str(n), str(tmp_path / "mut.npz")], capture_output=True, text=True, timeout=600, env=env)
        out = r.stdout + r.stderr
        assert r.returncode == 0 and "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
        assert "opened" in r.stdout

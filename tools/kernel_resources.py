"""Register / LDS / scratch usage of libmnv.so's kernels from the code-object metadata (what the hardware is told), not from
rocprofv3's kernel-trace columns (its VGPR_Count / LDS_Block_Size rows for these kernels are wrong on this stack: 32 / 0).

usage: kernel_resources.py [name-substring ...]     (default: march_accel_kernel<9,256,0> and friends)
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib=os.path.join(ROOT, "mega-nerf-viewer_amd", "libmnv.so")):
    out = []
    with tempfile.TemporaryDirectory() as d:
        so = os.path.join(d, "libmnv.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for co in sorted(glob.glob(so + ".*gfx950*")):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                blk = ".agpr_count:" + blk
                f = dict(re.findall(r"\.(\w+):\s+('?[^\n]+)", blk))
                name = f.get("name", "").strip("'")
                filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt", path=LLVM)
                demangled = (subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() if filt else "") or name
                out.append({"kernel": demangled, "vgpr": int(f.get("vgpr_count", -1)), "agpr": int(f.get("agpr_count", -1)), "sgpr": int(f.get("sgpr_count", -1)),
                            "vgpr_spills": int(f.get("vgpr_spill_count", 0)), "sgpr_spills": int(f.get("sgpr_spill_count", 0)),
                            "lds_static_bytes": int(f.get("group_segment_fixed_size", 0)), "scratch_bytes": int(f.get("private_segment_fixed_size", 0)),
                            "max_workgroup": int(f.get("max_flat_workgroup_size", 0))})
    return out


if __name__ == "__main__":
    keys = sys.argv[1:] or ["march_accel_kernel<9, 256, 0>", "march_accel_kernel<9, 256, 2>", "march_accel_kernel<9, 256, 3>", "march_ref_layout_kernel<9>", "mlp_forward"]
    for k in kernels():
        if any(s in k["kernel"] for s in keys):
            print(k)

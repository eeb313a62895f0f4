"""Golden camera matrices from the reference's own Camera (src/camera.cpp:29-82, glm), built into oracle/_ref.
Run on the GPU box (the reference Camera uploads its matrix in the constructor):
    python tests/golden/make_camera_goldens.py gpurun_out/goldens
then copy gpurun_out/goldens/ref_camera_pose.npz into tests/golden/."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402,F401

import mega_nerf_viewer_amd as mnv  # noqa: E402
import mnv_ref  # noqa: E402


def poses():
    rng = np.random.default_rng(77)
    out = [dict(width=256, height=256, fx=1111.0, fy=-1.0, cx=-1.0, cy=-1.0, center=(-3.55, 0.0, 3.55), back=(-0.7071068, 0.0, 0.7071068), up=(0.0, 0.0, 1.0)),
           dict(width=1921, height=1081, fx=1600.0, fy=1500.0, cx=-1.0, cy=-1.0, center=(2.4, 0.1, 0.9), back=(0.94, 0.0, 0.34), up=(0.0, 0.0, 1.0)),
           dict(width=800, height=600, fx=700.0, fy=-1.0, cx=410.5, cy=290.25, center=(2.2, 5.5, 0.0), back=(0.4, 0.9, 0.1), up=(1.0, 0.0, 0.0))]
    for _ in range(40):
        c = rng.normal(0, 3, 3)
        b = rng.normal(0, 1, 3) * rng.uniform(0.1, 10)   # not normalised: _update normalises
        u = rng.normal(0, 1, 3)
        out.append(dict(width=int(rng.integers(16, 4000)), height=int(rng.integers(16, 2200)), fx=float(rng.uniform(100, 3000)),
                        fy=float(rng.choice([-1.0, rng.uniform(100, 3000)])), cx=float(rng.choice([-1.0, rng.uniform(0, 1000)])), cy=-1.0,
                        center=tuple(np.float32(c)), back=tuple(np.float32(b)), up=tuple(np.float32(u))))
    return out


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    P = poses()
    keys = ("width", "height", "fx", "fy", "cx", "cy")
    inputs = np.array([[p[k] for k in keys] + list(p["center"]) + list(p["back"]) + list(p["up"]) for p in P], np.float64)
    c2w = {1: [], 2: [], 3: []}
    intr = []
    n_equal = 0
    for p in P:
        for updates in (1, 2, 3):
            m, it = mnv_ref.camera_pose(p["width"], p["height"], p["fx"], p["fy"], p["cx"], p["cy"], p["center"], p["back"], p["up"], updates)
            c2w[updates].append(m)
        intr.append(it)
        cam = mnv.Camera(p["width"], p["height"], p["fx"], p["fy"], p["cx"], p["cy"]).set_pose(p["center"], p["back"], p["up"])
        n_equal += int(np.array_equal(np.float32(list(cam.c2w)).view(np.uint32), c2w[1][-1].view(np.uint32)))
    print(f"{n_equal} of {len(P)} build cameras bit-identical to the reference's after one _update")
    np.savez_compressed(os.path.join(outdir, "ref_camera_pose.npz"), inputs=inputs, c2w_1=np.array(c2w[1]), c2w_2=np.array(c2w[2]), c2w_3=np.array(c2w[3]),
                        intrinsics=np.array(intr, np.float32))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "goldens"))

"""Golden poses of the reference's Camera drag helpers (include/camera.hpp:22-25, src/camera.cpp:132-187, glm), built into oracle/_ref.
Run on the GPU box (the reference Camera uploads its matrix in the constructor):
    python tests/golden/make_camera_drag_goldens.py gpurun_out/goldens
then copy gpurun_out/goldens/ref_camera_drag.npz into tests/golden/."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402,F401

import mnv_ref  # noqa: E402


def drags():
    """width, height, fx, center[3], back[3], up[3], origin[3], movement_speed, is_pan, about_origin, x0, y0, x1, y1"""
    rng = np.random.default_rng(99)
    out = []
    for k in range(64):
        w, h = int(rng.integers(64, 3000)), int(rng.integers(64, 2000))
        center = rng.normal(0, 3, 3)
        back = rng.normal(0, 1, 3)
        up = (0.0, 0.0, 1.0) if k % 3 else tuple(rng.normal(0, 1, 3))
        origin = rng.normal(0, 1, 3) if k % 2 else np.zeros(3)
        speed = float(rng.choice([1.0, 0.25, 3.0]))
        x0, y0 = float(rng.uniform(0, w)), float(rng.uniform(0, h))
        far = k % 8 == 7   # a drag far enough to run into the pole guard or to wrap the azimuth
        x1, y1 = x0 + float(rng.normal(0, w * (4.0 if far else 0.3))), y0 + float(rng.normal(0, h * (4.0 if far else 0.3)))
        out.append([w, h, float(rng.uniform(200, 2000))] + list(np.float32(center)) + list(np.float32(back)) + list(np.float32(up)) + list(np.float32(origin)) +
                   [speed, k % 4 == 1, k % 4 >= 2, x0, y0, x1, y1])
    return np.array(out, np.float64)


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    D = drags()
    res = []
    for d in D:
        c, b, o, m = mnv_ref.camera_drag(int(d[0]), int(d[1]), np.float32(d[2]), np.float32(d[3:6]), np.float32(d[6:9]), np.float32(d[9:12]), np.float32(d[12:15]),
                                         np.float32(d[15]), bool(d[16]), bool(d[17]), (np.float32(d[18]), np.float32(d[19])), (np.float32(d[20]), np.float32(d[21])))
        res.append(np.concatenate([c, b, o, m]))
    np.savez_compressed(os.path.join(outdir, "ref_camera_drag.npz"), inputs=D, outputs=np.array(res, np.float32))
    print("wrote", len(D), "drags")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "goldens"))

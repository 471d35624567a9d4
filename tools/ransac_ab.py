"""Robust-transform outputs (M and inlier masks) of isx_frame_transform on seeded random problems, saved for a bitwise
A/B between two builds of the library:  ISX_LIB_PATH=<lib> python3 tools/ransac_ab.py out.npz [cases=200];
python3 tools/ransac_ab.py --compare a.npz b.npz"""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "image-stitcher_amd"))
import numpy as np

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        same = a[k].shape == b[k].shape and np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))
        bad += not same
        if not same:
            print("DIFFERS", k, np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max() if a[k].shape == b[k].shape else "shape")
    print(len(a.files), "arrays,", bad, "differ")
    sys.exit(1 if bad else 0)

import torch
from image_stitcher_amd import ops
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctx = ops.FrameCtx("cuda:0")
out = {}
t_total = 0.0
for case in range(N):
    rng = np.random.default_rng(4000 + case)
    persp = bool(case & 1)
    n = int(rng.integers(8, 1500))
    src = rng.uniform(0, 1900, (n, 2)).astype(np.float32)
    th, sc = np.deg2rad(rng.uniform(-5, 5)), rng.uniform(0.9, 1.1)
    H = np.array([[sc * np.cos(th), -sc * np.sin(th), rng.uniform(-300, 300)], [sc * np.sin(th), sc * np.cos(th), rng.uniform(-50, 50)],
                  [rng.uniform(-2e-5, 2e-5) if persp else 0, rng.uniform(-2e-5, 2e-5) if persp else 0, 1.0]])
    p = np.c_[src, np.ones(n)] @ H.T
    dst = (p[:, :2] / p[:, 2:3] + rng.normal(0, rng.choice([0.2, 1.0, 3.0]), (n, 2))).astype(np.float32)
    bad = rng.random(n) < rng.choice([0.0, 0.2, 0.5])
    dst[bad] = rng.uniform(0, 1900, (int(bad.sum()), 2)).astype(np.float32)
    t0 = time.perf_counter()
    M, mask = ops.frame_transform(ctx, src.reshape(-1, 1, 2), dst.reshape(-1, 1, 2), persp)
    t_total += time.perf_counter() - t0
    out[f"M{case}"] = np.zeros(1) if M is None else np.asarray(M, np.float64)
    out[f"m{case}"] = np.zeros(1, np.uint8) if mask is None else np.asarray(mask, np.uint8)
np.savez(sys.argv[1], **out)
print(f"{N} problems, {t_total / N * 1e3:.3f} ms per isx_frame_transform call")

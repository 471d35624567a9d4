"""Times the robust-transform stage: GPU pipeline (host cv::RNG sampler -> hypotheses -> scoring
-> host select -> refine) against the CPU oracle's sequential RANSAC on the same points."""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ops
from oracle import orc


def case(n, seed, frac, persp):
    rng = np.random.default_rng(seed)
    th, s = np.deg2rad(3.0), 1.02
    H = np.array([[s * np.cos(th), -s * np.sin(th), 140.0], [s * np.sin(th), s * np.cos(th), -60.0], [0, 0, 1.0]])
    if persp:
        H[2, :2] = (3e-5, -2e-5)
    src = rng.uniform(0, 1920, (n, 2)).astype(np.float32)
    q = np.c_[src.astype(np.float64), np.ones(n)] @ H.T
    dst = (q[:, :2] / q[:, 2:3] + rng.normal(0, 0.5, (n, 2))).astype(np.float32)
    k = int(frac * n)
    dst[:k] = rng.uniform(0, 1920, (k, 2)).astype(np.float32)
    p = rng.permutation(n)
    return src[p], dst[p]


def best(fn, reps=5):
    b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        b = min(b, time.perf_counter() - t0)
    return b, r


print("| model | points | outliers | iterations | GPU pipeline (wall) | of which host sampler | CPU oracle | hypotheses x points / s (score kernel) |")
print("|---|---|---|---|---|---|---|---|")
for persp in (False, True):
    for n, frac in ((150, 0.3), (1500, 0.4), (4000, 0.6)):
        src, dst = case(n, n, frac, persp)
        t_gpu, (M, mask, tr) = best(lambda: ops.estimate_transform(src, dst, persp, return_trace=True))
        t_s, _ = best(lambda: ops.ransac_sample(src, dst, persp, 2000))
        t_cpu, _ = best(lambda: (orc.find_homography(src, dst) if persp else orc.estimate_affine_partial(src, dst)), reps=3)
        models = tr["models"]; valid = torch.from_numpy(tr["valid"]).cuda()
        sd, dd = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
        thr = ops.RANSAC_DEFAULTS[persp]["thr"]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            ops.ransac_score(models, valid, sd, dd, persp, thr)
        e0.record()
        for _ in range(20):
            ops.ransac_score(models, valid, sd, dd, persp, thr)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"| {'homography' if persp else 'affine partial'} | {n} | {int(frac*100)} % | {tr['iters']} of {len(models)} | {t_gpu*1e3:.2f} ms | "
              f"{t_s*1e3:.2f} ms | {t_cpu*1e3:.2f} ms | {len(models) * n / (us * 1e-6):.3g} ({us:.1f} us) |", flush=True)

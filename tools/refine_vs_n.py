"""ransac_refine_kernel (one wavefront: inlier mask + refit + Levenberg-Marquardt) against the number of points: does its time
come from the points (work that more wavefronts could share) or from the chain of dependent steps (reductions, the one-lane
solves)?  usage: python3 tools/refine_vs_n.py"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-stitcher_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from image_stitcher_amd import ops
from ransac_bench import case

for persp in (False, True):
    for n in (40, 150, 500, 1500, 3000, 6000):
        src, dst = case(n, n, 0.3, persp)
        M, mask, tr = ops.estimate_transform(src, dst, persp, return_trace=True)
        cfg = ops.RANSAC_DEFAULTS[persp]
        sd, dd = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
        best = tr["models"][int(tr["best"])]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for rep in range(6):
            Md = best.clone()
            torch.cuda.synchronize()
            e0.record()
            ops.ransac_refine(Md, sd, dd, persp, cfg["thr"], refit=persp, max_iters=cfg["refine_iters"])
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"{'homography' if persp else 'affine    '} n={n:5d} inliers={int(np.asarray(mask).sum()):5d}  refine {min(ts):6.1f} us", flush=True)

# the homography call split into its two halves: all-inlier DLT refit (9 x 9 Jacobi) and Levenberg-Marquardt (8 parameters)
src, dst = case(500, 500, 0.3, True)
M, mask, tr = ops.estimate_transform(src, dst, True, return_trace=True)
sd, dd = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
best = tr["models"][int(tr["best"])]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for refit, iters in ((True, 10), (True, 0), (False, 10), (False, 0)):
    ts = []
    for rep in range(6):
        Md = best.clone(); torch.cuda.synchronize(); e0.record()
        _, info = ops.ransac_refine(Md, sd, dd, True, 5.0, refit=refit, max_iters=iters)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"homography n=500 refit={refit} lm_iters={iters}: {min(ts):6.1f} us  info={info.cpu().tolist()}", flush=True)

for iters in (0, 1, 2, 3, 5, 10):
    ts = []
    for rep in range(6):
        Md = best.clone(); torch.cuda.synchronize(); e0.record()
        ops.ransac_refine(Md, sd, dd, True, 5.0, refit=True, max_iters=iters)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"homography n=500 refit=True lm_iters={iters:2d}: {min(ts):6.1f} us", flush=True)

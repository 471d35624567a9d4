"""Stage-by-stage probe of the RANSAC kernels (prints after every synchronised stage)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ops

def P(*a):
    print(*a, flush=True)

rng = np.random.default_rng(0)
n = 12
src = rng.uniform(0, 1000, (n, 2)).astype(np.float32)
dst = (src * 1.01 + (5, -7) + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
for persp in (False, True):
    P("persp", persp)
    sub = ops.ransac_sample(src, dst, persp, 64)
    P(" sampled", sub.shape)
    sd, dd = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    subd = torch.from_numpy(np.ascontiguousarray(sub)).cuda()
    torch.cuda.synchronize(); P(" uploaded")
    models, valid = ops.ransac_hypotheses(sd, dd, subd, persp)
    torch.cuda.synchronize(); P(" hypotheses ok", valid.sum().item())
    counts = ops.ransac_score(models, valid, sd, dd, persp, 5.0)
    torch.cuda.synchronize(); P(" score ok", counts.max().item())
    best = int(counts.argmax().item())
    M = models[best].clone()
    torch.cuda.synchronize(); P(" clone ok")
    for refit, iters in ((0, 0), (int(persp), 0), (0, 1), (int(persp), 10)):
        M2 = M.clone()
        mask, info = ops.ransac_refine(M2, sd, dd, persp, 5.0, refit, iters)
        torch.cuda.synchronize(); P(f" refine refit={refit} iters={iters} ok", info.item(), M2.cpu().numpy()[:3])
P("all stages ok")

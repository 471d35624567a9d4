"""Where does the one-workgroup-per-pair kernel start to pay?  Random subsets of the bench set's pairs (512 images x 4096 x 128-d)
of 128 .. 4096 pairs through ops.knn2_l2_pairs and through the directed kernel: ms per call (ops.PAIR_MIN_ITEMS is set from this)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ops, pairs as prs
from image_stitcher_amd.synthetic import overlapping_crops_descriptors
kp, d = 4096, 128
des, _ = overlapping_crops_descriptors(seed=1234, n_scenes=8, imgs_per_scene=64, kp=kp, step=448, d=d)
n = des.shape[0]
src = torch.from_numpy(des.reshape(-1, d)).cuda()
starts, counts = np.arange(n) * kp, np.full((n,), kp)
P = ops.prepare_descriptor_tensor(src, starts, counts)
S = ops.sort_rows_by_norm(P)
allp = prs.unordered_pairs(n)
rng = np.random.default_rng(5)
for m in (128, 256, 384, 512, 768, 1024, 1536, 2048, 4096):
    sub = allp[rng.permutation(len(allp))[:m]]
    plan_d = ops.plan_items(prs.directed_items(sub, starts, counts), src.device)
    plan_p = ops.plan_items(prs.pair_items(sub, starts, counts), src.device)
    best = [1e9, 1e9]
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cd = ops.knn2_l2_items(P, None, 0.65, plan=plan_d)
        torch.cuda.synchronize(); best[0] = min(best[0], time.perf_counter() - t0)
        t0 = time.perf_counter()
        cp = ops.knn2_l2_pairs(S, None, 0.65, plan=plan_p)
        torch.cuda.synchronize(); best[1] = min(best[1], time.perf_counter() - t0)
    print(f"{m:5d} pairs: directed {best[0]*1e3:7.3f} ms | pair kernel {best[1]*1e3:7.3f} ms | equal {bool(torch.equal(cd.view(-1, 2), cp))}", flush=True)

"""Worst case of the pair kernel's routing: a pair list in which EVERY pair overlaps heavily (both directions overflow the candidate
lists), so the pair kernel only costs (it leaves each pair after the first pass) and the two-pass kernel does all the counting.
64 scenes x 8 images x 4096 x 128-d, same-scene pairs only.  usage: python3 tools/pairs_all_overlap.py"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ops, pairs as prs
from image_stitcher_amd.synthetic import overlapping_crops_descriptors
kp, d = 4096, 128
des, scene_of = overlapping_crops_descriptors(seed=7, n_scenes=64, imgs_per_scene=8, kp=kp, step=256, d=d)
n = des.shape[0]
src = torch.from_numpy(des.reshape(-1, d)).cuda()
starts, counts = np.arange(n) * kp, np.full((n,), kp)
allp = prs.unordered_pairs(n)
sub = allp[scene_of[allp[:, 0]] == scene_of[allp[:, 1]]]
P = ops.prepare_descriptor_tensor(src, starts, counts)
S = ops.sort_rows_by_norm(P)
plan_d = ops.plan_items(prs.directed_items(sub, starts, counts), src.device)
plan_p = ops.plan_items(prs.pair_items(sub, starts, counts), src.device)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cd = ops.knn2_l2_items(P, None, 0.65, plan=plan_d)
    torch.cuda.synchronize(); td = time.perf_counter() - t0
    st = {}
    t0 = time.perf_counter()
    cp = ops.knn2_l2_pairs(S, None, 0.65, plan=plan_p, stats=st)
    torch.cuda.synchronize(); tp = time.perf_counter() - t0
    print(f"{len(sub)} overlapping pairs: directed {td*1e3:.2f} ms | pair kernel + second stage {tp*1e3:.2f} ms ({st['overflow_directions']} of {2*len(sub)} directions "
          f"handed on) | equal {bool(torch.equal(cd.view(-1, 2), cp))} | min count {int(cd.min())}", flush=True)

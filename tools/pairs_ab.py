"""A/B of the both-directions pair kernel (isx_knn2_l2_count_pairs + overflow stage) against the two-pass directed kernel on
an all-pairs set: equal counts, ms per pass.  usage: python3 tools/pairs_ab.py [n_images=128] [sort=1] [kp=4096] [d=128]   (kp > 4096: the kernel's two-group table; d = 64 or 61: 64-byte rows)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ops, pairs as prs
from image_stitcher_amd.synthetic import overlapping_crops_descriptors

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 128
do_sort = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kp, d = (int(sys.argv[3]) if len(sys.argv) > 3 else 4096), (int(sys.argv[4]) if len(sys.argv) > 4 else 128)
des, _ = overlapping_crops_descriptors(seed=1234, n_scenes=max(1, n_img // 64), imgs_per_scene=min(64, n_img), kp=kp, step=kp * 448 // 4096, d=d)
n_img = des.shape[0]
src = torch.from_numpy(des.reshape(-1, d)).cuda()
starts, counts = np.arange(n_img) * kp, np.full((n_img,), kp)
all_pairs = prs.unordered_pairs(n_img)
P = ops.prepare_descriptor_tensor(src, starts, counts)
plan_d = ops.plan_items(prs.directed_items(all_pairs, starts, counts), src.device)
plan_p = ops.plan_items(prs.pair_items(all_pairs, starts, counts), src.device)
ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c_d = ops.knn2_l2_items(P, None, 0.65, plan=plan_d, events=ev)
    torch.cuda.synchronize(); t_d = time.perf_counter() - t0
    k_d = ev[0].elapsed_time(ev[1])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    S = ops.sort_rows_by_norm(P) if do_sort else P
    torch.cuda.synchronize(); t_s = time.perf_counter() - t0
    st = {}
    t0 = time.perf_counter()
    c_p = ops.knn2_l2_pairs(S, None, 0.65, plan=plan_p, events=ev, stats=st)
    torch.cuda.synchronize(); t_p = time.perf_counter() - t0
    k_p = ev[0].elapsed_time(ev[1])
    same = bool(torch.equal(c_d.view(-1, 2), c_p))
    print(f"pairs={len(all_pairs)} directed {t_d*1e3:.1f} ms (kernel {k_d:.1f}) | pairs kernel {t_p*1e3:.1f} ms (events {k_p:.1f}, sort {t_s*1e3:.2f} ms, "
          f"overflow directions {st['overflow_directions']} of {2*len(all_pairs)}) | equal counts: {same} | speedup {t_d/(t_p+t_s):.2f}x", flush=True)

"""A/B of the count kernels (flags 2 = 32x32x32 MFMA shape, flags 4 = 16x16x64 with 4 column blocks per wave, flags 0 = the default: 16x16x64 with 8) on an all-pairs SIFT-like set: equal counts, ms per launch.
usage: python3 tools/count16_ab.py [n_img=128]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ops
from image_stitcher_amd.synthetic import overlapping_crops_descriptors

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 128
kp, d = 4096, 128
dev = torch.device("cuda", 0)
des, _ = overlapping_crops_descriptors(seed=1234, n_scenes=2, imgs_per_scene=n_img // 2, kp=kp, step=448, d=d)
starts = np.arange(n_img) * kp
items = np.array([(starts[i], kp, starts[j], kp) for i in range(n_img) for j in range(n_img) if i != j], np.int32)
P = ops.prepare_descriptor_tensor(torch.from_numpy(des.reshape(-1, d)).to(dev), starts, [kp] * n_img)
plan = ops.plan_items(items, dev)
ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
res = {}
for rep in range(3):
    for flags in (2, 4, 0):
        c = ops.knn2_l2_items(P, None, 0.65, plan=plan, events=ev, verify_all=flags)
        torch.cuda.synchronize()
        res.setdefault(flags, []).append(ev[0].elapsed_time(ev[1]))
        res[("c", flags)] = c
print("counts equal:", bool(torch.equal(res[("c", 0)], res[("c", 2)])), bool(torch.equal(res[("c", 0)], res[("c", 4)])), "sum", int(res[("c", 0)].sum()))
for flags in (2, 4, 0):
    print(f"flags={flags}: " + " ".join(f"{t:.2f}" for t in res[flags]) + " ms", flush=True)
# ragged sets with ties, every query also through the exact scan (flags 3)
rng = np.random.default_rng(3)
lst = [rng.integers(0, 40, (int(r), 128), dtype=np.uint8) for r in rng.integers(1, 700, 12)]
lst[3][5] = lst[3][9]
Pr = ops.prepare_descriptors(lst)
it = np.array([(Pr.starts[i], Pr.counts[i], Pr.starts[j], Pr.counts[j]) for i in range(12) for j in range(12) if i != j], np.int32)
c0 = ops.knn2_l2_items(Pr, it, 0.8, verify_all=0)
for f in (1, 2, 3, 4, 5):
    print("ragged flags", f, "equal:", bool(torch.equal(c0, ops.knn2_l2_items(Pr, it, 0.8, verify_all=f))))

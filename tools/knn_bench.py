"""Scratch micro-benchmark of the kNN-2 kernel (not the driver's bench.py)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ops
from image_stitcher_amd.synthetic import overlapping_crops_descriptors

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 64
kp = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
des, _ = overlapping_crops_descriptors(seed=1, n_scenes=2, imgs_per_scene=n_img // 2, kp=kp, step=448, d=d)
src = torch.from_numpy(des.reshape(-1, d)).cuda()
starts = np.arange(n_img) * kp
P = ops.prepare_descriptor_tensor(src, starts, [kp] * n_img)
items = np.array([(starts[i], kp, starts[j], kp) for i in range(n_img) for j in range(n_img) if i != j], np.int32)
for xo in (True, False):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c = ops.knn2_l2_items(P, items, 0.65, xcd_order=xo)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        upairs = len(items) / 2
        print(f"xcd_order={xo} items={len(items)} {dt*1e3:.1f} ms  {dt/upairs*1e6:.2f} us/unordered pair  "
              f"{upairs/dt:.0f} pairs/s  int8 {2*kp*kp*d*upairs/dt/1e12:.1f} TOP/s algorithmic "
              f"(x2 executed), good>=5: {(c>=5).sum().item()}", flush=True)

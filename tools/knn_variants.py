"""Diagnostic: time the count kernel of the library named by ISX_LIB_PATH (default: the product build).
Diagnostic builds (tools/build_diag.sh) give wrong counts by design."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ops, _lib
from image_stitcher_amd.synthetic import overlapping_crops_descriptors
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 128
flag_list = [0]
kp, d = 4096, 128
des, _ = overlapping_crops_descriptors(seed=1, n_scenes=2, imgs_per_scene=n_img // 2, kp=kp, step=448, d=d)
src = torch.from_numpy(des.reshape(-1, d)).cuda()
starts = np.arange(n_img) * kp
P = ops.prepare_descriptor_tensor(src, starts, [kp] * n_img)
items = np.array([(starts[i], kp, starts[j], kp) for i in range(n_img) for j in range(n_img) if i != j], np.int32)
run_items, perm, items_dev, inv_dev, inv_np = ops.plan_items(items, src.device)
m = len(run_items)
L = _lib.lib()
for fl in flag_list:
    for rep in range(3):
        counts = torch.zeros((m,), dtype=torch.int32, device=src.device)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ops.check(L.isx_knn2_l2_count_items(ops._ptr(P.desc), ops._ptr(P.nrm), ops._ptr(P.cin), P.d_pad, ops._ptr(items_dev), m, kp, 0.65,
                                            ops._ptr(counts), fl, ops._stream()))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    up = len(items) / 2
    print(f"lib={os.environ.get('ISX_LIB_PATH', 'product')[-24:]} {dt*1e3:.1f} ms {dt/up*1e6:.2f} us/pair {up/dt:.0f} pairs/s sum={int(counts.sum())}", flush=True)

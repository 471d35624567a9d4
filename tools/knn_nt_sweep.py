"""Diagnostic: per-MFMA time of the count kernel as a function of the train-set length of one work item
(fixed total work).  Separates per-workgroup overhead (prologue, epilogue, dispatch) from the steady-state loop."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ops, _lib

kp, d, n_img = 4096, 128, 128
rng = np.random.default_rng(0)
src = torch.from_numpy(rng.integers(0, 120, (n_img * kp, d), dtype=np.uint8)).cuda()
starts = np.arange(n_img) * kp
P = ops.prepare_descriptor_tensor(src, starts, [kp] * n_img)
L = _lib.lib()
for nt_imgs in (1, 2, 4, 16, 64):
    nt = nt_imgs * kp
    groups = n_img // nt_imgs
    items = np.array([(starts[i], kp, g * nt, nt) for i in range(n_img) for g in range(groups)], np.int32)
    items_dev = torch.from_numpy(items).cuda()
    m = len(items)
    for rep in range(3):
        counts = torch.zeros((m,), dtype=torch.int32, device=src.device)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ops.check(L.isx_knn2_l2_count_items(ops._ptr(P.desc), ops._ptr(P.nrm), ops._ptr(P.cin), P.d_pad, ops._ptr(items_dev), m, kp,
                                            0.65, ops._ptr(counts), 0, ops._stream()))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    mfma = m * (kp / 32) * (nt / 32) * 4
    print(f"nt={nt:7d} items={m:6d} {dt*1e3:7.2f} ms  {dt*1e9/(mfma/1024):.2f} ns per MFMA per SIMD  sum={int(counts.sum())}", flush=True)

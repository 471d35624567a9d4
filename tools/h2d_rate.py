"""Host -> HBM upload of the 512 x 4096 x 128 uint8 descriptor set (the PCIe-inclusive leg of bench.py's step)."""
import time, numpy as np, torch
a = np.random.default_rng(0).integers(0, 200, (512 * 4096, 128), dtype=np.uint8)
t = torch.from_numpy(a)
p = t.pin_memory()
for name, src in (("pageable", t), ("pinned", p)):
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d = src.to("cuda", non_blocking=True); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{name}: {best*1e3:.1f} ms  {a.nbytes/best/1e9:.1f} GB/s")

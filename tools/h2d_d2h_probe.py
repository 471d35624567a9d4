"""Diagnostic: host<->device copy times of one 1080p frame / mosaic region, as the frame path issues them."""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ops

dev = torch.device("cuda:0")
ctx = ops.FrameCtx(dev)
arena = torch.zeros((3240, 8000, 3), dtype=torch.uint8, device=dev)
view = arena[1000:2180, 3000:5020]
frame = np.random.default_rng(0).integers(0, 255, (1080, 1920, 3), dtype=np.uint8)


def t(f, n=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"frame_download 1180x2020x3 view : {t(lambda: ops.frame_download(ctx, view)):.3f} ms")
print(f"torch view.cpu()                : {t(lambda: view.cpu()):.3f} ms")
print(f"torch view.contiguous().cpu()   : {t(lambda: view.contiguous().cpu()):.3f} ms")
pin = torch.empty((1180, 2020, 3), dtype=torch.uint8, pin_memory=True)
print(f"torch pinned.copy_(view)        : {t(lambda: (pin.copy_(view, non_blocking=True), torch.cuda.synchronize())):.3f} ms")
print(f"torch from_numpy(frame).to(dev) : {t(lambda: torch.from_numpy(frame).to(dev)):.3f} ms")
pf = torch.empty((1080, 1920, 3), dtype=torch.uint8, pin_memory=True)
def up_pinned():
    pf.numpy()[...] = frame
    return pf.to(dev, non_blocking=True)
print(f"memcpy to pinned + async H2D    : {t(up_pinned):.3f} ms")
d = torch.from_numpy(frame).to(dev)
print(f"bbox_nonzero + .cpu()           : {t(lambda: ops.bbox_nonzero(d).cpu()):.3f} ms")
q = np.random.default_rng(1).integers(0, 255, (1500, 32), dtype=np.uint8)
print(f"frame_match 1500x1500x32 L2     : {t(lambda: ops.frame_match(ctx, q, q[::-1].copy(), 0.65)):.3f} ms")
print(f"frame_match 1500x1500x32 hamming: {t(lambda: ops.frame_match(ctx, q, q[::-1].copy(), 0.65, hamming=True)):.3f} ms")
src = np.random.default_rng(2).uniform(0, 1000, (400, 2)).astype(np.float32)
dst = src + np.float32(5)
print(f"frame_transform 400 pts affine  : {t(lambda: ops.frame_transform(ctx, src, dst, False)):.3f} ms")
print(f"frame_transform 400 pts homogr. : {t(lambda: ops.frame_transform(ctx, src, dst, True)):.3f} ms")

import torch, numpy as np, time
dev = torch.device("cuda:0")
frame = np.random.randint(0, 255, (1080, 1920, 3), dtype=np.uint8)
mos = torch.zeros((1190, 2030, 3), dtype=torch.uint8, device=dev)
pin = torch.empty((1190, 2030, 3), dtype=torch.uint8).pin_memory()
pin_in = torch.empty((1080, 1920, 3), dtype=torch.uint8).pin_memory()
side = torch.cuda.Stream(device=dev)
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def up(): return torch.from_numpy(frame).to(dev)
def down():
    pin.copy_(mos, non_blocking=True); torch.cuda.current_stream().synchronize()
def both():
    with torch.cuda.stream(side):
        pin.copy_(mos, non_blocking=True)
    x = torch.from_numpy(frame).to(dev)
    side.synchronize(); return x
def up_pinned():
    pin_in.numpy()[...] = frame
    x = pin_in.to(dev, non_blocking=True); torch.cuda.current_stream().synchronize(); return x
def both_pinned():
    with torch.cuda.stream(side):
        pin.copy_(mos, non_blocking=True)
    pin_in.numpy()[...] = frame
    x = pin_in.to(dev, non_blocking=True); torch.cuda.current_stream().synchronize()
    side.synchronize(); return x
def up_pinned_nocopy():
    x = pin_in.to(dev, non_blocking=True); torch.cuda.current_stream().synchronize(); return x
print("upload pageable 6.2 MB      %.3f ms" % t(up))
print("download pinned 7.2 MB      %.3f ms" % t(down))
print("both (side stream download) %.3f ms" % t(both))
print("upload via pinned (memcpy + H2D) %.3f ms" % t(up_pinned))
print("H2D from pinned only        %.3f ms" % t(up_pinned_nocopy))
print("both, upload via pinned     %.3f ms" % t(both_pinned))

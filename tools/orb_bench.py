"""On-GPU detector (detect.HipOrb) on a 1080p frame: wall time per detectAndCompute call (device-resident input), the
share of the two stream syncs / host selection, and -- under rocprofv3 --kernel-trace -- the five kernels.
usage: python3 tools/orb_bench.py [reps=50]"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd.detect import HipOrb
from image_stitcher_amd.synthetic import procedural_world

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda", 0)
world = procedural_world(seed=42, h=1300, w=2300, speckle=4)
img = torch.from_numpy(np.ascontiguousarray(world[100:1180, 200:2120])).to(dev)
det = HipOrb(nfeatures=1500)
for _ in range(3):
    kps, des = det.detectAndCompute(img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    kps, des = det.detectAndCompute(img)
dt = (time.perf_counter() - t0) / reps
cand = int(det._out_host.numpy()[:8].view(np.int32)[1])
print(f"1080p BGR: {dt * 1e6:.0f} us per detectAndCompute, {len(kps)} keypoints of {cand} NMS survivors", flush=True)
host_det = HipOrb(nfeatures=1500, device_select=False)
for _ in range(3):
    host_det.detectAndCompute(img)
t0 = time.perf_counter()
for _ in range(reps):
    host_det.detectAndCompute(img)
print(f"           {(time.perf_counter() - t0) / reps * 1e6:.0f} us with the selection on the host (device_select=False)", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    det.detectAndCompute(img)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)

"""Diagnostic: wall time of ArbitraryStitcher.get_collections() through the class API on BASELINE's set (512 images x
4096 x 128-d, descriptors handed out by a stub detector, as uint8 and as the float32 cv2.SIFT returns), with a
cProfile listing of the host side."""
import sys, os, time, cProfile, pstats
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ArbitraryStitcher
from image_stitcher_amd.synthetic import overlapping_crops_descriptors

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
des, scene_of = overlapping_crops_descriptors(seed=1234, n_scenes=n_scenes, imgs_per_scene=64, kp=4096, step=448)
n = len(des)


class Stub:
    def __init__(self, arrays):
        self.arrays, self.k = arrays, 0

    def detectAndCompute(self, image, mask=None):
        d = self.arrays[self.k % len(self.arrays)]
        self.k += 1
        return [None] * len(d), d


img = np.zeros((8, 8, 3), np.uint8)
for name, arrays in (("uint8", [d for d in des]), ("float32", [d.astype(np.float32) for d in des])):
    for rep in range(2):
        ar = ArbitraryStitcher(images=[img] * n, with_tqdm=False, log_level=3, detector=Stub(arrays))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cols = ar.get_collections()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = sorted(map(sorted, cols)) == sorted(sorted(np.nonzero(scene_of == s)[0].tolist()) for s in range(n_scenes))
    print(f"{name:8s} get_collections: {dt*1e3:8.1f} ms  ({n*(n-1)//2/dt:9.0f} pairs/s through the class API)  components ok: {ok}", flush=True)
    ar = ArbitraryStitcher(images=[img] * n, with_tqdm=False, log_level=3, detector=Stub(arrays))
    pr = cProfile.Profile(); pr.enable(); ar.get_collections(); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(10)

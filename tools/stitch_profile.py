"""Diagnostic: cProfile of the 1080p ConsecutiveStitcher loop of bench.py (host-side breakdown per frame).
usage: python3 tools/stitch_profile.py [frames=8]"""
import sys, os, time, cProfile, pstats, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ConsecutiveStitcher
from image_stitcher_amd.detect import NumpyOrbLite
from image_stitcher_amd.synthetic import crop_path, procedural_world

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8


class CachedDetector:
    """Replays detections recorded in a first pass (the very objects, no per-call construction), so the profiled
    pass shows the stitcher only."""
    def __init__(self, det):
        self.det, self.log, self.replay, self.k = det, [], False, 0

    def detectAndCompute(self, image, mask=None):
        if self.replay:
            self.k += 1
            return self.log[self.k - 1]
        r = self.det.detectAndCompute(image, mask)
        self.log.append(r)
        return r


world = procedural_world(seed=42, h=2160, w=1920 + int(0.15 * 1920 * (n + 4)) + 400)
frames, _ = crop_path(world, n + 3, (1080, 1920), step_frac=0.15, max_rot_deg=1.0, seed=43)
det = CachedDetector(NumpyOrbLite(nfeatures=1500))


def run():
    st = ConsecutiveStitcher(log_level=3, output_dir=tempfile.mkdtemp(prefix="isx_prof_"), algorithm=1,
                             consecutive_range=2, detector=det)
    for f in frames:
        st.stitch_consecutive(f)
    torch.cuda.synchronize()


run()
det.replay, det.k = True, 0
t0 = time.perf_counter()
run()
print(f"replayed pass: {(time.perf_counter() - t0) / len(frames) * 1e3:.2f} ms per frame (detector replayed from cache)")
det.k = 0
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
ps = pstats.Stats(pr)
ps.sort_stats("cumulative").print_stats(45)

"""Frame rate of the 1080p ConsecutiveStitcher loop with the two CPU detections of a frame run one after the
other (default) or concurrently (parallel_detect=True).  usage: python3 tools/parallel_detect_rate.py [frames=8]"""
import sys, os, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
import torch
from image_stitcher_amd import ConsecutiveStitcher
from image_stitcher_amd.detect import NumpyOrbLite
from image_stitcher_amd.synthetic import crop_path, procedural_world

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
world = procedural_world(seed=42, h=2160, w=1920 + int(0.15 * 1920 * (n + 4)) + 400)
frames, _ = crop_path(world, n + 1, (1080, 1920), step_frac=0.15, max_rot_deg=1.0, seed=43)
for par in (False, True, False, True):
    st = ConsecutiveStitcher(log_level=0, output_dir=tempfile.mkdtemp(prefix="isx_pd_"), algorithm=1, consecutive_range=2,
                             detector=NumpyOrbLite(nfeatures=1500), parallel_detect=par)
    st.stitch_consecutive(frames[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for f in frames[1:]:
        st.stitch_consecutive(f)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"parallel_detect={par!s:5}: {n / dt:6.2f} frames/s ({dt / n * 1e3:.1f} ms per frame), mosaic {tuple(st.refs[-1].shape)}", flush=True)

"""Worker of tests/test_gpu_strips.py -- launched by torch.distributed.run with 2 ranks that share cuda:0 and talk over
gloo.  Runs every scenario of strip_scenarios.py on a StripMosaic through Stitcher.stitch (given transforms) and one
ArbitraryStitcher.stitch_collections with strip_group set; rank 0 writes what the test compares."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "image-stitcher_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(outdir, which="small"):
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from image_stitcher_amd import ArbitraryStitcher, Stitcher, TransformDetails
    from image_stitcher_amd.detect import NumpyOrbLite
    from image_stitcher_amd.strips import StripMosaic
    from strip_scenarios import collection_images, scenarios, uhd_scenarios
    pts = np.zeros((5, 1, 2), np.float32)
    results = {}
    for name, frames, Ms, type_, blender, contrib in (uhd_scenarios() if which == "uhd" else scenarios()):
        st = Stitcher(log_level=3, type=type_, blender=blender, ref_image_contrib=contrib, consecutive_range=1,
                      detector=object(), device=dev)
        m = StripMosaic.from_image(frames[0], dist.group.WORLD, dev)
        per_frame = []
        for f, M in zip(frames[1:], Ms):
            st._inplace_ok = True
            res = st.stitch(m, f, tf=TransformDetails(np.asarray(M, np.float64), pts, pts))
            assert res not in (None, False), name
            m, corners, _ = res
            assert isinstance(m, StripMosaic)
            full = np.asarray(m).copy()                       # collective gather
            # what this rank holds: zero outside its rows of the mosaic (the arena invariant), its rows only inside
            a = m.arena
            own = None
            if a is not None:
                outside = a.t.clone()
                outside[a.y:a.y + a.h, a.x:a.x + a.w] = 0
                assert int(outside.max()) == 0, f"{name}: rank {rank} holds non-zero bytes outside its strip"
                y0, y1 = m.local_rows()
                own = (y0 - m.wy0, y1 - m.wy0)
                assert a.h == y1 - y0 and a.w == m.w
                assert np.array_equal(a.view().cpu().numpy(), full[own[0]:own[1]])
            owns = [None, None]
            dist.all_gather_object(owns, own)
            per_frame.append((full, corners, owns))
        results[name] = (per_frame, dict(st.frame_stats))
    if which == "uhd":
        if rank == 0:
            np.save(os.path.join(outdir, "results_uhd.npy"), np.array([results, None], dtype=object), allow_pickle=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    # the greedy merge with a real detector: the detector regions are assembled from both strips
    images = collection_images()
    ar = ArbitraryStitcher(images=list(images), with_tqdm=False, log_level=3, algorithm=1, detector=NumpyOrbLite(nfeatures=500),
                           min_match_count=40, output_dir=os.path.join(outdir, "strips"), device=dev)
    ar.strip_group = True
    cols = ar.get_collections()
    assert ar.stitch_collections(cols) == []
    if rank == 0:
        np.save(os.path.join(outdir, "results.npy"), np.array([results, cols], dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(*sys.argv[1:3])

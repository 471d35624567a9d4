#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X stitching core.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2], the configuration the metric "image pairs matched/sec (SIFT 4k kp)"
is quoted on; it fits one GPU): ArbitraryStitcher.get_collections over 512 shuffled crops x 4096
SIFT-like keypoints (integer-valued 128-d descriptors, 8 planted scenes of 64 overlapping images,
SURVEY.md section 8d).  One STEP = one full all-pairs pass: descriptor preparation (+ rows sorted by norm) of the
rank's own images -> [all_gather of the prepared rows over RCCL when N>1, started asynchronously: the pairs among the rank's
own images are counted while it travels] -> fused kNN-2 + Lowe-ratio
counts of both directions of this rank's share of the 130,816 unordered pairs (one launch over all pairs,
one over the overlapping ones) -> gather of the int32 good-match counts -> host union-find on rank 0 (deferred by one step:
it runs while the next step's kernel is on the GPU; the last one before the closing fence).  Inputs (the raw uint8 descriptor shard) are resident in HBM
when the timed region starts.  The pair space is fixed and split over the ranks => "scaling":
"strong".

The JSON line also carries:
  roofline     -- the dominant kernel (knn2_pair_filter_kernel, with the second-stage launch it hands the overlapping
                  pairs to) against the dense int8 MFMA peak, from the duration measured live with HIP events;
  cpu_baseline -- the CPU oracle's port of cv2.BFMatcher().knnMatch(float32) + ratio loop, timed on
                  a bounded sample of the same pairs on this host (rank 0, N=1 only);
  stitch       -- (N=1 only, skip with --no-stitch) the second half of BASELINE.json's metric, "HD
                  consecutive stitch Hz": a ConsecutiveStitcher.stitch_consecutive loop over synthetic
                  1920x1080 BGR frames (BASELINE.json configs[1]); see stitch_workload() below.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "image-stitcher_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import torch

INT8_DENSE_PEAK_TOPS = 5000.0  # MI355X_MICROARCH.md: i8 MFMA = 2x bf16 per clock, bf16 ~2.5 PF dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scenes", type=int, default=8)
    ap.add_argument("--imgs-per-scene", type=int, default=64)
    ap.add_argument("--kp", type=int, default=4096)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stitch", action="store_true", help="skip the 1080p consecutive-stitch leg")
    ap.add_argument("--stitch-frames", type=int, default=128)
    return ap.parse_args()


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured achievable)


class _TimedDetector:
    """Wraps the CPU feature detector (out of scope, stays on the CPU as in the reference) to account its time.
    With record=True every result is kept; replay() then turns the object into a zero-cost detector that hands the
    recorded results back in order, so that the SAME stream can be run again with detection taken out."""

    def __init__(self, det, record=False):
        self.det, self.seconds, self.calls = det, 0.0, 0
        self.log = [] if record else None
        self._replay = None

    def replay(self):
        self._replay = iter(self.log)
        self.seconds, self.calls = 0.0, 0

    def detectAndCompute(self, image, mask=None):
        if self._replay is not None:
            self.calls += 1
            return next(self._replay)
        t0 = time.perf_counter()
        r = self.det.detectAndCompute(image, mask)
        self.seconds += time.perf_counter() - t0
        self.calls += 1
        if self.log is not None:
            self.log.append(r)
        return r


def stitch_workload(dev, n_frames):
    """BASELINE.json configs[1]: ConsecutiveStitcher on a 1080p stream, ORB-like binary descriptors,
    consecutive_range=2, default blender (gradient) and type (affine partial), one MI355X.  Frames are crops of a
    seeded procedural texture along a SERPENTINE mapping path (15 % of the frame width per frame, +-1 degree, 28
    frames per row, rows half a frame apart), so the mosaic grows in both axes; n_frames >= 128 by default, which is
    what "sustained" is read from: the rate of the first and of the last 32 frames (per-frame cost must not depend on
    the mosaic's area).  Hz is frames / wall second exactly as the reference's tests/test.py:82-89 prints it; detection
    runs on the CPU (cv2 ORB when importable, else the numpy stand-in of image_stitcher_amd.detect) and its share is
    reported so that the GPU hot path can be read on its own."""
    import tempfile
    from image_stitcher_amd import ConsecutiveStitcher, ops
    from image_stitcher_amd.synthetic import crop_poses_torch, procedural_world_torch, serpentine_poses
    try:
        from image_stitcher_amd.detect import make_detector
        new_det = lambda: make_detector(1)
        det, det_name = new_det(), "cv2.ORB_create()"
    except ImportError:
        from image_stitcher_amd.detect import NumpyOrbLite
        new_det = lambda: NumpyOrbLite(nfeatures=1500)
        det, det_name = new_det(), "NumpyOrbLite (numpy/scipy stand-in, cv2 absent)"
    det = _TimedDetector(det, record=True)
    warm = 3
    poses, (wh, ww) = serpentine_poses(n_frames + warm, (1080, 1920), step_frac=0.15, per_row=28, down_steps=3, max_rot_deg=1.0, seed=43)
    frames = crop_poses_torch(procedural_world_torch(dev, seed=42, h=wh, w=ww), poses, (1080, 1920))
    k = len(frames) - warm

    def stream(detector, frames=frames, **kw):
        st = ConsecutiveStitcher(log_level=3, output_dir=tempfile.mkdtemp(prefix="isx_bench_"), algorithm=1,
                                 consecutive_range=2, detector=detector, **kw)
        for f in frames[:warm]:
            st.stitch_consecutive(f)
        torch.cuda.synchronize()
        det.seconds, det.calls = 0.0, 0
        st.phase_seconds.clear()
        for key in st.frame_stats:
            st.frame_stats[key] = 0
        ok, per_frame = 0, []
        t0 = time.perf_counter()
        for f in frames[warm:]:
            t1 = time.perf_counter()
            ok += st.stitch_consecutive(f) is not None
            per_frame.append(time.perf_counter() - t1)  # (every call ends on a stream synchronisation: wall time per frame)
        torch.cuda.synchronize()
        st.per_frame_seconds = per_frame
        return st, ok, time.perf_counter() - t0

    def quartiles(st):
        pf, q = st.per_frame_seconds, max(1, min(32, len(st.per_frame_seconds) // 4))
        first, last = q / sum(pf[:q]), q / sum(pf[-q:])
        return {"frames_per_window": q, "first_hz": first, "last_hz": last, "last_over_first": last / first,
                "final_mosaic_shape": [int(v) for v in st.refs[-1].shape], "frame_stats": dict(st.frame_stats)}

    st, ok, wall = stream(det)
    live = dict(seconds=det.seconds, calls=det.calls, phases=dict(st.phase_seconds), stats=dict(st.frame_stats))
    # the same stream once more with the recorded detections replayed at no cost: everything the stitching core does per
    # frame (upload, frame bbox, region download for the detector, matching, RANSAC, warp + blend, bookkeeping), measured
    # directly instead of as a difference of two large numbers
    det.replay()
    st_rp, ok_rp, wall_rp = stream(det)
    same_rp = bool(torch.equal(st.refs[-1].t, st_rp.refs[-1].t))
    replay_phases = dict(st_rp.phase_seconds)
    sustained = quartiles(st_rp)
    det.seconds, det.calls = live["seconds"], live["calls"]
    # BASELINE configs[1] names "ORB Hamming matching": the same stream with norm="hamming" (cv2.BFMatcher(NORM_HAMMING);
    # the reference's own matcher is L2 for every detector, stitcher.py:85).  Other matches give other transforms and so
    # other mosaic regions for the detector, hence its own recorded run -- on the first 24 frames, the CPU detector takes
    # a quarter of a second per frame -- replayed for the timing.
    kh = min(24, k)
    det_h = _TimedDetector(new_det(), record=True)
    frames_h = frames[:warm + kh]
    st_h, ok_h, _ = stream(det_h, frames=frames_h, norm="hamming")
    det_h.replay()
    st_hr, ok_hr, wall_hr = stream(det_h, frames=frames_h, norm="hamming")
    # ... and the L2 stream over the same 24 frames, for the per-phase comparison
    det.replay()
    st_l, ok_l, wall_l = stream(det, frames=frames_h)
    det.seconds, det.calls = live["seconds"], live["calls"]
    hamming = {"value": kh / wall_hr, "unit": "frames/s excluding CPU detection (replayed)", "frames": kh, "stitched": int(ok_hr),
               "same_mosaic_as_its_live_run": bool(torch.equal(st_h.refs[-1].t, st_hr.refs[-1].t)),
               "phase_ms_per_frame": {n: v / kh * 1e3 for n, v in st_hr.phase_seconds.items()},
               "l2_same_frames": {"value": kh / wall_l, "phase_ms_per_frame": {n: v / kh * 1e3 for n, v in st_l.phase_seconds.items()}}}
    # the opt-in on-GPU detector (detect.HipOrb): no CPU detector in the loop at all.  It is not cv2.ORB
    # (different keypoints), and FAST corners need a texture with corners: the same path over the same world with
    # small high-contrast blobs added (speckle=4).
    from image_stitcher_amd.detect import HipOrb
    frames_s = crop_poses_torch(procedural_world_torch(dev, seed=42, h=wh, w=ww, speckle=4), poses, (1080, 1920))
    hip_det = HipOrb(nfeatures=1500)
    stream(hip_det, frames=frames_s[:warm + 8])  # (first run: workspace allocation)
    st_g, ok_g, wall_g = stream(hip_det, frames=frames_s)
    gpu_det = {"value": k / wall_g, "unit": "frames/s", "ms_per_frame": wall_g / k * 1e3, "stitched": int(ok_g), "frames": k,
               "detector": "HipOrb(nfeatures=1500): FAST-9 + Harris + NMS + orientation + steered BRIEF on the GPU (opt-in, "
                           "not cv2.ORB: parity with cv2 unpinned by construction)",
               "phase_ms_per_frame": {n: v / k * 1e3 for n, v in st_g.phase_seconds.items()},
               "sustained": quartiles(st_g)}
    # kernel-level roofline of warp + feather blend on one representative 1080p frame: the launches of ONE
    # isx_frame_blend call (frame_warp_kernel, frame_extent_kernel, frame_gradient_kernel), queued back to back
    # between two HIP events on the launch stream.  Algorithmic bytes = frame read + mosaic read + write inside
    # the warped frame's region; executed = + the staged warped image written once and read once.
    img = torch.from_numpy(frames[-1]).to(dev)
    H, W = 1080 * 3, 1920 * 3
    M = np.array([[0.9998, -0.0175, 300.0], [0.0175, 0.9998, 12.0]])
    rect = (1920 + 250, 1080 - 40, 2030, 1190)
    arena = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    arena[1080:2160, 1920:3840] = 90
    ctx = ops.FrameCtx(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50

    def frame_kernels():  # exactly the launches Stitcher.stitch issues per frame for warp + feather blend + both bboxes
        ops.frame_blend(ctx, img, M, False, (1920, 1080), (W, H), arena, (0, 0), rect, 1, 0.2, 10, "f32", fetch=False)

    for _ in range(3):
        frame_kernels()
    fr = ops.frame_result(ctx)
    assert fr["done"] and fr["rows_ok"] and fr["wbbox"], fr
    timing = "back-to-back launches between two HIP events (no host gaps: the queue stays ahead of the GPU)"
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        frame_kernels()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    # the synchronous form Stitcher.stitch calls (kernels + one read-back + host fold)
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.frame_blend(ctx, img, M, False, (1920, 1080), (W, H), arena, (0, 0), rect, 1, 0.2, 10, "f32")
    sync_us = (time.perf_counter() - t0) / reps * 1e6
    roi_px = rect[2] * rect[3] * 3
    # the same frame through the reference's own sequence on the CPU (stitcher.py:178-179, :216-225: pad both
    # images into the canvas, corners of the frame, warp the padded frame to canvas size, gradient merge,
    # corners of the warped image, crop), restated by the oracle -- one frame, it takes seconds
    from oracle import orc
    ref_np, img_np = np.asarray(st_l.refs[-1]), frames_h[-1]   # (the mosaic after 24 frames: the reference's cost grows with its area)
    t0 = time.perf_counter()
    cur, refc = orc.get_padded_images(img_np, ref_np)
    orc.bbox_nonzero(cur)
    Hc, Wc = refc.shape[:2]
    Mc = np.array([[0.9998, -0.0175, 0.15 * 1920], [0.0175, 0.9998, 12.0]])
    warped = orc.warp(cur, Mc, (Wc, Hc), False)
    (bx0, by0), (bx1, by1) = orc.bbox_nonzero(warped)
    gx0, gy0, gx1, gy1 = max(0, bx0 - 50), max(0, by0 - 50), min(Wc, bx1 + 50), min(Hc, by1 + 50)
    refc[gy0:gy1, gx0:gx1] = orc.gradient_blend(warped[gy0:gy1, gx0:gx1], refc[gy0:gy1, gx0:gx1], 10)
    orc.bbox_nonzero(refc)
    cpu_frame_s = time.perf_counter() - t0
    alg_bytes = img.numel() + 2 * roi_px
    executed_bytes = img.numel() + 2 * roi_px + 2 * roi_px  # + the staged warped image, written once and read once
    frame_traffic = frame_traffic_src = None
    tf = os.path.join(ROOT, "profiles", "frame_traffic.json")
    if os.path.exists(tf):  # PMC counters of the three kernels at this frame size (tools/img_bench), collected separately
        ft = json.load(open(tf))
        frame_traffic = ft.get("hbm_bytes_per_frame")
        frame_traffic_src = f"profiles/frame_traffic.json, {ft.get('measured', 'round 2')}: not measured in this run"
    return {
        "metric": "HD consecutive stitch Hz", "value": k / wall, "unit": "frames/s",
        "frames": k, "stitched": int(ok), "ms_per_frame": wall / k * 1e3,
        "detector": det_name, "detector_ms_per_frame": det.seconds / k * 1e3, "detector_calls": det.calls,
        "hz_excluding_cpu_detection": k / max(wall_rp, 1e-9),
        "hz_excluding_cpu_detection_how": "the same stream re-run with the recorded detections replayed at zero cost "
                                          f"({wall_rp / k * 1e3:.3f} ms per frame, same mosaic: {same_rp}); as a difference "
                                          f"(wall - detector seconds) of the live run: {k / max(wall - det.seconds, 1e-9):.1f}",
        "frame_stats": live["stats"],
        "phase_ms_per_frame": {n: v / k * 1e3 for n, v in live["phases"].items()},
        "phase_ms_per_frame_replayed": {n: v / k * 1e3 for n, v in replay_phases.items()},
        "sustained": sustained,
        "hamming": hamming,
        "gpu_detection": gpu_det,
        "cpu_baseline": {"value": 1.0 / cpu_frame_s, "unit": "frames/s excluding detection", "cores": 1, "kind": "port",
                         "sample": f"1 frame: pad + warp + gradient merge + 3 bboxes on the {Wc}x{Hc} canvas as the reference "
                                   f"does them (oracle/orc_image.c), {cpu_frame_s:.2f} s"},
        "config": {"workload": f"ConsecutiveStitcher 1920x1080 BGR stream of {k} frames on a serpentine path, consecutive_range=2, type=0 "
                               "(affine partial), blender=1 (gradient), binary descriptors matched with L2 as the reference's "
                               "cv2.BFMatcher() does (`hamming`: the same stream with norm='hamming')",
                   "final_mosaic_shape": [int(v) for v in st.refs[-1].shape]},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": frame_traffic, "traffic_source": frame_traffic_src,
                     "kernel": "frame_warp_kernel<3,false,false> + frame_extent_kernel + frame_gradient_kernel<3> (the 3 launches "
                               "of one isx_frame_blend call: warp, both bounding boxes, extents, feather blend in place)",
                     "kernel_us": us, "timing": timing, "synchronous_call_us": sync_us, "algorithmic_bytes": int(alg_bytes), "executed_bytes": int(executed_bytes)},
    }


def cpu_baseline(des, pairs, ratio, seconds, gpu_counts=None):
    """Oracle port of the reference's per-pair CPU work (float32 descriptors as cv2.SIFT returns them:
    BFMatcher L2 + Python ratio loop), multi-threaded over query rows like cv2's parallel_for_.  The oracle's
    counts of the sampled pairs are compared with the GPU's counts of the same pairs (gpu_counts[p] = both
    directions of pairs[p], from the last step): "sample_counts_equal"."""
    from oracle import orc
    threads = max(1, min(16, os.cpu_count() or 1))
    rng = np.random.default_rng(0)
    order = rng.permutation(len(pairs))
    done, equal = 0, True
    t0 = time.perf_counter()
    for p in order:
        i, j = pairs[p]
        a = des[i].astype(np.float32)
        b = des[j].astype(np.float32)
        cij = orc.match_count(a, b, ratio, nthreads=threads)
        cji = orc.match_count(b, a, ratio, nthreads=threads)
        if gpu_counts is not None:
            equal = equal and (cij, cji) == (int(gpu_counts[p, 0]), int(gpu_counts[p, 1]))
        done += 1
        if time.perf_counter() - t0 > seconds and done >= 2:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "image pairs matched/s", "cores": threads, "kind": "port",
            "sample_counts_equal": bool(equal) if gpu_counts is not None else None,
            "sample": f"{done} unordered pairs (both directed knnMatch float32 4096x4096x128 + ratio loops) "
                      f"drawn at random from the same 512-image set, {dt:.1f} s, oracle/orc_match.c with OpenMP"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # ISX_BENCH_BACKEND=gloo is a rehearsal switch: several ranks share the GPUs that exist and exchange through
    # host memory, so the multi-rank code path can be run on a one-GPU box.  The measured configuration is
    # one rank per GPU over RCCL ("nccl").
    backend = os.environ.get("ISX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from image_stitcher_amd import ops, pairs as prs
    from image_stitcher_amd.synthetic import overlapping_crops_descriptors

    n = args.scenes * args.imgs_per_scene
    kp, d, ratio, min_match = args.kp, 128, 0.65, 5
    assert n % world == 0, "image count must divide over the ranks"
    des, scene_of = overlapping_crops_descriptors(seed=1234, n_scenes=args.scenes,
                                                  imgs_per_scene=args.imgs_per_scene, kp=kp, step=448, d=d)
    per = n // world
    local = torch.from_numpy(des[rank * per:(rank + 1) * per]).to(dev)  # this rank's "detected" shard
    starts = np.arange(n, dtype=np.int64) * kp
    counts = np.full((n,), kp, np.int64)
    all_pairs = prs.unordered_pairs(n)
    my_pairs = prs.shard_pairs(all_pairs, rank, world)
    # one work item per unordered pair: isx_knn2_l2_count_pairs forms a pair's 4096 x 4096 products once for both directions
    # (max-only filter + exact recount of the candidates); the directions it reports as overflowing (overlapping images)
    # run on the two-pass kernel isx_knn2_l2_count_items inside the same step.  ISX_BENCH_TWO_PASS=1: round 2's path.
    two_pass = os.environ.get("ISX_BENCH_TWO_PASS") == "1"
    # N > 1: the all_gather of the prepared rows is started asynchronously and hidden behind the pairs a rank can form from its
    # OWN images (C(n / N, 2) of them, the same number on every rank); the pairs that span two shards are dealt round-robin and
    # run when the gather has landed.  ISX_BENCH_OVERLAP=0: gather first, then one launch over pairs p % N (the earlier form).
    overlap = world > 1 and not two_pass and os.environ.get("ISX_BENCH_OVERLAP", "1") == "1"
    result_pairs = all_pairs  # the pair each row of rank 0's gathered counts belongs to
    if overlap:
        loc_lists, x_lists = prs.overlap_shards(n, world)
        my_loc, my_x = loc_lists[rank], x_lists[rank]
        my_pairs = np.concatenate([my_loc, my_x])
        plan_loc = ops.plan_items(prs.pair_items(my_loc - rank * per, starts[:per], counts[:per]), dev) if len(my_loc) else None
        plan = ops.plan_items(prs.pair_items(my_x, starts, counts), dev)
        list_lens = [len(a) + len(b) for a, b in zip(loc_lists, x_lists)]
        result_pairs = np.concatenate([np.concatenate([a, b]) for a, b in zip(loc_lists, x_lists)])
    else:
        plan = ops.plan_items(prs.directed_items(my_pairs, starts, counts) if two_pass else prs.pair_items(my_pairs, starts, counts), dev)
    stage = {}
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev_loc = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    # step start | all_gather landed | after the kernels | after gather | [4] after local prep | [5] after the local pairs (overlap)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    kernel_ms, parts = [], []
    result = {}

    # rank 0's host part of a step (counts read-back + union-find) is deferred by one step in the timed loop: the counts go to
    # pinned memory by an asynchronous copy behind the step's kernels, and the union-find runs while the NEXT step's pair kernel
    # is on the GPU (ops.knn2_l2_pairs(while_running=...)); the last step's runs before the closing fence, inside the timed region
    host_bufs = [torch.empty((len(all_pairs), 2), dtype=torch.int32).pin_memory() for _ in range(2)] if rank == 0 else None
    pending, step_no = [None], [0]

    def finish_pending():
        if pending[0] is not None:
            hb, done = pending[0]
            pending[0] = None
            done.synchronize()
            h = hb.numpy()
            comps, _ = prs.components_from_counts(n, result_pairs, h[:, 0], h[:, 1], min_match)
            result["comps"], result["counts"] = comps, h.copy()

    def step(record):
        if record:
            marks[0].record()
        # every rank prepares (and, for the pair kernel, sorts by norm) the rows of ITS images only, and the prepared rows
        # travel: the all_gather moves 9 % more bytes than raw descriptors would, and no rank repeats another's preparation
        # (round 2 gathered the raw bytes and prepared all n images on every rank: ~1.2 ms of every step, whatever N)
        Pl = ops.prepare_descriptor_tensor(local.reshape(-1, d), starts[:per], counts[:per])
        if not two_pass:
            Pl = ops.sort_rows_by_norm(Pl)
        if record:
            marks[4].record()
        if overlap:
            landed = prs.all_gather_prepared(Pl, starts, counts, async_op=True)  # travels on the backend's stream from here on
            st_loc = {}
            c_loc = (ops.knn2_l2_pairs(Pl, None, ratio, plan=plan_loc, events=ev_loc if record else None, stats=st_loc,
                                       while_running=finish_pending) if plan_loc is not None
                     else torch.zeros((0, 2), dtype=torch.int32, device=dev))
            if record:
                marks[5].record()
            P = landed()  # the current stream waits for the gather (the host does not)
            if record:
                marks[1].record()
            c_x = ops.knn2_l2_pairs(P, None, ratio, plan=plan, events=ev if record else None, stats=stage)
            stage["overflow_directions"] = stage.get("overflow_directions", 0) + st_loc.get("overflow_directions", 0)
            c = torch.cat([c_loc.view(-1, 2), c_x.view(-1, 2)])
        else:
            P = prs.all_gather_prepared(Pl, starts, counts) if world > 1 else Pl
            if record:
                marks[1].record()
            if two_pass:
                c = ops.knn2_l2_items(P, None, ratio, plan=plan, events=ev if record else None)
            else:
                c = ops.knn2_l2_pairs(P, None, ratio, plan=plan, events=ev if record else None, stats=stage, while_running=finish_pending)
        if record:
            marks[2].record()
        if overlap:
            allc = prs.gather_rank_lists(c, max(list_lens), rank, world)  # rank 0: (world, longest list, 2)
            if rank == 0:
                allc = torch.cat([allc[r, :list_lens[r]] for r in range(world)])  # rows in result_pairs' order
        else:
            allc = prs.gather_counts(c.view(-1, 2), len(all_pairs), rank, world)
        if record:
            marks[3].record()
            torch.cuda.synchronize()  # (instrumented steps only: the host figure below must not include the wait for the GPU)
        t_host = time.perf_counter()
        if rank == 0:
            finish_pending()  # (two-pass path, or nothing left to do)
            hb = host_bufs[step_no[0] % 2]
            step_no[0] += 1
            hb.copy_(allc, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            pending[0] = (hb, done)
            if record:
                finish_pending()  # instrumented steps: in line, so that host_ms_rank0 is the plain cost
        if record:
            torch.cuda.synchronize()
            host_ms = (time.perf_counter() - t_host) * 1e3
            # per-phase GPU time of this rank's step: all_gather of the prepared rows (overlap: only what of it is still
            # outstanding when the rank's local pairs are done) | desc_prep + sort of the rank's own images + count kernels |
            # gather; host = counts read-back + union-find (rank 0)
            if overlap:
                kernel_ms.append(ev[0].elapsed_time(ev[1]) + (ev_loc[0].elapsed_time(ev_loc[1]) if plan_loc is not None else 0.0))
                parts.append((marks[5].elapsed_time(marks[1]), marks[0].elapsed_time(marks[5]) + marks[1].elapsed_time(marks[2]),
                              marks[2].elapsed_time(marks[3]), host_ms))
            else:
                kernel_ms.append(ev[0].elapsed_time(ev[1]))
                parts.append((marks[4].elapsed_time(marks[1]), marks[0].elapsed_time(marks[4]) + marks[1].elapsed_time(marks[2]),
                              marks[2].elapsed_time(marks[3]), host_ms))

    def fence():
        finish_pending()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # kernel-only duration for the roofline: separate, event-bracketed launches (not in the timed loop)
    for _ in range(3):
        step(True)
    # attribution of a step over the ranks (so that a scaling-efficiency figure can be explained): every rank's mean
    # all_gather / kernel / gather / host milliseconds, max and min over the ranks
    mine = torch.tensor([float(np.mean(kernel_ms))] + [float(v) for v in np.mean(np.array(parts), 0)], dtype=torch.float64, device=dev)
    if world > 1:
        hi, lo = mine.clone(), mine.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    else:
        hi = lo = mine
    hi, lo = hi.cpu().numpy(), lo.cpu().numpy()
    breakdown = {"kernel_ms": {"max": hi[0], "min": lo[0]}, "allgather_ms": {"max": hi[1], "min": lo[1]},
                 "prep_plus_kernel_ms": {"max": hi[2], "min": lo[2]}, "gather_ms": {"max": hi[3], "min": lo[3]},
                 "host_ms_rank0": float(np.mean(np.array(parts), 0)[3]),
                 "note": "per rank: HIP events around each phase of 3 instrumented steps after the timed loop; host = counts "
                         "read-back + union-find on rank 0"}

    if rank == 0:
        comps = result["comps"]
        truth = sorted(sorted(np.nonzero(scene_of == s)[0].tolist()) for s in range(args.scenes))
        ok = sorted(sorted(c) for c in comps) == truth
        pairs_per_step = len(all_pairs)
        ms_per_step = elapsed / args.steps * 1e3
        k_ms = float(np.mean(kernel_ms))
        ops_per_launch = 2.0 * kp * kp * d * len(my_pairs)  # 2*Nq*Nt*D per unordered pair
        achieved = ops_per_launch / (k_ms * 1e-3) / 1e12
        # roofline.traffic is NOT measured in this run: it is the PMC figure (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes)
        # committed under profiles/ for the kernel launched here on this very workload, or null when there is none
        kernel_name = "knn2_count16w_kernel" if two_pass else "knn2_pair_filter_kernel"
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "knn2_traffic.json")
        if os.path.exists(tf) and world == 1 and (args.scenes, args.imgs_per_scene, kp) == (8, 64, 4096):
            ent = json.load(open(tf)).get("kernels", {}).get(kernel_name)
            if ent:
                traffic = ent["hbm_bytes_per_launch"]
                traffic_src = f"profiles/knn2_traffic.json [{kernel_name}], {ent.get('measured')}: not measured in this run"
        line = {
            "metric": "image pairs matched/sec (SIFT 4k kp)",
            "value": pairs_per_step * args.steps / elapsed,
            "unit": "image pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "i8",
            "data": "synthetic",
            "config": {"workload": f"ArbitraryStitcher.get_collections all-pairs: {n} images x {kp} SIFT-like kp x {d}-d "
                                   f"({pairs_per_step} unordered pairs, both directions, ratio {ratio}, min_match {min_match})",
                       "pairs_per_step": pairs_per_step,
                       "parallelism": f"pair-sharded x{world}" + (" (own-shard pairs first, behind the all_gather)" if overlap else ""),
                       "components_correct": bool(ok)},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": INT8_DENSE_PEAK_TOPS, "unit": "TFLOP/s",
                         "frac": achieved / INT8_DENSE_PEAK_TOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name,
                         "kernel_ms": k_ms,
                         "note": "int8 MFMA ops; algorithmic 2*Nq*Nt*D per unordered pair (the kernel executes 2x: one pass per direction)"
                                 if two_pass else
                                 "int8 MFMA ops; algorithmic 2*Nq*Nt*D per unordered pair.  kernel_ms spans BOTH launches of a step between "
                                 "two HIP events: knn2_pair_filter_kernel over every pair (one distance matrix per pair) and "
                                 f"knn2_count16w_kernel over the {stage.get('overflow_directions')} of {2 * len(my_pairs)} directions the filter "
                                 "reports as overflowing (overlapping images), incl. the list compaction between them",
                         "overflow_directions": None if two_pass else stage.get("overflow_directions"),
                         # context, not a second roofline: what this chip sustains on the kernel's bare instruction mix
                         "sustained_reference": {"value": 3459.0, "unit": "TFLOP/s", "measured": "2026-11-11, not in this run",
                                                 "source": "tools/mfma_rate.hip ('s'): 16 v_mfma_i32_16x16x64_i8 + 40 v_max3_u32 per step, "
                                                           "2 waves per SIMD, SIFT-like operand bytes, no LDS or memory traffic; the clock "
                                                           "gives way under this mix (profiles/r03_knn2_count_pmc.md)"}},
        }
        if overlap:
            line["roofline"]["note"] += ".  N > 1: kernel_ms is the sum of the own-shard call and the cross-shard call"
        line["step_breakdown"] = breakdown
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(des, all_pairs, ratio, args.cpu_seconds, gpu_counts=result["counts"])
        if world == 1 and not args.no_stitch:
            del des, local
            line["stitch"] = stitch_workload(dev, args.stitch_frames)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

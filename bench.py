#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X stitching core.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2], the configuration the metric "image pairs matched/sec (SIFT 4k kp)"
is quoted on; it fits one GPU): ArbitraryStitcher.get_collections over 512 shuffled crops x 4096
SIFT-like keypoints (integer-valued 128-d descriptors, 8 planted scenes of 64 overlapping images,
SURVEY.md section 8d).  One STEP = one full all-pairs pass: [all_gather of the per-rank descriptor
shards over RCCL when N>1] -> descriptor preparation -> ONE fused kNN-2 + Lowe-ratio launch over both
directions of this rank's share of the 130,816 unordered pairs -> gather of the int32 good-match
counts -> host union-find on rank 0.  Inputs (the raw uint8 descriptor shard) are resident in HBM
when the timed region starts.  The pair space is fixed and split over the ranks => "scaling":
"strong".

The JSON line also carries:
  roofline     -- the dominant kernel (knn2_l2_kernel<4>) against the dense int8 MFMA peak, from the
                  kernel duration measured live with HIP events around the launch;
  cpu_baseline -- the CPU oracle's port of cv2.BFMatcher().knnMatch(float32) + ratio loop, timed on
                  a bounded sample of the same pairs on this host (rank 0, N=1 only).
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
    return ap.parse_args()


def cpu_baseline(des, pairs, ratio, seconds):
    """Oracle port of the reference's per-pair CPU work (float32 descriptors as cv2.SIFT returns them:
    BFMatcher L2 + Python ratio loop), multi-threaded over query rows like cv2's parallel_for_."""
    from oracle import orc
    threads = max(1, min(16, os.cpu_count() or 1))
    rng = np.random.default_rng(0)
    sample = pairs[rng.permutation(len(pairs))]
    done = 0
    t0 = time.perf_counter()
    for (i, j) in sample:
        a = des[i].astype(np.float32)
        b = des[j].astype(np.float32)
        orc.match_count(a, b, ratio, nthreads=threads)
        orc.match_count(b, a, ratio, nthreads=threads)
        done += 1
        if time.perf_counter() - t0 > seconds and done >= 2:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "image pairs matched/s", "cores": threads, "kind": "port",
            "sample": f"{done} unordered pairs (both directed knnMatch float32 4096x4096x128 + ratio loops) "
                      f"drawn at random from the same 512-image set, {dt:.1f} s, oracle/orc_match.c with OpenMP"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

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
    plan = ops.plan_items(prs.directed_items(my_pairs, starts, counts), dev)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    kernel_ms = []
    result = {}

    def step(record):
        full = prs.all_gather_descriptors(local) if world > 1 else local
        P = ops.prepare_descriptor_tensor(full.reshape(-1, d), starts, counts)
        c = ops.knn2_l2_items(P, None, ratio, plan=plan, events=ev if record else None)
        allc = prs.gather_counts(c.view(-1, 2), len(all_pairs), rank, world)
        if rank == 0:
            h = allc.cpu().numpy()
            comps, _ = prs.components_from_counts(n, all_pairs, h[:, 0], h[:, 1], min_match)
            result["comps"] = comps
        if record:
            torch.cuda.synchronize()
            kernel_ms.append(ev[0].elapsed_time(ev[1]))

    def fence():
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

    if rank == 0:
        comps = result["comps"]
        truth = sorted(sorted(np.nonzero(scene_of == s)[0].tolist()) for s in range(args.scenes))
        ok = sorted(sorted(c) for c in comps) == truth
        pairs_per_step = len(all_pairs)
        ms_per_step = elapsed / args.steps * 1e3
        k_ms = float(np.mean(kernel_ms))
        ops_per_launch = 2.0 * kp * kp * d * len(my_pairs)  # 2*Nq*Nt*D per unordered pair
        achieved = ops_per_launch / (k_ms * 1e-3) / 1e12
        traffic = None
        tf = os.path.join(ROOT, "profiles", "knn2_traffic.json")
        if os.path.exists(tf) and world == 1 and (args.scenes, args.imgs_per_scene, kp) == (8, 64, 4096):
            traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
        line = {
            "metric": "image pairs matched/sec (SIFT 4k kp)",
            "value": pairs_per_step * args.steps / elapsed,
            "unit": "image pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int8 (exact int32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"ArbitraryStitcher.get_collections all-pairs: {n} images x {kp} SIFT-like kp x {d}-d "
                                   f"({pairs_per_step} unordered pairs, both directions, ratio {ratio}, min_match {min_match})",
                       "pairs_per_step": pairs_per_step, "parallelism": f"pair-sharded x{world}",
                       "components_correct": bool(ok)},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": INT8_DENSE_PEAK_TOPS, "unit": "TFLOP/s",
                         "frac": achieved / INT8_DENSE_PEAK_TOPS, "traffic": traffic,
                         "kernel": "knn2_l2_kernel<4>", "kernel_ms": k_ms,
                         "note": "int8 MFMA ops; algorithmic 2*Nq*Nt*D per unordered pair (the kernel executes 2x: one pass per direction)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(des, all_pairs, ratio, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Times knn2_l2_kernel (the index/distance matcher of the `stitch` path) with whatever library is loaded (ISX_LIB_PATH):
all-pairs 4096 x 128 B items (us/item by HIP events) and single-pair launches of frame size (us per launch).
usage: python3 tools/l2_index_ab.py [n_img=32]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ops
from image_stitcher_amd.synthetic import overlapping_crops_descriptors

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
print("lib:", os.environ.get("ISX_LIB_PATH", "default"), flush=True)
for d, kp in ((128, 4096), (61, 2048), (32, 2048)):
    des, _ = overlapping_crops_descriptors(seed=1, n_scenes=2, imgs_per_scene=n_img // 2, kp=kp, step=kp // 9, d=d)
    starts = np.arange(n_img) * kp
    items = np.array([(starts[i], kp, starts[j], kp) for i in range(n_img) for j in range(n_img) if i != j], np.int32)
    P = ops.prepare_descriptor_tensor(torch.from_numpy(des.reshape(-1, d)).to(dev), starts, [kp] * n_img)
    plan = ops.plan_items(items, dev)
    ts = []
    for rep in range(4):
        c, m = ops.knn2_l2_items(P, None, 0.65, plan=plan, want_matches=True, events=ev)
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    chk = int((m[0].to(torch.int64) % 1009).sum())
    print(f"all-pairs d={d:3d} kp={kp}: " + " ".join(f"{t:.3f}" for t in ts) + f" ms  = {min(ts) * 1e3 / len(items):.3f} us/item   counts {int(c.sum())} idx-checksum {chk}", flush=True)
for n in (600, 2000, 4096):
    rng = np.random.default_rng(n)
    q = rng.integers(0, 120, (n, 128), dtype=np.uint8)
    t = rng.integers(0, 120, (n, 128), dtype=np.uint8)
    P = ops.prepare_descriptors([q, t], device=dev)
    it = np.array([(P.starts[0], n, P.starts[1], n)], np.int32)
    plan = ops.plan_items(it, dev)
    ts = []
    for rep in range(20):
        ops.knn2_l2_items(P, None, 0.65, plan=plan, want_matches=True, events=ev)
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
    print(f"single pair {n} x {n} x 128 B: median {np.median(ts):.1f} us  min {min(ts):.1f} us", flush=True)

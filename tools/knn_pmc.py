"""Run the count-only kNN-2 kernel a few times on a mid-size problem (for rocprofv3 --pmc / --kernel-trace).
usage: python3 tools/knn_pmc.py [n_images=96] [reps=3] [mode=count|index] [train_group=4]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ops
from image_stitcher_amd.synthetic import overlapping_crops_descriptors

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 96
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else "count"
tg = int(sys.argv[4]) if len(sys.argv) > 4 else ops.TRAIN_GROUP  # train images resident per XCD (1 = round-1 order)
kp, d = 4096, 128
des, _ = overlapping_crops_descriptors(seed=1, n_scenes=2, imgs_per_scene=n_img // 2, kp=kp, step=448, d=d)
src = torch.from_numpy(des.reshape(-1, d)).cuda()
starts = np.arange(n_img) * kp
P = ops.prepare_descriptor_tensor(src, starts, [kp] * n_img)
items = np.array([(starts[i], kp, starts[j], kp) for i in range(n_img) for j in range(n_img) if i != j], np.int32)
plan = ops.plan_items(items, src.device, train_group=tg)
for rep in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == "count":
        c = ops.knn2_l2_items(P, None, 0.65, plan=plan, verify_all=int(os.environ.get('ISX_COUNT_FLAGS', '0')))  # flags: 0 = count16w, 4 = count16, 2 = 32x32x32
    else:
        c, _ = ops.knn2_l2_items(P, None, 0.65, plan=plan, want_matches=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    up = len(items) / 2
    print(f"{mode} tg={tg} items={len(items)} {dt*1e3:.1f} ms {dt/up*1e6:.2f} us/pair {up/dt:.0f} pairs/s", flush=True)

"""Filter kernel alone (isx_knn2_l2_count_pairs, no overflow stage) on an all-pairs set: ms per launch, and with a
-DISX_PAIR_T=16 build the candidate statistics.  usage: ISX_LIB_PATH=... python3 tools/pairs_diag.py [n_images=128] [sort=1]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ops, pairs as prs, _lib
from image_stitcher_amd.ops import _ptr, _stream
from image_stitcher_amd.synthetic import overlapping_crops_descriptors
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 128
do_sort = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kp, d = (int(sys.argv[3]) if len(sys.argv) > 3 else 4096), 128
des, scene_of = overlapping_crops_descriptors(seed=1234, n_scenes=max(1, n_img // 64), imgs_per_scene=min(64, n_img), kp=kp, step=kp * 448 // 4096, d=d)
n_img = des.shape[0]
if os.environ.get("ISX_DIAG_DATA") == "low":   # power experiment: the same work on bytes of 0/1 (few operand bits toggle in the matrix pipe)
    des = des & 1
elif os.environ.get("ISX_DIAG_DATA") == "random":
    des = np.random.default_rng(1).integers(0, 256, des.shape, dtype=np.uint8)
src = torch.from_numpy(des.reshape(-1, d)).cuda()
starts, counts = np.arange(n_img) * kp, np.full((n_img,), kp)
all_pairs = prs.unordered_pairs(n_img)
P = ops.prepare_descriptor_tensor(src, starts, counts)
if do_sort:
    P = ops.sort_rows_by_norm(P)
run_items, perm, items_dev, inv_dev, inv_np = ops.plan_items(prs.pair_items(all_pairs, starts, counts), src.device)
m = len(run_items)
c = torch.empty((m, 2), dtype=torch.int32, device=src.device)
L = _lib.lib()
ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
for rep in range(3):
    ev[0].record()
    _lib.check(L.isx_knn2_l2_count_pairs(_ptr(P.desc), _ptr(P.nrm), _ptr(P.cin), 128, _ptr(items_dev), m, kp, 0.65, _ptr(c), 0, _stream()))
    ev[1].record(); torch.cuda.synchronize()
    print(f"{os.environ.get('ISX_LIB_PATH', 'product')}: pairs={m} {ev[0].elapsed_time(ev[1]):.2f} ms", flush=True)
if hasattr(L, "isx_pair_stamp_read"):   # -DISX_PAIR_STAMP build: wave 0's shader cycles per phase, mean per pair
    import ctypes
    buf = (ctypes.c_ulonglong * 16)()
    _lib.check(L.isx_pair_stamp_read(buf, 1))
    v = np.array(list(buf), float); n = max(v[15], 1)
    names = ["prologue", "class set-up", "pass-start wait", "chunk loop", "requests + column epilogue", "after the last pass (barrier)", "table scan", "recounts", "issuing the requests"]
    tot = v[:9].sum()
    print("stamps over", int(n), "workgroups: " + ", ".join(f"{nm} {v[k]/n:.0f} ({100*v[k]/tot:.1f} %)" for k, nm in enumerate(names)) + f" | total {tot/n:.0f} cycles per pair")
h = c[inv_dev].cpu().numpy() if inv_np is not None else c.cpu().numpy()
cross = scene_of[all_pairs[:, 0]] != scene_of[all_pairs[:, 1]]
print("values: cross-scene mean %.2f max %d | same-scene: <0: %d, mean of the rest %.2f, histogram of all >=0: %s" % (
    h[cross].mean(), h[cross].max(), (h[~cross] < 0).sum(), h[~cross][h[~cross] >= 0].mean() if (h[~cross] >= 0).any() else -1,
    np.histogram(h[h >= 0], bins=[0, 1, 2, 4, 8, 16, 32, 64, 129, 257, 513, 1025, 2049, 3073, 100000])[0].tolist()))
bad = np.argwhere((h > 16) & cross[:, None])
print("cross-scene entries > 16:", len(bad), "by direction", np.bincount(bad[:, 1], minlength=2).tolist())
for p, dd in bad[:12]:
    print("  pair", all_pairs[p].tolist(), "dir", int(dd), "value", int(h[p, dd]))
if len(bad):
    imgs, cnt = np.unique(np.concatenate([all_pairs[bad[:, 0], 0], all_pairs[bad[:, 0], 1]]), return_counts=True)
    print("  images involved most:", sorted(zip(cnt.tolist(), imgs.tolist()), reverse=True)[:8])

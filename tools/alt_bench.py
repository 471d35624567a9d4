"""Times the matchers that are not on bench.py's headline line, on all-pairs workloads of SURVEY 8d shape:
  index   exact idx/dist kernel (knn2_l2_kernel), SIFT-like 4096 x 128-d   -> us/pair, executed int8 TOP/s
  f32     fp32-MFMA matcher, same descriptors + 0.25 (non-integer)          -> us/pair, TF/s vs 157.3 TF
  hamming XOR/popcount matcher, ORB-like 512 images x 500 x 32 B            -> us/pair, word-ops/s
  prep    desc_prep on the 512-image set                                    -> GB/s
usage: python3 tools/alt_bench.py [n_img_sift=32]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ops
from image_stitcher_amd.synthetic import overlapping_crops_descriptors

dev = torch.device("cuda", 0)
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kp, d = 4096, 128


def timed(fn, reps=3):
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best, r


des, _ = overlapping_crops_descriptors(seed=1, n_scenes=2, imgs_per_scene=n_img // 2, kp=kp, step=448, d=d)
starts = np.arange(n_img) * kp
items = np.array([(starts[i], kp, starts[j], kp) for i in range(n_img) for j in range(n_img) if i != j], np.int32)
up = len(items) / 2
ops_exec = 2.0 * kp * kp * d * len(items)

src = torch.from_numpy(des.reshape(-1, d)).to(dev)
P = ops.prepare_descriptor_tensor(src, starts, [kp] * n_img)
plan = ops.plan_items(items, dev)
dt, (c_idx, _) = timed(lambda: ops.knn2_l2_items(P, None, 0.65, plan=plan, want_matches=True))
dt_c, c_cnt = timed(lambda: ops.knn2_l2_items(P, None, 0.65, plan=plan))
assert torch.equal(c_idx, c_cnt)
print(f"index   {n_img} img: {dt*1e3:8.2f} ms  {dt/up*1e6:7.2f} us/pair  {ops_exec/dt/1e12:7.1f} TOP/s executed "
      f"(count kernel on the same items: {dt_c*1e3:.2f} ms)", flush=True)

desf = [des[i].astype(np.float32) + 0.25 for i in range(n_img)]
items_of = lambda st, ct: np.array([(st[i], ct[i], st[j], ct[j]) for i in range(n_img) for j in range(n_img) if i != j], np.int32)
dt, c_f = timed(lambda: ops.alt_item_counts(desf, items_of, "f32", 0.65, dev), reps=2)
print(f"f32     {n_img} img: {dt*1e3:8.2f} ms  {dt/up*1e6:7.2f} us/pair  {ops_exec/dt/1e12:7.1f} TF/s executed "
      f"(incl. upload of {n_img * kp * d * 4 / 1e6:.0f} MB; counts equal to the integer path: {bool(torch.equal(c_f, c_cnt))})", flush=True)

n_orb, kpo = 512, 500
rng = np.random.default_rng(5)
orb = [rng.integers(0, 256, (kpo, 32), dtype=np.uint8) for _ in range(n_orb)]
items_orb = lambda st, ct: np.array([(st[i], ct[i], st[j], ct[j]) for i in range(n_orb) for j in range(n_orb) if i != j], np.int32)
m = n_orb * (n_orb - 1)
res = {}
for impl in ("valu", "fp4"):
    dt, c_h = timed(lambda: ops.alt_item_counts(orb, items_orb, "hamming", 0.65, dev, hamming_impl=impl), reps=2)
    res[impl] = c_h
    print(f"hamming[{impl:4s}] {n_orb} img x {kpo} x 32 B: {dt*1e3:8.2f} ms  {dt/(m/2)*1e6:7.3f} us/pair "
          f"(host item build and upload included; kernel times: rocprofv3)", flush=True)
assert torch.equal(res["valu"], res["fp4"])
Po = ops.prepare_descriptors(orb, device=dev)
it_o = items_orb(Po.starts, np.asarray(Po.counts))
plan_o = ops.plan_items(it_o, dev)
dt, _ = timed(lambda: ops.knn2_l2_items(Po, None, 0.65, plan=plan_o))
print(f"l2-int8 on the same ORB set (the reference's default norm): {dt*1e3:8.2f} ms  {dt/(m/2)*1e6:7.3f} us/pair", flush=True)

big = torch.from_numpy(np.tile(des.reshape(-1, d), (512 // n_img, 1))).to(dev)
st_b = np.arange(512) * kp
dt, _ = timed(lambda: ops.prepare_descriptor_tensor(big, st_b, [kp] * 512))
byt = big.numel() * 2 + big.shape[0] * 12
print(f"prep    512 img: {dt*1e3:8.3f} ms  {byt/dt/1e9:7.1f} GB/s (read u8 + write i8 + 3 x int32 per row; includes the stats read-back)", flush=True)

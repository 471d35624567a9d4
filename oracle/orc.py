"""ctypes loader for oracle/liborc.so -- TEST INFRASTRUCTURE ONLY (see oracle/orc.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (image-stitcher_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liborc.so")

NORM_L2, NORM_HAMMING = 0, 1
MODEL_AFFINE_PARTIAL, MODEL_HOMOGRAPHY = 0, 1

_lib = None


def build(force=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".c", ".h")) or f == "Makefile"]
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.run(["make", "-C", HERE, "-B", "liborc.so"], check=True, capture_output=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.orc_ratio_filter.restype = ctypes.c_int
        _lib.orc_match_count_u8.restype = ctypes.c_int
        _lib.orc_match_count_f32.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def knn2(q, t, norm=NORM_L2, nthreads=1):
    """cv2.BFMatcher(norm).knnMatch(q, t, k=2) as SoA: idx (nq,2) int32 (-1 absent), dist (nq,2) f32."""
    q = np.ascontiguousarray(q)
    t = np.ascontiguousarray(t)
    assert q.dtype == t.dtype and q.ndim == 2 and t.ndim == 2 and q.shape[1] == t.shape[1]
    nq, d = q.shape
    nt = t.shape[0]
    idx = np.empty((nq, 2), np.int32)
    dist = np.empty((nq, 2), np.float32)
    if q.dtype == np.uint8:
        lib().orc_knn2_u8(_p(q), nq, _p(t), nt, d, norm, _p(idx), _p(dist), nthreads)
    elif q.dtype == np.float32:
        assert norm == NORM_L2
        lib().orc_knn2_f32(_p(q), nq, _p(t), nt, d, _p(idx), _p(dist), nthreads)
    else:
        raise TypeError(q.dtype)
    return idx, dist


def ratio_filter(idx, dist, ratio):
    idx = np.ascontiguousarray(idx, np.int32)
    dist = np.ascontiguousarray(dist, np.float32)
    nq = idx.shape[0]
    gq = np.empty(nq, np.int32)
    gt = np.empty(nq, np.int32)
    g = lib().orc_ratio_filter(_p(idx), _p(dist), nq, ctypes.c_double(ratio), _p(gq), _p(gt))
    return gq[:g].copy(), gt[:g].copy()


def match_count(q, t, ratio, norm=NORM_L2, nthreads=1):
    q = np.ascontiguousarray(q)
    t = np.ascontiguousarray(t)
    if t.shape[0] == 0 or q.shape[0] == 0:
        return 0
    if q.dtype == np.uint8:
        return lib().orc_match_count_u8(_p(q), q.shape[0], _p(t), t.shape[0], q.shape[1], norm,
                                        ctypes.c_double(ratio), nthreads)
    return lib().orc_match_count_f32(_p(q), q.shape[0], _p(t), t.shape[0], q.shape[1],
                                     ctypes.c_double(ratio), nthreads)


# ---------------------------------------------------------------------------------------------
# image space (oracle/orc_image.c)
# ---------------------------------------------------------------------------------------------
def _img(a):
    a = np.ascontiguousarray(a, np.uint8)
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    return a, h, w, c


def bgr2gray(img):
    a, h, w, c = _img(img)
    assert c == 3
    out = np.empty((h, w), np.uint8)
    lib().orc_bgr2gray(_p(a), h, w, _p(out))
    return out


def bbox_nonzero(img):
    """get_corners_from_image: ((x0,y0),(x1,y1)) or False"""
    a, h, w, c = _img(img)
    bb = np.zeros(4, np.int32)
    lib().orc_bbox_nonzero.restype = ctypes.c_int
    if not lib().orc_bbox_nonzero(_p(a), h, w, c, _p(bb)):
        return False
    return ((int(bb[0]), int(bb[1])), (int(bb[2]), int(bb[3])))


def pad_image(img, new_h, new_w):
    a, h, w, c = _img(img)
    out = np.empty((new_h, new_w) + ((c,) if a.ndim == 3 else ()), np.uint8)
    lib().orc_pad_image(_p(a), h, w, c, new_h, new_w, _p(out))
    return out


def get_padded_images(img, ref):
    """proc.py:295-304 (channels taken from img)."""
    new_w = ref.shape[1] + 2 * img.shape[1]
    new_h = ref.shape[0] + 2 * img.shape[0]
    return pad_image(img, new_h, new_w), pad_image(ref, new_h, new_w)


def warp(src, M, dsize_wh, perspective, arith=0):
    a, sh, sw, c = _img(src)
    dw, dh = dsize_wh
    M = np.ascontiguousarray(M, np.float64)
    assert M.shape == ((3, 3) if perspective else (2, 3))
    out = np.empty((dh, dw) + ((c,) if a.ndim == 3 else ()), np.uint8)
    lib().orc_warp(_p(a), sh, sw, c, _p(M), int(bool(perspective)), _p(out), dh, dw, arith)
    return out


def alpha_blend(warped, ref, contrib):
    a, h, w, c = _img(warped)
    b, h2, w2, c2 = _img(ref)
    assert (h, w, c) == (h2, w2, c2)
    out = np.empty_like(a)
    lib().orc_alpha_blend(_p(a), _p(b), h, w, c, ctypes.c_double(contrib), _p(out))
    return out


def warp16(src, M, dsize_wh, perspective):
    """orc_warp16: the warp on a 2-D uint16 image (opt-in depth-preserving mode)."""
    a = np.ascontiguousarray(src)
    assert a.dtype == np.uint16 and a.ndim == 2
    dw, dh = dsize_wh
    M = np.ascontiguousarray(M, np.float64)
    out = np.empty((dh, dw), np.uint16)
    lib().orc_warp16(_p(a), a.shape[0], a.shape[1], _p(M), int(bool(perspective)), _p(out), dh, dw)
    return out


def alpha_blend16(warped, ref, contrib):
    a, b = np.ascontiguousarray(warped), np.ascontiguousarray(ref)
    assert a.dtype == b.dtype == np.uint16 and a.shape == b.shape and a.ndim == 2
    out = np.empty_like(a)
    lib().orc_alpha_blend16(_p(a), _p(b), a.shape[0], a.shape[1], ctypes.c_double(contrib), _p(out))
    return out


def gradient_blend(warped, ref, feather=10):
    """Returns the modified copy of ref (the reference mutates its argument)."""
    a, h, w, c = _img(warped)
    b = np.array(ref, dtype=np.uint8, order="C", copy=True)
    assert b.shape == a.shape
    lib().orc_gradient_blend(_p(a), _p(b), h, w, c, feather)
    return b


# ---------------------------------------------------------------------------------------------
# robust transform (oracle/orc_ransac.c)
# ---------------------------------------------------------------------------------------------
def _pts(p):
    return np.ascontiguousarray(np.asarray(p, np.float32).reshape(-1, 2))


def _msz(model):
    return 9 if model == MODEL_HOMOGRAPHY else 6


def _mshape(model):
    return (3, 3) if model == MODEL_HOMOGRAPHY else (2, 3)


def ransac_sample(model, src, dst, max_iters=2000):
    s, d = _pts(src), _pts(dst)
    mp = 4 if model == MODEL_HOMOGRAPHY else 2
    out = np.zeros((max_iters, mp), np.int32)
    lib().orc_ransac_sample.restype = ctypes.c_int
    k = lib().orc_ransac_sample(model, _p(s), _p(d), len(s), max_iters, _p(out))
    return out[:k]


def fit_model(model, src, dst):
    s, d = _pts(src), _pts(dst)
    M = np.zeros(_msz(model), np.float64)
    lib().orc_fit_model.restype = ctypes.c_int
    ok = lib().orc_fit_model(model, _p(s), _p(d), len(s), _p(M))
    return M.reshape(_mshape(model)) if ok else None


def reproj_err(model, M, src, dst):
    s, d = _pts(src), _pts(dst)
    M = np.ascontiguousarray(M, np.float64)
    err = np.empty(len(s), np.float32)
    lib().orc_reproj_err(model, _p(M), _p(s), _p(d), len(s), _p(err))
    return err


def score(model, M, src, dst, thr):
    s, d = _pts(src), _pts(dst)
    M = np.ascontiguousarray(M, np.float64)
    mask = np.empty(len(s), np.uint8)
    lib().orc_score.restype = ctypes.c_int
    n = lib().orc_score(model, _p(M), _p(s), _p(d), len(s), ctypes.c_double(thr), _p(mask))
    return n, mask


def find_homography(src, dst, thr=5.0, max_iters=2000, confidence=0.995):
    """cv2.findHomography(src, dst, cv2.RANSAC, 5.0) -> (M 3x3 | None, mask (n,1) uint8)"""
    s, d = _pts(src), _pts(dst)
    M = np.zeros(9, np.float64)
    mask = np.zeros(len(s), np.uint8)
    lib().orc_find_homography_ransac.restype = ctypes.c_int
    ok = lib().orc_find_homography_ransac(_p(s), _p(d), len(s), ctypes.c_double(thr), max_iters,
                                          ctypes.c_double(confidence), _p(M), _p(mask))
    return (M.reshape(3, 3) if ok else None), mask.reshape(-1, 1)


def estimate_affine_partial(src, dst, thr=3.0, max_iters=2000, confidence=0.99, refine_iters=10):
    """cv2.estimateAffinePartial2D(src, dst) -> (M 2x3 | None, inliers (n,1) uint8)"""
    s, d = _pts(src), _pts(dst)
    M = np.zeros(6, np.float64)
    mask = np.zeros(len(s), np.uint8)
    lib().orc_estimate_affine_partial.restype = ctypes.c_int
    ok = lib().orc_estimate_affine_partial(_p(s), _p(d), len(s), ctypes.c_double(thr), max_iters,
                                           ctypes.c_double(confidence), refine_iters, _p(M), _p(mask))
    return (M.reshape(2, 3) if ok else None), mask.reshape(-1, 1)


def ransac_trace(model, src, dst, thr, max_iters=2000, confidence=None):
    """Every hypothesis the sequential RANSAC tries, in order: (hyps (k,msz), counts (k,), valid (k,), best_iter)."""
    if confidence is None:
        confidence = 0.995 if model == MODEL_HOMOGRAPHY else 0.99
    s, d = _pts(src), _pts(dst)
    hyp = np.zeros((max_iters, _msz(model)), np.float64)
    cnt = np.zeros(max_iters, np.int32)
    valid = np.zeros(max_iters, np.int32)
    n_iter = ctypes.c_int(0)
    best = ctypes.c_int(-1)
    lib().orc_ransac_trace.restype = ctypes.c_int
    ok = lib().orc_ransac_trace(model, _p(s), _p(d), len(s), ctypes.c_double(thr), max_iters,
                                ctypes.c_double(confidence), _p(hyp), _p(cnt), _p(valid),
                                ctypes.byref(n_iter), ctypes.byref(best))
    k = n_iter.value
    return bool(ok), hyp[:k], cnt[:k], valid[:k], best.value


def refine(model, M, src_inl, dst_inl, refit, max_iters=10):
    s, d = _pts(src_inl), _pts(dst_inl)
    M = np.array(M, np.float64).reshape(-1).copy()
    lib().orc_refine(model, _p(s), _p(d), len(s), int(refit), max_iters, _p(M))
    return M.reshape(_mshape(model))

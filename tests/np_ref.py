"""Small pure-numpy/Python restatements of the reference's image helpers (utils/proc.py) used ONLY to
cross-check the C oracle on tiny inputs.  Written from the reference's semantics, with each cv2 call
replaced by its documented arithmetic:
    cv2.cvtColor(BGR2GRAY)      -> (B*3735 + G*19235 + R*9798 + 16384) >> 15        [OCV-K]
    cv2.threshold(g,10,255,BIN) -> g > 10
    cv2.findNonZero/boundingRect-> min/max of the non-zero coordinates
"""
import numpy as np


def gray(img):
    if img.ndim == 2:
        return img
    b, g, r = (img[..., k].astype(np.int64) for k in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)


def corners(img):                                    # proc.py:34-45
    ys, xs = np.nonzero(gray(img))
    if len(xs) == 0:
        return False
    return ((int(xs.min()), int(ys.min())), (int(xs.max()) + 1, int(ys.max()) + 1))


def alpha_merge(warped, ref, contrib):               # proc.py:48-82
    ov = (gray(warped) > 10) & (gray(ref) > 10)
    r2 = ref.copy()
    w2 = warped.copy()
    r2[ov] = contrib * ref[ov]                       # float64 -> uint8 store (truncation)
    w2[ov] = (1 - contrib) * warped[ov]
    return w2 + r2                                   # uint8 wrap-around


def gradient_merge(warped, ref, feather=10):         # proc.py:192-254
    ref = ref.copy()
    cs = corners(warped)
    if cs is False:
        return ref
    (x0, y0), (x1, y1) = cs
    x0, y0 = max(0, x0 - 50), max(0, y0 - 50)
    x1, y1 = min(ref.shape[1], x1 + 50), min(ref.shape[0], y1 + 50)
    r1 = warped[y0:y1, x0:x1]
    r2 = ref[y0:y1, x0:x1]
    mask = np.zeros(r1.shape, np.float64)
    mask[r1 > 0] = 1
    ch0 = (lambda m: m) if mask.ndim == 2 else (lambda m: m[..., 0])
    for y in range(mask.shape[0]):
        nz = np.nonzero(ch0(mask)[y] > 0)[0]
        if len(nz):
            lo, hi = nz[0], nz[-1]
            for x in range(lo, min(lo + feather, mask.shape[1])):
                mask[y, x] *= (x - lo) / feather
            for x in range(max(0, hi - feather), hi + 1):
                mask[y, x] *= (hi - x) / feather
    for x in range(mask.shape[1]):
        nz = np.nonzero(ch0(mask)[:, x] > 0)[0]
        if len(nz):
            lo, hi = nz[0], nz[-1]
            for y in range(lo, min(lo + feather, mask.shape[0])):
                mask[y, x] *= (y - lo) / feather
            for y in range(max(0, hi - feather), hi + 1):
                mask[y, x] *= (hi - y) / feather
    ref[y0:y1, x0:x1] = (r1 * mask + r2 * (1 - mask)).astype(np.uint8)
    return ref


def pad(img, new_h, new_w):                          # proc.py:280-292
    out = np.zeros((new_h, new_w) + img.shape[2:], np.uint8)
    xc, yc = (new_w - img.shape[1]) // 2, (new_h - img.shape[0]) // 2
    out[yc:yc + img.shape[0], xc:xc + img.shape[1]] = img
    return out


def warp_bilinear_f64(src, M_inv, dsize_wh):
    """Unquantised bilinear inverse warp in float64 (no rounding subtleties): a +-1 LSB yardstick."""
    dw, dh = dsize_wh
    ys, xs = np.mgrid[0:dh, 0:dw].astype(np.float64)
    if M_inv.shape[0] == 3:
        wv = M_inv[2, 0] * xs + M_inv[2, 1] * ys + M_inv[2, 2]
        sx = (M_inv[0, 0] * xs + M_inv[0, 1] * ys + M_inv[0, 2]) / wv
        sy = (M_inv[1, 0] * xs + M_inv[1, 1] * ys + M_inv[1, 2]) / wv
    else:
        sx = M_inv[0, 0] * xs + M_inv[0, 1] * ys + M_inv[0, 2]
        sy = M_inv[1, 0] * xs + M_inv[1, 1] * ys + M_inv[1, 2]
    ix, iy = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    a, b = sx - ix, sy - iy
    sh, sw = src.shape[:2]
    s = src.astype(np.float64).reshape(sh, sw, -1)

    def at(yy, xx):
        ok = (xx >= 0) & (xx < sw) & (yy >= 0) & (yy < sh)
        v = s[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)]
        return v * ok[..., None]
    a, b = a[..., None], b[..., None]
    v = (at(iy, ix) * (1 - a) + at(iy, ix + 1) * a) * (1 - b) + (at(iy + 1, ix) * (1 - a) + at(iy + 1, ix + 1) * a) * b
    return v.reshape((dh, dw) + src.shape[2:])

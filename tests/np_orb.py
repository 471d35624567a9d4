"""numpy DEFINITION of the on-GPU ORB-style detector (image_stitcher_amd.detect.HipOrb, csrc/isx_orb.hip) -- TEST
INFRASTRUCTURE.  SURVEY 8f-4 / reference stitcher.py:92-100, :163, :193-194, :263 (`cv2.ORB_create().detectAndCompute`).

This is NOT cv2.ORB: cv2 is absent here and its learned 256-pair pattern, pyramid resampling and float Harris / angle
arithmetic cannot be restated from memory, so parity with cv2.ORB is unpinned BY CONSTRUCTION.  What is defined here is
a detector of ORB's structure -- FAST-9 corners, Harris ranking, intensity-centroid orientation, steered BRIEF on a
blurred image, single scale -- in integer arithmetic only, so that the HIP kernels can be checked bit for bit:

  gray      (B*3735 + G*19235 + R*9798 + 16384) >> 15                      (the library's BGR2GRAY)
  blur      separable 7 taps [18,34,49,54,49,34,18]/256, replicated border; one rounding: (sum + 32768) >> 16
  corners   FAST-9, threshold t, on the 16-pixel radius-3 circle, for pixels at least E = 16 from the border
  response  R = 25 (a c - b^2) - (a + c)^2 over a 7x7 block of 3x3 Sobel products (Harris with k = 1/25), int64; kept
            if R > 0 and no gray == 0 pixel lies within `clean` pixels (Chebyshev); key = bits of float32(R)
  NMS       radius 3 on the key, ties to the lower linear index
  select    the nfeatures largest keys, ties to the lower linear index, in that order
  angle     patch moments m10, m01 over the radius-15 disc of the UNBLURRED gray image; bin k of 32 = argmax of the
            integer dot products with DIRS[k] (lowest k on ties)
  bits      pattern[k] (the base pairs rotated by k * 11.25 degrees, rounded once on the host) on the blurred image:
            bit = blur[p1] < blur[p2], packed MSB first (np.packbits)
"""
import numpy as np

E = 16
BLUR_W = np.array([18, 34, 49, 54, 49, 34, 18], np.int64)
CIRCLE = [(0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3), (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1),
          (-2, -2), (-1, -3)]
DIRS = np.array([[16384, 0], [16069, 3196], [15137, 6270], [13623, 9102], [11585, 11585], [9102, 13623], [6270, 15137],
                 [3196, 16069], [0, 16384], [-3196, 16069], [-6270, 15137], [-9102, 13623], [-11585, 11585], [-13623, 9102],
                 [-15137, 6270], [-16069, 3196], [-16384, 0], [-16069, -3196], [-15137, -6270], [-13623, -9102],
                 [-11585, -11585], [-9102, -13623], [-6270, -15137], [-3196, -16069], [0, -16384], [3196, -16069],
                 [6270, -15137], [9102, -13623], [11585, -11585], [13623, -9102], [15137, -6270], [16069, -3196]], np.int64)
UMAX = [15, 14, 14, 14, 14, 14, 13, 13, 12, 12, 11, 10, 9, 7, 5, 0]


def gray(img):
    img = np.asarray(img)
    if img.ndim == 2:
        return img.astype(np.uint8)
    b, g, r = (img[..., k].astype(np.int64) for k in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)


def blur(g):
    h, w = g.shape
    p = np.pad(g.astype(np.int64), ((0, 0), (3, 3)), mode="edge")
    hs = sum(BLUR_W[i] * p[:, i:i + w] for i in range(7))
    p = np.pad(hs, ((3, 3), (0, 0)), mode="edge")
    return ((sum(BLUR_W[j] * p[j:j + h] for j in range(7)) + 32768) >> 16).astype(np.uint8)


def _shift(a, dx, dy, fill=0):
    """out[y, x] = a[y + dy, x + dx], `fill` outside"""
    h, w = a.shape
    out = np.full_like(a, fill)
    ys0, ys1 = max(0, -dy), min(h, h - dy)
    xs0, xs1 = max(0, -dx), min(w, w - dx)
    if ys1 > ys0 and xs1 > xs0:
        out[ys0:ys1, xs0:xs1] = a[ys0 + dy:ys1 + dy, xs0 + dx:xs1 + dx]
    return out


def fast9(g, t):
    gi = g.astype(np.int32)
    br = np.zeros(g.shape, np.uint32)
    dk = np.zeros(g.shape, np.uint32)
    for i, (dx, dy) in enumerate(CIRCLE):
        q = _shift(gi, dx, dy)
        br |= (q > gi + t).astype(np.uint32) << i
        dk |= (q < gi - t).astype(np.uint32) << i

    def run9(m):
        x = m | (m << 16)
        x &= x >> 1
        x &= x >> 2
        x &= x >> 4
        x &= x >> 1
        return x != 0
    c = run9(br) | run9(dk)
    valid = np.zeros(g.shape, bool)
    valid[E:g.shape[0] - E, E:g.shape[1] - E] = True
    return c & valid


def harris(g):
    gi = g.astype(np.int64)
    ix = (_shift(gi, 1, -1) + 2 * _shift(gi, 1, 0) + _shift(gi, 1, 1)) - (_shift(gi, -1, -1) + 2 * _shift(gi, -1, 0) + _shift(gi, -1, 1))
    iy = (_shift(gi, -1, 1) + 2 * _shift(gi, 0, 1) + _shift(gi, 1, 1)) - (_shift(gi, -1, -1) + 2 * _shift(gi, 0, -1) + _shift(gi, 1, -1))

    def box7(a):
        return sum(_shift(a, dx, dy) for dy in range(-3, 4) for dx in range(-3, 4))
    a, b, c = box7(ix * ix), box7(ix * iy), box7(iy * iy)
    return 25 * (a * c - b * b) - (a + c) ** 2


def keymap(g, t, clean):
    from scipy import ndimage
    R = harris(g)
    ok = fast9(g, t) & (R > 0)
    if clean > 0:
        near_zero = ndimage.maximum_filter((g == 0).astype(np.uint8), size=2 * clean + 1, mode="constant", cval=1)
        ok &= near_zero == 0
    key = np.where(ok, R, 0).astype(np.float32).view(np.uint32)
    return np.where(ok, key, 0).astype(np.uint32)


def nms(key):
    h, w = key.shape
    idx = np.arange(h * w, dtype=np.int64).reshape(h, w)
    keep = key > 0
    for dy in range(-3, 4):
        for dx in range(-3, 4):
            if dx == 0 and dy == 0:
                continue
            kq, iq = _shift(key, dx, dy), _shift(idx, dx, dy, fill=-1)
            keep &= (kq < key) | ((kq == key) & (iq > idx)) | (iq < 0)
    return keep


def base_pattern(seed=11):
    """256 point pairs (x1, y1, x2, y2), isotropic Gaussian of sigma 31/5 clipped to the radius-13 disc."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pts = []
    while len(pts) < 512:
        p = np.rint(rng.normal(0, 31 / 5.0, 2))
        if p[0] * p[0] + p[1] * p[1] <= 13 * 13:
            pts.append(p)
    return np.array(pts, np.int64).reshape(256, 4)


def rotated_patterns(seed=11):
    """int8 [32][256][4]: the base pattern rotated by k * 11.25 degrees (float64 on the host, rounded once)."""
    base = base_pattern(seed).astype(np.float64)
    out = np.zeros((32, 256, 4), np.int8)
    for k in range(32):
        c, s = np.cos(k * 2 * np.pi / 32), np.sin(k * 2 * np.pi / 32)
        for j in (0, 2):
            x, y = base[:, j], base[:, j + 1]
            out[k, :, j] = np.rint(x * c - y * s)
            out[k, :, j + 1] = np.rint(x * s + y * c)
    assert np.abs(out.astype(int)).max() <= 14
    return out


_DISC = np.array([(u, v) for v in range(-15, 16) for u in range(-UMAX[abs(v)], UMAX[abs(v)] + 1)], np.int64)


def detect(img, nfeatures=500, fast_threshold=20, clean=16, patterns=None):
    """-> (xy (n,2) int, response float32 (n,), angle_bin (n,), des (n,32) uint8) in selection order"""
    g = gray(img)
    h, w = g.shape
    if h < 2 * E + 1 or w < 2 * E + 1:
        z = np.zeros(0, np.int64)
        return np.zeros((0, 2), np.int64), np.zeros(0, np.float32), z, np.zeros((0, 32), np.uint8)
    key = keymap(g, fast_threshold, clean)
    ys, xs = np.nonzero(nms(key))
    k = key[ys, xs]
    order = np.argsort(((~k).astype(np.uint64) << np.uint64(32)) | (ys * w + xs).astype(np.uint64), kind="stable")[:nfeatures]
    ys, xs, k = ys[order], xs[order], k[order]
    bl = blur(g)
    pat = rotated_patterns() if patterns is None else patterns
    gi = g.astype(np.int64)
    vals = gi[ys[:, None] + _DISC[None, :, 1], xs[:, None] + _DISC[None, :, 0]]
    m10 = (vals * _DISC[None, :, 0]).sum(1)
    m01 = (vals * _DISC[None, :, 1]).sum(1)
    dots = m10[:, None] * DIRS[None, :, 0] + m01[:, None] * DIRS[None, :, 1]
    ang = np.argmax(dots, axis=1)          # first maximum = lowest k on ties
    p = pat[ang].astype(np.int64)          # (n, 256, 4)
    a = bl[ys[:, None] + p[:, :, 1], xs[:, None] + p[:, :, 0]]
    b = bl[ys[:, None] + p[:, :, 3], xs[:, None] + p[:, :, 2]]
    des = np.packbits(a < b, axis=1)
    return np.stack([xs, ys], 1), k.view(np.float32), ang, des

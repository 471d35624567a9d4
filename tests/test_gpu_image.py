"""GPU parity of the image-space kernels (through the C ABI) vs the CPU oracle -- bit exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rand_img(rng, h, w, c, lo=0, hi=256):
    return rng.integers(lo, hi, (h, w) if c == 1 else (h, w, c), dtype=np.uint8)


@pytest.mark.parametrize("c", [1, 3])
def test_bbox(cuda, orc, c):
    from image_stitcher_amd import ops
    rng = np.random.default_rng(c)
    for (h, w, y0, y1, x0, x1) in [(40, 60, 5, 22, 7, 33), (300, 1000, 0, 300, 0, 1000), (257, 513, 256, 257, 512, 513),
                                   (64, 64, 10, 11, 3, 60)]:
        im = np.zeros((h, w) + ((3,) if c == 3 else ()), np.uint8)
        im[y0:y1, x0:x1] = rand_img(rng, y1 - y0, x1 - x0, c, 0, 40)
        got = ops.bbox_to_corners(ops.bbox_nonzero(dev(im)).cpu().numpy())
        assert got == orc.bbox_nonzero(im)
    z = np.zeros((33, 70, 3), np.uint8)
    z[3, 4] = (4, 0, 0)   # gray == 0 under the 8-bit formula
    assert ops.bbox_to_corners(ops.bbox_nonzero(dev(z)).cpu().numpy()) is False
    # strided view (a crop of a bigger device image)
    big = rand_img(rng, 100, 120, c, 1, 256)
    view = dev(big)[10:50, 20:90]
    assert ops.bbox_to_corners(ops.bbox_nonzero(view).cpu().numpy()) == ((0, 0), (70, 40))


def _texture(rng, h, w, c):
    base = rng.integers(12, 256, (h // 8 + 2, w // 8 + 2, c)).astype(np.float64)
    img = np.kron(base, np.ones((8, 8, 1)))[:h, :w]
    img = 0.5 * img + 0.5 * np.roll(img, 3, 1)
    img = img.astype(np.uint8)
    return img[..., 0] if c == 1 else img


@pytest.mark.parametrize("c", [1, 3])
@pytest.mark.parametrize("persp", [False, True])
def test_warp_padded_canvas_parity(cuda, orc, c, persp):
    """stitcher.py:178 + :216: pad frame into the (ref+2*img) canvas, warp the padded frame to canvas size.
    The HIP kernel samples the un-padded frame at its canvas offset; the oracle warps the padded canvas."""
    from image_stitcher_amd import ops
    rng = np.random.default_rng(10 + c + 2 * persp)
    img = _texture(rng, 96, 128, c)
    ref = _texture(rng, 150, 210, c)
    pimg, pref = orc.get_padded_images(img, ref)
    H, W = pref.shape[:2]
    ox, oy = (ref.shape[1] + img.shape[1]) // 2, (ref.shape[0] + img.shape[0]) // 2
    for trial in range(3):
        th = np.deg2rad(rng.uniform(-8, 8))
        s = rng.uniform(0.9, 1.1)
        M = np.array([[s * np.cos(th), -s * np.sin(th), rng.uniform(-60, 60)],
                      [s * np.sin(th), s * np.cos(th), rng.uniform(-40, 40)], [0, 0, 1]])
        # rotate about the frame centre so that the frame stays mostly inside the canvas
        cx, cy = ox + img.shape[1] / 2, oy + img.shape[0] / 2
        M[0, 2] += cx - (M[0, 0] * cx + M[0, 1] * cy)
        M[1, 2] += cy - (M[1, 0] * cx + M[1, 1] * cy)
        if persp:
            M[2, :2] = rng.uniform(-1e-4, 1e-4, 2)
        Mm = M if persp else M[:2]
        exp = orc.warp(pimg, Mm, (W, H), persp)
        bb = ops.new_bbox(cuda)
        got = ops.warp(dev(img), Mm, persp, (0, 0, W, H), src_xy=(ox, oy), bbox=bb).cpu().numpy()
        assert np.array_equal(got, exp), f"{(got != exp).sum()} pixels differ, max {np.abs(got.astype(int) - exp).max()}"
        assert ops.bbox_to_corners(bb.cpu().numpy()) == orc.bbox_nonzero(exp)
        # sub-rectangle of the canvas only
        x0, y0, dw, dh = 37, 51, 201, 133
        sub = ops.warp(dev(img), Mm, persp, (x0, y0, dw, dh), src_xy=(ox, oy)).cpu().numpy()
        assert np.array_equal(sub, exp[y0:y0 + dh, x0:x0 + dw])


@pytest.mark.parametrize("c", [1, 3])
@pytest.mark.parametrize("persp", [False, True])
def test_warp_classic_arithmetic_parity(cuda, orc, c, persp):
    """arith="classic": the fixed-point remap of OpenCV <= 4.10 (oracle arith=1), bit exact, on the full
    canvas and on a sub-rectangle that does not start on one of the invoker's 64-column blocks."""
    from image_stitcher_amd import ops
    rng = np.random.default_rng(40 + c + 2 * persp)
    img = _texture(rng, 96, 128, c)
    ref = _texture(rng, 150, 210, c)
    pimg, pref = orc.get_padded_images(img, ref)
    H, W = pref.shape[:2]
    ox, oy = (ref.shape[1] + img.shape[1]) // 2, (ref.shape[0] + img.shape[0]) // 2
    for trial in range(3):
        th = np.deg2rad(rng.uniform(-8, 8))
        s = rng.uniform(0.9, 1.1)
        M = np.array([[s * np.cos(th), -s * np.sin(th), rng.uniform(-60, 60)],
                      [s * np.sin(th), s * np.cos(th), rng.uniform(-40, 40)], [0, 0, 1]])
        cx, cy = ox + img.shape[1] / 2, oy + img.shape[0] / 2
        M[0, 2] += cx - (M[0, 0] * cx + M[0, 1] * cy)
        M[1, 2] += cy - (M[1, 0] * cx + M[1, 1] * cy)
        if persp:
            M[2, :2] = rng.uniform(-1e-4, 1e-4, 2)
        Mm = M if persp else M[:2]
        exp = orc.warp(pimg, Mm, (W, H), persp, arith=1)
        assert (exp != orc.warp(pimg, Mm, (W, H), persp, arith=0)).any()     # the two arithmetics do differ
        bb = ops.new_bbox(cuda)
        got = ops.warp(dev(img), Mm, persp, (0, 0, W, H), src_xy=(ox, oy), bbox=bb, arith="classic").cpu().numpy()
        assert np.array_equal(got, exp), f"{(got != exp).sum()} pixels differ, max {np.abs(got.astype(int) - exp).max()}"
        assert ops.bbox_to_corners(bb.cpu().numpy()) == orc.bbox_nonzero(exp)
        x0, y0, dw, dh = 37, 51, 201, 133
        sub = ops.warp(dev(img), Mm, persp, (x0, y0, dw, dh), src_xy=(ox, oy), arith="classic", canvas=(W, H)).cpu().numpy()
        assert np.array_equal(sub, exp[y0:y0 + dh, x0:x0 + dw])
    # exact-copy cases
    eye = np.eye(3)
    got = ops.warp(dev(img), eye if persp else eye[:2], persp, (0, 0, img.shape[1], img.shape[0]), arith="classic")
    assert np.array_equal(got.cpu().numpy(), img)


def test_warp_identity_is_copy(cuda):
    from image_stitcher_amd import ops
    rng = np.random.default_rng(5)
    img = rand_img(rng, 50, 70, 3)
    eye = np.eye(3)
    out = ops.warp(dev(img), eye[:2], False, (0, 0, 70, 50)).cpu().numpy()
    assert np.array_equal(out, img)
    out = ops.warp(dev(img), eye, True, (0, 0, 70, 50)).cpu().numpy()
    assert np.array_equal(out, img)


@pytest.mark.parametrize("c", [1, 3])
@pytest.mark.parametrize("contrib", [0.5, 0.2])
def test_alpha_blend_parity(cuda, orc, c, contrib):
    from image_stitcher_amd import ops
    rng = np.random.default_rng(20 + c)
    w = rand_img(rng, 131, 257, c)
    r = rand_img(rng, 131, 257, c)
    w[:40] = 0
    r[:, :30] = rng.integers(0, 12, r[:, :30].shape)
    exp = orc.alpha_blend(w, r, contrib)
    got = ops.alpha_blend(dev(w), dev(r), contrib).cpu().numpy()
    assert np.array_equal(got, exp)
    rd = dev(r)
    ops.alpha_blend(dev(w), rd, contrib, out=rd)  # in place on ref
    assert np.array_equal(rd.cpu().numpy(), exp)


@pytest.mark.parametrize("c", [1, 3])
def test_gradient_blend_parity(cuda, orc, c):
    """Whole-canvas oracle (proc.py:192-254) vs HIP ROI kernel on the ROI the host cuts (proc.py:199-205)."""
    from image_stitcher_amd import ops
    rng = np.random.default_rng(30 + c)
    H, W = 420, 610
    for trial in range(4):
        warped = np.zeros((H, W) + ((3,) if c == 3 else ()), np.uint8)
        y0, x0 = rng.integers(0, 150, 2)
        hh, ww = rng.integers(60, 250, 2)
        blk = rand_img(rng, hh, ww, c)
        blk[rng.random((hh, ww)) < 0.03] = 0
        warped[y0:y0 + hh, x0:x0 + ww] = blk
        for yy in range(y0, y0 + hh):
            warped[yy, x0:x0 + (yy - y0) // 4] = 0
            warped[yy, x0 + ww - (y0 + hh - yy) // 5:x0 + ww] = 0
        ref = rand_img(rng, H, W, c)
        exp = orc.gradient_blend(warped, ref)
        (bx0, by0), (bx1, by1) = orc.bbox_nonzero(warped)
        rx0, ry0 = max(0, bx0 - 50), max(0, by0 - 50)
        rx1, ry1 = min(W, bx1 + 50), min(H, by1 + 50)
        wd, rd = dev(warped), dev(ref)
        ops.gradient_blend_roi(wd[ry0:ry1, rx0:rx1], rd[ry0:ry1, rx0:rx1])
        got = rd.cpu().numpy()
        assert np.array_equal(got, exp), f"{(got != exp).sum()} differ"


def test_large_images_exceed_the_capped_grids(cuda, orc):
    """Images with more tiles than the capped (strided) grids of warp_kernel / bbox_nonzero_kernel and an
    odd-offset view: exercises the tile loop, the per-block bbox table and the finish kernel."""
    from image_stitcher_amd import ops
    rng = np.random.default_rng(77)
    big = np.zeros((3011, 4003, 3), np.uint8)
    big[501:2777, 333:3901] = rand_img(rng, 2276, 3568, 3, 0, 256)
    d = dev(big)
    assert ops.bbox_to_corners(ops.bbox_nonzero(d).cpu().numpy()) == orc.bbox_nonzero(big)
    view = d[7:3000, 13:3999]
    assert ops.bbox_to_corners(ops.bbox_nonzero(view).cpu().numpy()) == orc.bbox_nonzero(np.ascontiguousarray(big[7:3000, 13:3999]))
    # warp of a 1-channel frame into a large canvas rectangle (> 2048 tiles of 256 x 4)
    img = _texture(rng, 700, 900, 1)
    H, W = 2400, 3100
    canvas = np.zeros((H, W), np.uint8)
    ox, oy = 1001, 803
    canvas[oy:oy + 700, ox:ox + 900] = img
    th = np.deg2rad(3.0)
    M = np.array([[1.05 * np.cos(th), -1.05 * np.sin(th), 40.5], [1.05 * np.sin(th), 1.05 * np.cos(th), -33.25]])
    exp = orc.warp(canvas, M, (W, H), False)
    bb = ops.new_bbox(cuda)
    got = ops.warp(dev(img), M, False, (0, 0, W, H), src_xy=(ox, oy), bbox=bb).cpu().numpy()
    assert np.array_equal(got, exp)
    assert ops.bbox_to_corners(bb.cpu().numpy()) == orc.bbox_nonzero(exp)


def test_mosaic_larger_than_4_gib(cuda, orc):
    """A resident mosaic of more than 2^32 bytes (large aerial mosaics, BASELINE configs[4]): row offsets inside
    the kernels are 64-bit.  Known answers: planted pixels for the bounding box, an integer-translation warp
    (an exact copy in both INTER_LINEAR arithmetics) fused with the alpha blend into the far corner, and a copy."""
    import torch
    from image_stitcher_amd import ops
    H, W = 36100, 40004
    big = torch.zeros((H, W, 3), dtype=torch.uint8, device=cuda)
    assert big.numel() > (1 << 32)
    big[5, 7, 1] = 9
    big[H - 3, W - 2, 2] = 200
    assert ops.bbox_to_corners(ops.bbox_nonzero(big).cpu().numpy()) == ((7, 5), (W - 1, H - 2))
    rng = np.random.default_rng(5)
    img = rand_img(rng, 301, 403, 3, 12, 256)
    x0, y0 = W - 500, H - 400                      # canvas position of the frame after the warp
    M = np.array([[1.0, 0.0, float(x0 - 100)], [0.0, 1.0, float(y0 - 50)]])
    region = big[y0 - 20:y0 + 330, x0 - 30:x0 + 430]
    region[:] = 77
    bb = ops.new_bbox(cuda)
    ops.warp_alpha_blend(dev(img), M, False, (x0 - 30, y0 - 20, 460, 350), region, 0.25, src_xy=(100, 50), bbox=bb,
                         canvas=(W, H))
    exp = np.full((350, 460, 3), 77, np.uint8)
    exp[20:321, 30:433] = (0.25 * 77) // 1 + np.trunc(0.75 * img.astype(np.float64)).astype(np.uint8)
    assert np.array_equal(big[y0 - 20:y0 + 330, x0 - 30:x0 + 430].cpu().numpy(), exp)
    assert ops.bbox_to_corners(bb.cpu().numpy()) == ((x0, y0), (x0 + 403, y0 + 301))
    out = torch.empty((350, 460, 3), dtype=torch.uint8, device=cuda)
    ops.copy2d(big[y0 - 20:y0 + 330, x0 - 30:x0 + 430], out)
    assert np.array_equal(out.cpu().numpy(), exp)
    mb = ops.bbox_nonzero(big).cpu().numpy()
    assert ops.bbox_to_corners(mb) == ((7, 5), (W - 1, H - 2))


@pytest.mark.parametrize("c", [1, 3])
@pytest.mark.parametrize("persp,arith", [(False, "f32"), (True, "f32"), (False, "classic"), (True, "classic")])
def test_fused_warp_alpha_blend(cuda, orc, c, persp, arith):
    """isx_warp_alpha_blend == isx_warp followed by isx_alpha_blend == the oracle's warp + seamless_merge, bit
    exact, in place on a strided view of the canvas, with the warped image's bbox."""
    from image_stitcher_amd import ops
    rng = np.random.default_rng(70 + c + 2 * persp)
    img = _texture(rng, 96, 128, c)
    ref = _texture(rng, 150, 210, c)
    ref[:40] = rng.integers(0, 14, ref[:40].shape, dtype=np.uint8)     # dark-but-nonzero pixels: the wrap-around cases
    pimg, pref = orc.get_padded_images(img, ref)
    H, W = pref.shape[:2]
    ox, oy = (ref.shape[1] + img.shape[1]) // 2, (ref.shape[0] + img.shape[0]) // 2
    th = np.deg2rad(5.0)
    M = np.array([[np.cos(th), -np.sin(th), 31.0], [np.sin(th), np.cos(th), -12.5], [0, 0, 1]])
    cx, cy = ox + img.shape[1] / 2, oy + img.shape[0] / 2
    M[0, 2] += cx - (M[0, 0] * cx + M[0, 1] * cy)
    M[1, 2] += cy - (M[1, 0] * cx + M[1, 1] * cy)
    if persp:
        M[2, :2] = (6e-5, -4e-5)
    Mm = M if persp else M[:2]
    warped = orc.warp(pimg, Mm, (W, H), persp, arith=1 if arith == "classic" else 0)
    for contrib in (0.2, 0.5):
        exp = orc.alpha_blend(warped, pref, contrib)
        canvas = dev(pref)
        (bx0, by0), (bx1, by1) = orc.bbox_nonzero(warped)
        x0, y0, dw, dh = bx0 - 3, by0 - 2, bx1 - bx0 + 8, by1 - by0 + 5      # a rectangle that contains the warped frame
        bb = ops.new_bbox(cuda)
        view = canvas[y0:y0 + dh, x0:x0 + dw]
        ops.warp_alpha_blend(dev(img), Mm, persp, (x0, y0, dw, dh), view, contrib, src_xy=(ox, oy), bbox=bb, arith=arith,
                             canvas=(W, H))
        got = canvas.cpu().numpy()
        # outside the rectangle warped is 0: blend(0, ref) = ref there (no overlap, 0 + ref), so the whole canvas matches
        assert np.array_equal(got, exp), f"{(got != exp).sum()} pixels differ"
        assert ops.bbox_to_corners(bb.cpu().numpy()) == orc.bbox_nonzero(warped)


def test_bbox_calls_overlap_on_two_streams(cuda, orc):
    """Every bbox accumulator carries its own partial table (include/isx.h ISX_BBOX_INTS), so bbox-producing
    calls on different accumulators may be in flight on one GPU at once: many rounds of two different images
    reduced concurrently on two streams give each image its own box."""
    import torch
    from image_stitcher_amd import ops
    rng = np.random.default_rng(5)
    a = np.zeros((2100, 3900, 3), np.uint8)
    b = np.zeros((1900, 4100, 3), np.uint8)
    a[300:1500, 700:3000] = rng.integers(1, 255, (1200, 2300, 3), dtype=np.uint8)
    b[40:1800, 90:4000] = rng.integers(1, 255, (1760, 3910, 3), dtype=np.uint8)
    da, db = dev(a), dev(b)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    acc1, acc2 = ops.new_bbox(cuda), ops.new_bbox(cuda)
    torch.cuda.synchronize()
    res = []
    for _ in range(20):
        with torch.cuda.stream(s1):
            r1 = ops.bbox_nonzero(da, acc=acc1)
        with torch.cuda.stream(s2):
            r2 = ops.bbox_nonzero(db, acc=acc2)
        torch.cuda.synchronize()
        res.append((ops.bbox_to_corners(r1.cpu().numpy()), ops.bbox_to_corners(r2.cpu().numpy())))
    assert all(r == (orc.bbox_nonzero(a), orc.bbox_nonzero(b)) for r in res)


def test_clear_rects_matches_numpy_slices(cuda):
    """isx_clear_rects: up to four rectangles of a pitched view zeroed in one launch, byte-exact at unaligned edges."""
    import torch
    from image_stitcher_amd import ops
    rng = np.random.default_rng(3)
    for c in (1, 3):
        shape = (97, 211) if c == 1 else (97, 211, 3)
        a = rng.integers(1, 255, shape, dtype=np.uint8)
        t = torch.from_numpy(a).cuda()
        view, ref = t[5:90, 7:200], a[5:90, 7:200].copy()
        rects = [(0, 3, 0, 193), (80, 85, 1, 192), (3, 80, 0, 5), (10, 70, 150, 151), (4, 4, 0, 10)]
        n = ops.clear_rects(view, rects[:4]) + ops.clear_rects(view, rects[4:])
        assert n == 4  # (the empty rectangle is skipped)
        for (y0, y1, x0, x1) in rects:
            ref[y0:y1, x0:x1] = 0
        torch.cuda.synchronize()
        assert np.array_equal(view.cpu().numpy(), ref)
        assert np.array_equal(t.cpu().numpy()[:5], a[:5]) and np.array_equal(t.cpu().numpy()[:, :7], a[:, :7])

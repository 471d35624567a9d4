"""Deterministic scenarios shared by tests/test_gpu_strips.py (which checks them against the oracle pipeline) and
tests/strip_worker.py (which runs them on a strip-partitioned mosaic, two ranks on one GPU).  Not a test module."""
import numpy as np


def _shift(dx, dy):
    return [[1.0, 0.0, float(dx)], [0.0, 1.0, float(dy)]]


def _rot(deg, tx, ty):
    th = np.deg2rad(deg)
    c, s = np.cos(th), np.sin(th)
    return [[c, -s, tx], [s, c, ty]]


def uhd_scenarios():
    """BASELINE configs[4]'s frame size: three 3840x2160 frames marching down (1760 new rows each), so the mosaic ends 5680 rows
    tall and every frame's ROI crosses the boundary between the two strips.  Same tuple layout as scenarios()."""
    from image_stitcher_amd.synthetic import procedural_world
    world = procedural_world(seed=7, h=1320, w=2300).repeat(2, axis=0).repeat(2, axis=1)  # (2640, 4600): a quarter of the generation time
    ih, iw = 2160, 3840
    frames = [np.ascontiguousarray(world[y:y + ih, x:x + iw]) for y, x in ((0, 0), (230, 310), (480, 700))]
    Ms, rh = [], ih
    for k, (deg, tx) in enumerate(((0.3, 13.7), (-0.25, -21.3))):
        ty = (ih + rh - 400) - (rh + ih) // 2 + 0.4
        Ms.append(_rot(deg, tx, ty))
        rh += ih - 400
    return [("uhd_down_b1", frames, Ms, 0, 1, 0.4)]


def scenarios():
    """[(name, frames, Ms, type, blender, contrib)] -- every M is given in the reference's padded-canvas coordinates
    (frame at ((rw+iw)//2, (rh+ih)//2), mosaic at (iw, ih)); the placements only have to overlap roughly."""
    from image_stitcher_amd.synthetic import crop_path, procedural_world
    world = procedural_world(seed=42, h=800, w=2200)
    out = []
    frames = crop_path(world, 9, (200, 260), seed=4, max_rot_deg=0.0)[0]
    # (1) marching DOWN then up-left past the origin: strips gain rows, boundaries are crossed by every frame's ROI
    Ms, rh = [], 200
    for k in range(1, 9):
        if k <= 5:
            ty = (200 + rh - 120) - (rh + 200) // 2 + 0.4
            Ms.append(_rot(0.8 * (-1) ** k, 7 * (k % 3) - 5.3, ty))
            rh += 80
        else:
            Ms.append(_rot(-0.5, -150.25 - 20 * k, -(rh // 2) - 30.5))
    for blender in (1, 0):
        out.append((f"march_down_b{blender}", frames, Ms, 0, blender, 0.3))
    # (2) homographies
    Hs = []
    for k in range(1, 5):
        Hm = np.array(_rot(1.1 * (-1) ** k, 40.0 * k - 60, 25.5 * k - 10) + [[0, 0, 1]], np.float64)
        Hm[2, 0], Hm[2, 1] = 2e-5 * k, -1.5e-5
        Hs.append(Hm.tolist())
    out.append(("homography_b1", frames[:5], Hs, 1, 1, 0.5))
    # (3) the gradient blender's exact slow path (rows_ok == 0), on the strip boundary
    ref = np.full((220, 300, 3), 90, np.uint8)
    f = np.zeros((200, 300, 3), np.uint8)
    f[30:170, 20:140] = (120, 80, 200)
    f[90:110, 250:290] = (3, 0, 0)
    f[5:9, 40:60] = (4, 0, 1)
    out.append(("rows_slow", [ref, f], [_shift(-40, 6)], 0, 1, 0.5))
    # (4) alpha wrap-to-zero: the crop shrinks (rescan) and gray == 0 leftovers are cleared, then two more frames
    ref = np.zeros((120, 200, 3), np.uint8)
    ref[10:110, 20:200] = 70
    ref[60, 0] = (250, 250, 250)
    f = np.zeros((100, 160, 3), np.uint8)
    f[50, 30] = 6
    f[20:80, 60:120] = 6
    f[95:97, 60:80] = (3, 0, 0)
    f2 = np.zeros((100, 160, 3), np.uint8)
    f2[10:90, 10:150] = 40
    out.append(("alpha_wrap_b0", [ref, f, f2], [_shift(-50, 0), _shift(30, -5)], 0, 0, 0.5))
    out.append(("alpha_wrap_b1", [ref, f, f2], [_shift(-50, 0), _shift(30, -5)], 0, 1, 0.5))
    return out


def collection_images():
    """Two interleaved chains for ArbitraryStitcher.get_collections / stitch_collections."""
    from image_stitcher_amd.synthetic import crop_path, procedural_world
    world = procedural_world(seed=42, h=800, w=2200)
    a = crop_path(world, 3, (240, 320), seed=3, max_rot_deg=1.0)[0]
    b = crop_path(world, 2, (240, 320), seed=5, start=(1500.0, 400.0))[0]
    return [a[0], b[0], a[2], a[1], b[1]]

"""gloo tests (CPU, world 1 / 2 / 3) of the strip-partitioned mosaic's plumbing (image_stitcher_amd.strips):
row ownership, arena growth in every direction, the union of the strips' bounding boxes, the agreement on the slow paths
and the all_gather that assembles a detector region.  There is no GPU in this test: ops.frame_blend is replaced INSIDE THE
TEST by a stand-in that computes the same contract (warp, blend only the clip rows, result block) with the CPU oracle.
The strips must concatenate to the mosaic the oracle's single-canvas pipeline (tests/oracle_stitch.py) produces, bit for
bit; tests/test_gpu_strips.py proves the same with the HIP kernels on two ranks sharing one GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frames(n, h, w, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        f = rng.integers(40, 256, size=(h, w, 3), dtype=np.uint8)
        f[rng.integers(0, h, 12), rng.integers(0, w, 12)] = 0       # black pixels inside the frame
        out.append(f)
    return out


# world-coordinate placements (dx, dy, angle): the mosaic grows right/down, then up/left past the first image's origin
PLACEMENTS = [(30, 22, 0.05), (55, 47, -0.04), (-25, -30, 0.03), (10, 70, 0.0), (-40, 35, 0.08)]


def _canvas_M(place, ox, oy, cwx, cwy, iw, ih):
    """2x3 M: frame pasted at canvas (ox, oy) -> rotated about its centre, top-left at world (dx, dy); cw* = canvas->world."""
    dx, dy, ang = place
    c, s = np.cos(ang), np.sin(ang)
    cx, cy = ox + iw / 2.0, oy + ih / 2.0
    tx, ty = dx - cwx + iw / 2.0, dy - cwy + ih / 2.0
    return np.array([[c, -s, tx - c * cx + s * cy], [s, c, ty - s * cx - c * cy]], np.float64)


def _oracle_chain(frames, blender, orc, placements=None):
    """The reference's loop on whole canvases; returns [(mosaic, world origin)] after each frame."""
    ref, wx, wy = frames[0], 0, 0
    states = []
    for f, place in zip(frames[1:], PLACEMENTS if placements is None else placements):
        ih, iw = f.shape[:2]
        rh, rw = ref.shape[:2]
        cur_c, ref_c = orc.get_padded_images(f, ref)
        H, W = ref_c.shape[:2]
        ox, oy = (rw + iw) // 2, (rh + ih) // 2
        M = _canvas_M(place, ox, oy, wx - iw, wy - ih, iw, ih)
        warped = orc.warp(cur_c, M, (W, H), False)
        merged = orc.alpha_blend(warped, ref_c, 0.5) if blender == 0 else orc.gradient_blend(warped, ref_c)
        (x0, y0), (x1, y1) = orc.bbox_nonzero(merged)
        ref, wx, wy = merged[y0:y1, x0:x1], wx - iw + x0, wy - ih + y0
        states.append((ref.copy(), wx, wy))
    return states


def _install_oracle_frame_blend(ops, orc):
    """Stand-in for ops.frame_blend on CPU tensors -- same arguments, same result block, oracle arithmetic."""

    def frame_blend(ctx, src, M, perspective, src_xy, canvas_wh, arena_t, c2a, q_rect, blender, contrib=0.5, feather=10,
                    arith="f32", fetch=True, clip_rows=None):
        W, H = canvas_wh
        ox, oy = src_xy
        cax, cay = c2a
        s = src.numpy()
        ih, iw = s.shape[:2]
        cur = np.zeros((H, W, 3), np.uint8)
        cur[oy:oy + ih, ox:ox + iw] = s
        warped = orc.warp(cur, np.asarray(M), (W, H), bool(perspective))
        wb = orc.bbox_nonzero(warped)
        c0, c1 = (0, H) if clip_rows is None else clip_rows
        a = arena_t.numpy()
        ref = np.zeros((H, W, 3), np.uint8)                       # my rows of the mosaic on the virtual canvas
        y0, y1 = max(c0, -cay), min(c1, a.shape[0] - cay)
        x0, x1 = max(0, -cax), min(W, a.shape[1] - cax)
        if y1 > y0 and x1 > x0:
            ref[y0:y1, x0:x1] = a[y0 + cay:y1 + cay, x0 + cax:x1 + cax]
        qx0, qy0, qw, qh = q_rect
        res = dict(wbbox=False, roi=None, mbbox=None, mfound=False, zeroed=False, dirty=True, done=True)
        if wb is False:
            return res
        (bx0, by0), (bx1, by1) = wb
        res["wbbox"] = (bx0, by0, bx1, by1)
        assert qx0 <= bx0 and qy0 <= by0 and bx1 <= qx0 + qw and by1 <= qy0 + qh, "Q must contain the warped image"
        if blender == 0:
            merged = orc.alpha_blend(warped, ref, contrib)
            wx0, wy0, wx1, wy1 = qx0, qy0, qx0 + qw, qy0 + qh
        else:
            merged = orc.gradient_blend(warped, ref, feather)
            wx0, wy0, wx1, wy1 = max(bx0 - 50, 0), max(by0 - 50, 0), min(bx1 + 50, W), min(by1 + 50, H)
            res["roi"] = (wx0, wy0, wx1, wy1)
        wy0, wy1 = max(wy0, c0), min(wy1, c1)
        if wy1 <= wy0:
            return res
        assert y0 <= wy0 and wy1 <= y1 and x0 <= wx0 and wx1 <= x1, "rows inside the clip must be addressable"
        before = orc.bgr2gray(np.ascontiguousarray(ref[wy0:wy1, wx0:wx1])) != 0
        after = orc.bgr2gray(np.ascontiguousarray(merged[wy0:wy1, wx0:wx1])) != 0
        res["zeroed"] = bool((before & ~after).any())
        a[wy0 + cay:wy1 + cay, wx0 + cax:wx1 + cax] = merged[wy0:wy1, wx0:wx1]
        ys, xs = np.nonzero(after)
        if len(xs):
            res["mfound"] = True
            res["mbbox"] = (int(xs.min()) + wx0, int(ys.min()) + wy0, int(xs.max()) + 1 + wx0, int(ys.max()) + 1 + wy0)
        return res

    ops.frame_blend = frame_blend


def _worker(rank, world, port, blender, out_path, placements=None):
    placements = PLACEMENTS if placements is None else placements
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from image_stitcher_amd import ops
        from image_stitcher_amd.device_image import DeviceImage
        from image_stitcher_amd.stitcher import Stitcher
        from image_stitcher_amd.strips import StripMosaic
        from oracle import orc
        _install_oracle_frame_blend(ops, orc)
        frames = _frames(1 + len(placements), 48, 64, seed=11)
        st = Stitcher(algorithm=0, blender=blender, device=torch.device("cpu"))
        st._ctx = object()                                          # the stand-in ignores the workspace handle
        m = StripMosaic.from_image(frames[0], dist.group.WORLD, "cpu")
        assert m.shape == frames[0].shape and np.array_equal(np.asarray(m), frames[0])
        snaps = []
        for f, place in zip(frames[1:], placements):
            ih, iw = f.shape[:2]
            rh, rw = m.shape[:2]
            W, H = rw + 2 * iw, rh + 2 * ih
            ox, oy = (rw + iw) // 2, (rh + ih) // 2
            M = _canvas_M(place, ox, oy, m.wx0 - iw, m.wy0 - ih, iw, ih)
            got, wc = m.blend(st, DeviceImage.from_any(f, torch.device("cpu")), M, W, H, iw, ih, ox, oy)
            assert got is m and wc
            # the rows a rank holds are exactly its share of the mosaic's rectangle
            y0, y1 = m.local_rows()
            assert m.bounds[rank] <= y0 <= y1 <= m.bounds[rank + 1]
            full = np.asarray(m).copy()
            # a detector region straddling every strip boundary, partly outside the mosaic
            reg = m.region(-7, -5, m.w // 2 + 9, m.h + 6)
            sizes = [b - a for a, b in (m._rows_of(r) for r in range(world))]
            snaps.append((full, m.wx0, m.wy0, reg, sizes, dict(st.frame_stats)))
        if rank == 0:
            np.save(out_path, np.array(snaps, dtype=object), allow_pickle=True)
            assert st.frame_stats["frames"] == len(placements)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,blender", [(1, 1), (2, 1), (3, 1), (2, 0)])
def test_strips_concatenate_to_the_single_canvas_mosaic(tmp_path, orc, world, blender):
    out = str(tmp_path / "snaps.npy")
    mp.spawn(_worker, args=(world, _free_port(), blender, out), nprocs=world, join=True)
    snaps = np.load(out, allow_pickle=True)  # written by this test itself
    frames = _frames(1 + len(PLACEMENTS), 48, 64, seed=11)
    exp = _oracle_chain(frames, blender, orc)
    assert len(snaps) == len(exp)
    for k, ((full, wx, wy, reg, _sizes, _stats), (emos, ewx, ewy)) in enumerate(zip(snaps, exp)):
        assert (wx, wy) == (ewx, ewy), f"frame {k}: world origin {(wx, wy)} != {(ewx, ewy)}"
        assert full.shape == emos.shape and np.array_equal(full, emos), f"frame {k}: mosaic differs"
        h, w = emos.shape[:2]
        pad = np.zeros((h + 11, w // 2 + 16, 3), np.uint8)
        src = emos[:, :w // 2 + 9]
        pad[5:5 + h, 7:7 + src.shape[1]] = src
        assert np.array_equal(reg, pad), f"frame {k}: gathered region differs"


def test_strip_bounds_split_the_first_image_evenly():
    from image_stitcher_amd.strips import _INF
    for world, h in ((1, 10), (2, 1080), (8, 1081)):
        b = [-_INF] + [(k * h) // world for k in range(1, world)] + [_INF]
        assert len(b) == world + 1 and all(b[i] <= b[i + 1] for i in range(world))
        sizes = [min(h, b[r + 1]) - max(0, b[r]) for r in range(world)]
        assert sum(sizes) == h and max(sizes) - min(sizes) <= 1


# the mosaic grows to 4x the first image's height, downwards only: without re-splitting every new row lands on the last rank
DOWNWARD = [(6 * (k % 3) - 4, 20 * k, 0.03 * (-1) ** k) for k in range(1, 9)]


@pytest.mark.parametrize("world,blender", [(2, 1), (3, 0), (4, 1)])
def test_strips_rebalance_when_the_mosaic_grows_downward(tmp_path, orc, world, blender):
    """A mosaic that grows 4x downward: the bounds are re-split (rows move point to point) so that no strip holds more
    than twice the mean number of rows, frames that do not meet a rank's rows run no kernel there (routing), and after
    every frame the strips still concatenate to the oracle's single-canvas mosaic, bit for bit."""
    out = str(tmp_path / "snaps.npy")
    mp.spawn(_worker, args=(world, _free_port(), blender, out, DOWNWARD), nprocs=world, join=True)
    snaps = np.load(out, allow_pickle=True)  # written by this test itself
    frames = _frames(1 + len(DOWNWARD), 48, 64, seed=11)
    exp = _oracle_chain(frames, blender, orc, DOWNWARD)
    for k, ((full, wx, wy, reg, sizes, stats), (emos, ewx, ewy)) in enumerate(zip(snaps, exp)):
        assert (wx, wy) == (ewx, ewy) and full.shape == emos.shape and np.array_equal(full, emos), f"frame {k}"
        assert sum(sizes) == emos.shape[0]
        assert max(sizes) * world <= 2 * emos.shape[0] + world and max(sizes) <= 2 * min(sizes) + 2, (k, sizes)
    assert snaps[-1][0].shape[0] >= 4 * 48 - 8
    final_sizes, final_stats = snaps[-1][4], snaps[-1][5]
    assert final_stats.get("rebalances", 0) >= 1 and min(final_sizes) > 0 and max(final_sizes) <= 2 * min(final_sizes) + 2
    assert final_stats.get("routed_away", 0) >= 1, "rank 0 holds the top rows: late frames must not run kernels there"

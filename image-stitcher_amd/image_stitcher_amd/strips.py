"""Strip-partitioned mosaic (BASELINE configs[4]: "output canvas strip-partitioned across 8 GPUs").

The mosaic that ArbitraryStitcher.stitch_collections grows (reference stitcher.py:306-374) is split into contiguous
ROW STRIPS, one per rank of a torch.distributed group; every rank keeps only its strip resident (a MosaicArena).  All
ranks run the same greedy loop in lockstep (SPMD): the CPU detector sees the same pixels on every rank, the GPU
matcher and RANSAC are deterministic, so every rank derives the same transform M without exchanging it.

Per stitched frame
  * the region of the mosaic the detector needs (Stitcher.detectAndComputeFromCorners, stitcher.py:159-165) is
    assembled from the strips that hold its rows: ONE all_gather_into_tensor of fixed-shape uint8 row blocks (every
    rank derives every rank's rows from the bounds, so no header travels; RCCL on device tensors when the group is
    nccl); the "halo" of a strip boundary is whatever part of the region lies on the other side of it;
  * warp + blend is ROUTED: the warped frame, bbox(warped), the gradient blender's ROI (bbox +- 50 px,
    utils/proc.py:199-201) and its row / column feather extents depend on the frame and M only, NOT on the mosaic, so
    a rank whose rows meet the frame's rectangle Q (the warped quad +- 52 px) runs isx_frame_blend -- K1/K2 for the
    whole frame, K3 on its own rows (clip_rows) -- and a rank whose rows do not meet it runs NOTHING: it takes the
    warped image's bounding box from the owners in the reduction below.  No pixel crosses a strip boundary; a frame
    whose ROI straddles a boundary is blended by both owners, each on its side;
  * the merged canvas's bounding box (crop_image, stitcher.py:225), the owners' bbox(warped) and the slow-path flags
    travel in ONE all_reduce(MAX) of 14 integers (a second one only on the rare rescan path);
  * when the mosaic has grown so that one strip holds more than twice the mean number of rows, the bounds are re-split
    evenly over the mosaic's rows and the rows that change owner move point to point (batch_isend_irecv).

Coordinates: "world" = pixel coordinates fixed to the first reference image's top-left corner; rank r owns world rows
[bounds[r], bounds[r+1]) (the first and last strip are open ended, so the mosaic may grow in every direction).
"""
import numpy as np
import torch

from . import ops
from .device_image import MosaicArena

_INF = 1 << 30


def _dist():
    import torch.distributed as dist
    return dist


class StripMosaic:
    """Handle of a row-partitioned mosaic; stands in for the ndarray / DeviceImage `reference_image` of
    stitch_collections.  np.asarray() gathers the whole mosaic (a collective: every rank must call it)."""

    def __init__(self, group, device, bounds, rect, channels, ndim):
        dist = _dist()
        self.group, self.device = group, torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.bounds = list(bounds)          # world rows: rank r owns [bounds[r], bounds[r+1])
        self.wx0, self.wy0, self.w, self.h = rect   # the mosaic's rectangle in world coordinates (same on all ranks)
        self.channels, self.ndim = channels, ndim
        self.arena = None                   # MosaicArena of the local rows (None: this rank holds no row yet)
        self.ox = self.oy = 0               # world coordinates of arena.t[0, 0]
        self._host = None
        self.bbox = False                   # bbox of the gray != 0 pixels of the whole mosaic, world coordinates (or False)
        self.dirty = True                   # bytes of gray == 0 pixels may lie outside the crop
        self._flags, self._wbox = (0, 0), False   # results of the last _union_over_ranks

    # ---- construction ---------------------------------------------------------------------------------
    @classmethod
    def from_image(cls, img, group, device, pad_xy=(0, 0)):
        """Partition a host / device image: inner boundaries split ITS rows evenly."""
        dist = _dist()
        world = dist.get_world_size(group)
        t = img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(img).astype(np.uint8, copy=False)))
        h, w = t.shape[:2]
        bounds = [-_INF] + [(k * h) // world for k in range(1, world)] + [_INF]
        m = cls(group, device, bounds, (0, 0, w, h), 1 if t.ndim == 2 else t.shape[2], t.ndim)
        y0, y1 = m.local_rows()
        if y1 > y0:
            part = t[y0:y1].to(m.device)
            m.arena = MosaicArena.around(part, int(pad_xy[0]), int(pad_xy[1]))
            m.ox, m.oy = -m.arena.x, y0 - m.arena.y
        # bbox of the gray != 0 pixels of the whole image (a raw first frame may have black borders)
        lb = m._local_bbox_world(None)
        gb = m._union_over_ranks(lb)
        m.bbox = gb                          # world coordinates or False
        m.dirty = True
        return m

    # ---- geometry -------------------------------------------------------------------------------------
    def local_rows(self):
        """World rows [y0, y1) of the mosaic that this rank holds."""
        y0 = max(self.wy0, self.bounds[self.rank])
        y1 = min(self.wy0 + self.h, self.bounds[self.rank + 1])
        return (y0, max(y0, y1))

    @property
    def shape(self):
        return (self.h, self.w) if self.ndim == 2 else (self.h, self.w, self.channels)

    def __len__(self):
        return self.h

    def _sync_arena_rect(self):
        """arena.x/y/w/h <- (global rect ∩ this rank's rows) in arena coordinates."""
        if self.arena is None:
            return
        y0, y1 = self.local_rows()
        self.arena.x, self.arena.w = self.wx0 - self.ox, self.w
        self.arena.y, self.arena.h = y0 - self.oy, y1 - y0

    def _ensure_cover(self, wx_lo, wy_lo, wx_hi, wy_hi, iw, ih, st=None):
        """Make the world rectangle addressable in the local arena (created on first need, grown with slack)."""
        if wx_hi <= wx_lo or wy_hi <= wy_lo:
            return
        if self.arena is None:
            shape = (wy_hi - wy_lo + 2 * ih, wx_hi - wx_lo + 2 * iw) + ((self.channels,) if self.ndim == 3 else ())
            self.arena = MosaicArena(torch.zeros(shape, dtype=torch.uint8, device=self.device), iw, ih, 0, 0, bbox=False)
            self.ox, self.oy = wx_lo - iw, wy_lo - ih
            if st is not None:
                st.frame_stats["arena_allocs"] += 1
            return
        a = self.arena
        t_before = a.t
        if a.w == 0 or a.h == 0:  # nothing to carry over: anchor the empty rectangle inside the arena
            a.x, a.y, a.w, a.h = 0, 0, 0, 0
        dx, dy = a.ensure(wx_lo - self.ox, wy_lo - self.oy, wx_hi - self.ox, wy_hi - self.oy,
                          max(iw, a.t.shape[1] // 2), max(ih, a.t.shape[0] // 2))
        self.ox, self.oy = self.ox - dx, self.oy - dy
        if st is not None:
            st.frame_stats["arena_allocs"] += a.t is not t_before

    # ---- collectives ----------------------------------------------------------------------------------
    def _union_over_ranks(self, box, flags=(0, 0), wbox=False):
        """box: (x0,y0,x1,y1) world or False -> union over the ranks (False if none); flags are OR-ed; wbox (the warped
        image's bounding box, known to the ranks that ran the frame kernels: False elsewhere) comes back in
        self._wbox.  One all_reduce(MAX): minima travel negated."""
        dist = _dist()
        v = [-_INF] * 10
        if box:
            v[0:4] = [-box[0], -box[1], box[2], box[3]]
        v[4], v[5] = int(flags[0]), int(flags[1])
        if wbox:
            v[6:10] = [-wbox[0], -wbox[1], wbox[2], wbox[3]]
        t = torch.tensor(v, dtype=torch.int64)
        if self._comm_on_device():
            t = t.to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        v = t.cpu().tolist()
        self._flags = (max(0, v[4]), max(0, v[5]))
        self._wbox = (-v[6], -v[7], v[8], v[9]) if v[8] > -v[6] and v[9] > -v[7] else False
        return (-v[0], -v[1], v[2], v[3]) if v[2] > -v[0] and v[3] > -v[1] else False

    def _comm_on_device(self):
        dist = _dist()
        return dist.get_backend(self.group) == "nccl"

    def _local_bbox_world(self, rect_arena):
        """bbox of gray != 0 pixels of the local arena inside rect_arena = (x0,y0,x1,y1) (None: the local mosaic part),
        in world coordinates, or False."""
        if self.arena is None:
            return False
        a = self.arena
        x0, y0, x1, y1 = (a.x, a.y, a.x + a.w, a.y + a.h) if rect_arena is None else rect_arena
        if x1 <= x0 or y1 <= y0:
            return False
        view = a.t[y0:y1, x0:x1]
        if view.is_cuda:
            bb = ops.bbox_nonzero(view).cpu().numpy()
            if not bb[4]:
                return False
            bx0, by0, bx1, by1 = (int(v) for v in bb[:4])
        else:  # CPU tensors: only the plumbing tests (gloo, no GPU) come here
            v = view.numpy()
            g = v if v.ndim == 2 else ((v[..., 0].astype(np.int32) * 3735 + v[..., 1].astype(np.int32) * 19235
                                        + v[..., 2].astype(np.int32) * 9798 + 16384) >> 15)
            ys, xs = np.nonzero(g)
            if len(xs) == 0:
                return False
            bx0, by0, bx1, by1 = int(xs.min()), int(ys.min()), int(xs.max()) + 1, int(ys.max()) + 1
        return (bx0 + x0 + self.ox, by0 + y0 + self.oy, bx1 + x0 + self.ox, by1 + y0 + self.oy)

    def _rows_of(self, r, bounds=None):
        """World rows [y0, y1) of the mosaic held by rank r (every rank can tell: bounds and rectangle are common)."""
        b = self.bounds if bounds is None else bounds
        y0 = max(self.wy0, b[r])
        return y0, max(y0, min(self.wy0 + self.h, b[r + 1]))

    def _comm_device(self):
        return self.device if self._comm_on_device() else torch.device("cpu")

    def region(self, x0, y0, x1, y1):
        """Pixels [y0:y1, x0:x1] of the mosaic (mosaic coordinates, zeros outside it) as a numpy array on EVERY rank.
        Each rank contributes the rows it holds as one fixed-shape uint8 block (padded to the tallest contribution;
        which rows belong to whom follows from the bounds, so no header is exchanged): ONE all_gather_into_tensor --
        RCCL on device tensors when the group is nccl, host tensors over gloo -- and one download of the result."""
        dist = _dist()
        tail = (self.channels,) if self.ndim == 3 else ()
        shape = (max(0, y1 - y0), max(0, x1 - x0)) + tail
        rx0, rx1 = max(x0, 0) + self.wx0, min(x1, self.w) + self.wx0          # world columns of the request inside the mosaic
        spans = []
        for r in range(self.world):
            ly0, ly1 = self._rows_of(r)
            ry0, ry1 = max(y0 + self.wy0, ly0), min(y1 + self.wy0, ly1)
            spans.append((ry0, max(ry0, ry1)))
        rows = max(b - a for a, b in spans)
        if rows == 0 or rx1 <= rx0 or shape[0] == 0 or shape[1] == 0:
            return np.zeros(shape, np.uint8)
        cdev = self._comm_device()
        block = torch.zeros((rows, rx1 - rx0) + tail, dtype=torch.uint8, device=cdev)
        my0, my1 = spans[self.rank]
        if self.arena is not None and my1 > my0:
            block[:my1 - my0] = self.arena.t[my0 - self.oy:my1 - self.oy, rx0 - self.ox:rx1 - self.ox].to(cdev)
        allb = torch.empty((self.world * rows,) + tuple(block.shape[1:]), dtype=torch.uint8, device=cdev)
        dist.all_gather_into_tensor(allb, block, group=self.group)
        out = torch.zeros(shape, dtype=torch.uint8, device=cdev)
        cx = rx0 - self.wx0 - x0
        for r, (a, b) in enumerate(spans):
            if b > a:
                out[a - self.wy0 - y0:b - self.wy0 - y0, cx:cx + (rx1 - rx0)] = allb[r * rows:r * rows + (b - a)]
        return out.cpu().numpy()

    def numpy(self):
        if self._host is None:
            self._host = self.region(0, 0, self.w, self.h)
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    # ---- warp + blend (the strip form of Stitcher._blend_resident) ------------------------------------------
    def blend(self, st, img_d, M, W, H, rox, roy, ox, oy):
        """stitcher.py:216-225 on the partitioned mosaic, in place.  Returns (self | None, corners-of-warped | False)."""
        ih, iw = img_d.shape[:2]
        C = img_d.channels
        self._host = None
        qx0, qy0, qx1, qy1 = st._quad_region(M, ox, oy, iw, ih, W, H, 52)
        qw, qh = max(0, qx1 - qx0), max(0, qy1 - qy0)
        if qw == 0 or qh == 0:
            qx0 = qy0 = qw = qh = 0
        # canvas -> world: the mosaic's top-left (world wx0, wy0) is canvas (rox, roy)
        cwx, cwy = self.wx0 - rox, self.wy0 - roy
        clip0 = max(0, min(H, self.bounds[self.rank] - cwy))
        clip1 = max(clip0, min(H, self.bounds[self.rank + 1] - cwy))
        ry0, ry1 = max(qy0, clip0), min(qy0 + qh, clip1)              # canvas rows of Q that are mine
        mine = qw > 0 and ry1 > ry0
        if mine:
            # the arena must hold the mosaic's columns and Q's, over my rows of the mosaic and of Q
            ly0, ly1 = self.local_rows()
            self._ensure_cover(min(self.wx0, qx0 + cwx), min(ly0, ry0 + cwy) if ly1 > ly0 else ry0 + cwy,
                               max(self.wx0 + self.w, qx0 + qw + cwx), max(ly1, ry1 + cwy) if ly1 > ly0 else ry1 + cwy, iw, ih, st)
        arena_t = self.arena.t if self.arena is not None else torch.zeros((1, 1) + ((C,) if self.ndim == 3 else ()), dtype=torch.uint8, device=self.device)
        cax, cay = cwx - self.ox, cwy - self.oy                     # canvas -> arena
        ctx = st._frame_ctx()
        if mine:
            res = ops.frame_blend(ctx, img_d.t, M, st.type == 1, (ox, oy), (W, H), arena_t, (cax, cay), (qx0, qy0, qw, qh),
                                  st.blender, st.ref_image_contrib, 10, st.warp_arith, clip_rows=(clip0, clip1))
        else:
            # routing: none of this rank's rows meets the frame's rectangle -- no kernel runs here; bbox(warped) comes
            # from the owners with the reduction below
            res = dict(wbbox=False, roi=None, mbbox=None, mfound=False, zeroed=False, dirty=False, done=True)
            st.frame_stats["routed_away"] = st.frame_stats.get("routed_away", 0) + 1
        st.frame_stats["frames"] += 1
        wb = res["wbbox"]
        written = None
        rescan = False
        if st.blender == 0:
            written = (qx0, ry0, qx0 + qw, ry1) if mine else None
        elif wb:
            gx0, gy0, gx1, gy1 = res["roi"]
            wy0_, wy1_ = max(gy0, clip0), min(gy1, clip1)
            written = (gx0, wy0_, gx1, wy1_) if wy1_ > wy0_ else None
            if not res["done"]:
                # rows_ok == 0: extents must be taken inside the ROI (the four-kernel blender does that).  It blends a
                # whole ROI, so the rows of other strips are given as zeros and only this strip's rows are kept
                rescan = True
                st.frame_stats["rows_slow"] += 1
                if written is not None:
                    wv = ops.frame_warped_view(ctx, (qx0, qy0, qw, qh), C, img_d.ndim == 3)
                    tmp = torch.zeros((gy1 - gy0, gx1 - gx0) + ((C,) if self.ndim == 3 else ()), dtype=torch.uint8, device=self.device)
                    mine_t = arena_t[wy0_ + cay:wy1_ + cay, gx0 + cax:gx1 + cax]
                    tmp[wy0_ - gy0:wy1_ - gy0] = mine_t
                    ops.gradient_blend_roi(wv[gy0 - qy0:gy1 - qy0, gx0 - qx0:gx1 - qx0], tmp)
                    mine_t.copy_(tmp[wy0_ - gy0:wy1_ - gy0])
        # local union rectangle (arena coordinates) of what can hold non-zero bytes now
        box = False
        flags = (int(bool(res["zeroed"]) or rescan), int(bool(res["dirty"]) or rescan or self.dirty))
        # ONE reduction: the slow-path flags, bbox(warped) from the owners, and the union of the boxes of the pixels written
        # (valid unless some rank has to rescan)
        mb = res["mbbox"]
        box = (mb[0] + cwx, mb[1] + cwy, mb[2] + cwx, mb[3] + cwy) if res["mfound"] else False
        new = self._union_over_ranks(box, flags, wb)
        g_rescan, g_dirty = self._flags
        wcorners = ((self._wbox[0], self._wbox[1]), (self._wbox[2], self._wbox[3])) if self._wbox else False
        U = None
        if self.arena is not None:
            a = self.arena
            ux0, uy0, ux1, uy1 = a.x, a.y, a.x + a.w, a.y + a.h
            if a.w == 0 or a.h == 0:
                ux0, uy0, ux1, uy1 = _INF, _INF, -_INF, -_INF
            if written is not None:
                ux0, uy0 = min(ux0, written[0] + cax), min(uy0, written[1] + cay)
                ux1, uy1 = max(ux1, written[2] + cax), max(uy1, written[3] + cay)
            if ux1 > ux0 and uy1 > uy0:
                U = (ux0, uy0, ux1, uy1)
        if g_rescan:  # (rare: uint8 wrap-around of the alpha blender, the rows-slow path) every rank rescans its rectangle
            st.frame_stats["rescans"] += 1
            new = self._union_over_ranks(self._local_bbox_world(U) if U else False)
        if not g_rescan and self.bbox:
            ob = self.bbox
            new = (min(ob[0], new[0]), min(ob[1], new[1]), max(ob[2], new[2]), max(ob[3], new[3])) if new else ob
        if new is False:
            return None, wcorners
        self.bbox = new
        self.wx0, self.wy0, self.w, self.h = new[0], new[1], new[2] - new[0], new[3] - new[1]
        if self.arena is not None:
            # other strips may have widened the mosaic: keep its whole column range addressable over my rows
            ly0, ly1 = self.local_rows()
            if ly1 > ly0:
                if U:
                    # a re-allocation copies the arena's rectangle only: make it the union rectangle first, so that what this
                    # frame wrote outside the old one travels too (the rectangle is set to the new mosaic's right below)
                    self.arena.x, self.arena.y, self.arena.w, self.arena.h = U[0], U[1], U[2] - U[0], U[3] - U[1]
                    # arena coordinates shift when the arena is re-allocated
                    U = (U[0] + self.ox, U[1] + self.oy, U[2] + self.ox, U[3] + self.oy)
                self._ensure_cover(self.wx0, ly0, self.wx0 + self.w, ly1, iw, ih, st)
                if U:
                    U = (U[0] - self.ox, U[1] - self.oy, U[2] - self.ox, U[3] - self.oy)
            self._sync_arena_rect()
            if g_dirty and U:
                a = self.arena
                n0, n1, n2, n3 = a.x, a.y, a.x + a.w, a.y + a.h
                if a.w == 0 or a.h == 0:
                    a.t[U[1]:U[3], U[0]:U[2]] = 0
                else:
                    for (y0, y1, x0, x1) in ((U[1], n1, U[0], U[2]), (n3, U[3], U[0], U[2]), (n1, n3, U[0], n0), (n1, n3, n2, U[2])):
                        if y1 > y0 and x1 > x0:
                            a.t[y0:y1, x0:x1] = 0
                            st.frame_stats["clears"] += 1
        self.dirty = False
        self._maybe_rebalance(iw, ih, st)
        return self, wcorners

    # ---- re-splitting the bounds --------------------------------------------------------------------------
    def _maybe_rebalance(self, iw, ih, st=None):
        """The bounds were set by the first image's rows; a mosaic that grows downwards (or upwards) piles every new
        row onto the last (first) strip.  When one strip holds more than twice the mean number of rows, or twice another strip's, the bounds are
        re-split evenly over the mosaic's rows and the rows that change owner are sent point to point.  Every rank
        takes the same decision from the same numbers (bounds and rectangle are common): nothing is exchanged to decide."""
        if self.world == 1 or self.h < 4 * self.world:
            return False
        sizes = [b - a for a, b in (self._rows_of(r) for r in range(self.world))]
        if max(sizes) * self.world <= 2 * self.h and max(sizes) <= 2 * min(sizes):
            return False   # no strip above twice the mean, none above twice another
        new_bounds = [-_INF] + [self.wy0 + (k * self.h) // self.world for k in range(1, self.world)] + [_INF]
        self._move_rows(new_bounds, iw, ih)
        if st is not None:
            st.frame_stats["rebalances"] = st.frame_stats.get("rebalances", 0) + 1
        return True

    def _move_rows(self, new_bounds, pad_x, pad_y):
        dist = _dist()
        cdev = self._comm_device()
        tail = (self.channels,) if self.ndim == 3 else ()
        old = [self._rows_of(r) for r in range(self.world)]
        new = [self._rows_of(r, new_bounds) for r in range(self.world)]
        me = self.rank
        peer = (lambda r: dist.get_global_rank(self.group, r)) if self.group is not None else (lambda r: r)

        def cut(a, b):
            y0, y1 = max(a[0], b[0]), min(a[1], b[1])
            return (y0, y1) if y1 > y0 else None

        def my_rows(y0, y1):
            return self.arena.t[y0 - self.oy:y1 - self.oy, self.wx0 - self.ox:self.wx0 + self.w - self.ox]

        p2p, incoming = [], []
        for r in range(self.world):
            if r == me:
                continue
            out_rows = cut(old[me], new[r])
            if out_rows is not None:
                p2p.append(dist.P2POp(dist.isend, my_rows(*out_rows).to(cdev).contiguous(), peer(r), self.group))
            in_rows = cut(old[r], new[me])
            if in_rows is not None:
                buf = torch.empty((in_rows[1] - in_rows[0], self.w) + tail, dtype=torch.uint8, device=cdev)
                p2p.append(dist.P2POp(dist.irecv, buf, peer(r), self.group))
                incoming.append((in_rows, buf))
        if p2p:
            for req in dist.batch_isend_irecv(p2p):
                req.wait()
        ny0, ny1 = new[me]
        if ny1 > ny0:
            part = torch.zeros((ny1 - ny0, self.w) + tail, dtype=torch.uint8, device=self.device)
            kept = cut(old[me], new[me])
            if kept is not None:
                part[kept[0] - ny0:kept[1] - ny0] = my_rows(*kept)
            for (y0, y1), buf in incoming:
                part[y0 - ny0:y1 - ny0] = buf.to(self.device)
            self.arena = MosaicArena.around(part, int(pad_x), int(pad_y))
            self.ox, self.oy = self.wx0 - self.arena.x, ny0 - self.arena.y
        else:
            self.arena = None
        self.bounds = list(new_bounds)
        self._sync_arena_rect()

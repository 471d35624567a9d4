"""Tensor-level wrappers over the C ABI (include/isx.h).  PyTorch is used only to own device memory
and streams; every kernel is in libisx_hip.so.  There is no CPU fallback."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import IsxError, NormLimitError, NotIntegerValuedError, check


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Raw handle of the current stream of the current device.  (torch.cuda.current_stream() builds a Stream object and
    resolves the device index in Python -- 8-10 us per call, five calls per stitched frame.)"""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _require_cuda(t, name):
    if not t.is_cuda:
        raise IsxError(f"{name} must live on a HIP device (got {t.device}); the stitching core has no CPU path")


class PreparedDescriptors:
    """Descriptors of one or more images, resident in HBM in the layout the matcher consumes
    (int8 value-128 rows padded to 32/64/128 bytes, per-row key bases and norms)."""

    def __init__(self, desc, cbase, nrm, cin, starts, counts, d, d_pad, max_norm2, wide=False):
        self.desc, self.cbase, self.nrm, self.cin = desc, cbase, nrm, cin
        self.starts = np.asarray(starts, np.int64)  # first row of every image
        self.counts = np.asarray(counts, np.int64)  # rows of every image
        self.d, self.d_pad, self.max_norm2 = d, d_pad, max_norm2
        # True: norms beyond ISX_L2_EXACT_LIMIT -- squared distances may collide under sqrtf, so the set is matched by
        # the kernel that ranks on the float distances themselves (isx_knn2_l2_wide_items), still exactly as cv2
        self.wide = wide

    @property
    def n_images(self):
        return len(self.counts)

    @property
    def device(self):
        return self.desc.device


def prepare_descriptors(des_list, device=None, check_range=True, wide_ok=True):
    """des_list: list of (n_i, d) uint8 / float32 arrays or tensors (one per image; None or empty
    allowed).  Float descriptors must be integer valued in [0,255] (cv2 SIFT's are).  Returns
    PreparedDescriptors on `device`."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    tens, counts = [], []
    d = None
    dtype = None
    for a in des_list:
        if a is None:
            counts.append(0)
            continue
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        if t.ndim != 2:
            raise IsxError("descriptors must be 2-D (n, d)")
        if t.shape[0] == 0:
            counts.append(0)
            continue
        if d is None:
            d, dtype = t.shape[1], t.dtype
        elif t.shape[1] != d or t.dtype != dtype:
            raise IsxError("all descriptor arrays must share dtype and width")
        tens.append(t)
        counts.append(t.shape[0])
    if d is None:
        raise IsxError("no descriptors given")
    if dtype not in (torch.uint8, torch.float32):
        raise IsxError(f"descriptor dtype must be uint8 or float32 (got {dtype})")
    src = torch.cat([t.to(device, non_blocking=True) for t in tens], 0).contiguous()
    starts = np.concatenate([[0], np.cumsum(counts)])[:-1]
    return prepare_descriptor_tensor(src, starts, counts, check_range, wide_ok)


def prepare_descriptor_tensor(src, starts, counts, check_range=True, wide_ok=True):
    """src: (R, d) uint8/float32 device tensor holding all images back to back.  Descriptor sets whose norms exceed
    the exact range of the MFMA kernels are marked `wide` (matched by isx_knn2_l2_wide_items, equally exact) unless
    wide_ok=False, which raises NormLimitError instead."""
    _require_cuda(src, "src")
    L = _lib.lib()
    rows, d = src.shape
    d_pad = check(L.isx_desc_pad(d))
    dev = src.device
    with torch.cuda.device(dev):
        desc = torch.empty((rows, d_pad), dtype=torch.int8, device=dev)
        cbase = torch.empty((rows,), dtype=torch.int32, device=dev)
        nrm = torch.empty((rows,), dtype=torch.int32, device=dev)
        cin = torch.empty((rows,), dtype=torch.int32, device=dev)
        stats = torch.zeros((2,), dtype=torch.int32, device=dev)
        check(L.isx_desc_prep(_ptr(src), _lib.U8 if src.dtype == torch.uint8 else _lib.F32, rows, d,
                              _ptr(desc), _ptr(cbase), _ptr(nrm), _ptr(cin), _ptr(stats), _stream()))
        max_norm2 = None
        wide = False
        if check_range:
            s = stats.cpu().numpy()
            if s[0] & 1:
                raise NotIntegerValuedError("float descriptors are not integer valued in [0,255]: "
                               "use the float32 matcher (knn2_f32) instead of the exact integer path")
            max_norm2 = int(s[1])
            if 2 * max_norm2 >= _lib.L2_EXACT_LIMIT:
                if not wide_ok:
                    raise NormLimitError(f"descriptor norms too large for the MFMA integer path "
                                         f"(2*{max_norm2} >= {_lib.L2_EXACT_LIMIT})")
                wide = True
    return PreparedDescriptors(desc, cbase, nrm, cin, starts, counts, d, d_pad, max_norm2, wide)


def sort_rows_by_norm(P):
    """The same prepared set with the rows of every image sorted by norm (ascending, rotated by 64).  For the count kernels only: the
    number of good matches of an image pair does not depend on the row order, and isx_knn2_l2_count_pairs settles more
    queries without an exact recount when the 128 queries of a class have similar norms.  Needs images stored back to
    back (starts = cumsum(counts)), which is what prepare_descriptors produces; anything else is returned unchanged."""
    counts = np.asarray(P.counts, np.int64)
    starts = np.asarray(P.starts, np.int64)
    rows = P.desc.shape[0]
    if len(counts) == 0 or not np.array_equal(starts, np.concatenate([[0], np.cumsum(counts)[:-1]])) or int(counts.sum()) != rows:
        return P
    dev = P.device
    max_count = int(counts.max())
    if max_count <= SORT_MAX_ROWS:
        # one workgroup per image: keys in LDS, bitonic sort, the rows moved by the same workgroup (csrc/isx_rowsort.hip)
        uniform = int(counts[0]) if len(set(counts.tolist())) == 1 else 0
        with torch.cuda.device(dev):
            st_dev = None if uniform else torch.from_numpy(np.concatenate([starts, [rows]]).astype(np.int32)).to(dev)
            src = [t.contiguous() for t in (P.desc, P.cbase, P.nrm, P.cin)]
            out = [torch.empty_like(t) for t in src]
            check(_lib.lib().isx_sort_rows_by_norm(_ptr(src[0]), _ptr(src[1]), _ptr(src[2]), _ptr(src[3]), P.d_pad,
                                                   _ptr(st_dev), len(counts), uniform, max_count, *[_ptr(t) for t in out], _stream()))
        return PreparedDescriptors(out[0], out[1], out[2], out[3], P.starts, P.counts, P.d, P.d_pad, P.max_norm2, P.wide)
    with torch.cuda.device(dev):  # an image beyond ISX_SORT_MAX_ROWS rows: the same order (up to ties) with torch's sort
        if len(set(counts.tolist())) == 1:
            img = torch.arange(rows, device=dev, dtype=torch.int64) // int(counts[0])
        else:
            img = torch.repeat_interleave(torch.arange(len(counts), device=dev, dtype=torch.int64),
                                          torch.from_numpy(counts).to(dev))
        perm = torch.argsort((img << 32) | P.nrm.to(torch.int64))
        # The two tails of the sorted order are where a class of 128 rows spreads widely in norm (it then takes the kernel's
        # slower per-value correction).  Rotating every image's order by 64 puts its 64 smallest norms behind its 64 largest:
        # ONE wide class per image (the last) instead of two.
        if len(set(counts.tolist())) == 1:
            if int(counts[0]) > 256:
                perm = perm.view(len(counts), int(counts[0])).roll(-64, dims=1).reshape(-1)
        else:
            cnt = torch.from_numpy(counts).to(dev)[img]
            st = torch.from_numpy(starts).to(dev)[img]
            pos = torch.arange(rows, device=dev, dtype=torch.int64) - st
            src = st + torch.where(cnt > 256, (pos + 64) % cnt.clamp(min=1), pos)
            perm = perm[src]
        return PreparedDescriptors(P.desc[perm], P.cbase[perm], P.nrm[perm], P.cin[perm], P.starts, P.counts, P.d, P.d_pad,
                                   P.max_norm2, P.wide)


FRAME_MAX_Q = 32768    # include/isx.h ISX_FRAME_MAX_Q: largest side of the rectangle isx_frame_blend warps into
PAIR_MAX_ROWS = 4096   # isx_knn2_l2_count_pairs: rows per image for its four-group table (the per-row table lives in LDS)
PAIR_BIG_TRAIN, PAIR_BIG_QUERY = 8192, 16384  # beyond: two groups per row -- rows of the smaller / the larger image of a pair
SORT_MAX_ROWS = 16384  # include/isx.h ISX_SORT_MAX_ROWS: rows per image isx_sort_rows_by_norm sorts in LDS
PAIR_MIN_ITEMS = 512   # below this many pairs the one-workgroup-per-pair kernel does not beat the directed kernel (tools/pairs_min_items.py)


def pairs_kernel_applies(P, pair_items):
    """True when the both-directions kernel (isx_knn2_l2_count_pairs) can and should run this pair list."""
    it = np.asarray(pair_items).reshape(-1, 4)
    if P.wide or P.d_pad not in (64, 128) or len(it) < PAIR_MIN_ITEMS:
        return False
    lo, hi = np.minimum(it[:, 1], it[:, 3]), np.maximum(it[:, 1], it[:, 3])
    if int(hi.max()) <= PAIR_MAX_ROWS:
        return True
    # larger images: the kernel hands a pair it cannot hold back as (-1, -1) and knn2_l2_pairs re-runs it on the directed kernel;
    # worth a launch when most pairs fit
    return float(np.mean((lo <= PAIR_BIG_TRAIN) & (hi <= PAIR_BIG_QUERY))) >= 0.5


def knn2_l2_pairs(P, pair_items, ratio, plan=None, events=None, flags=0, stats=None, while_running=None):
    """Good-match counts of both directions of unordered image pairs (the loop body of get_collections, reference
    stitcher.py:285-296) from ONE distance matrix per pair: isx_knn2_l2_count_pairs, then isx_knn2_l2_count_items for
    the directions it reports as overflowing (overlapping images, more than 128 candidate queries).
    pair_items: (n,4) int32 rows (q_start, nq, t_start, nt).  Returns (n,2) int32 on the device:
    [:,0] = len(good) of (query = the q image, train = the t image), [:,1] = the other direction.  Exact counts.
    stats: optional dict, receives 'overflow_directions'.  while_running: optional callable, called once after the first launch
    is queued and before this function waits for it (host work that should overlap the kernel)."""
    L = _lib.lib()
    dev = P.device
    with torch.cuda.device(dev):
        if plan is None:
            plan = plan_items(pair_items, dev)
        run_items, perm, items_dev, inv_dev, inv_np = plan
        m = len(run_items)
        if m == 0:
            return torch.zeros((0, 2), dtype=torch.int32, device=dev)
        max_rows = int(max(run_items[:, 1].max(), run_items[:, 3].max()))
        counts = torch.empty((m, 2), dtype=torch.int32, device=dev)
        if events is not None:
            events[0].record()
        check(L.isx_knn2_l2_count_pairs(_ptr(P.desc), _ptr(P.nrm), _ptr(P.cin), P.d_pad, _ptr(items_dev), m, max_rows,
                                        float(ratio), _ptr(counts), int(flags), _stream()))
        flat = counts.view(-1)
        if while_running is not None:
            while_running()
        over = torch.nonzero(flat < 0).view(-1)          # (host sync: the length of the second stage's item list)
        n_over = int(over.shape[0])
        if stats is not None:
            stats["overflow_directions"] = n_over
        if n_over:
            e, d = over >> 1, (over & 1).bool()
            src = items_dev[e]
            di = torch.where(d[:, None], src[:, [2, 3, 0, 1]], src).contiguous()
            c2 = torch.zeros((n_over,), dtype=torch.int32, device=dev)
            check(L.isx_knn2_l2_count_items(_ptr(P.desc), _ptr(P.nrm), _ptr(P.cin), P.d_pad, _ptr(di), n_over, max_rows,
                                            float(ratio), _ptr(c2), 0, _stream()))
            flat[over] = c2
        if events is not None:
            events[1].record()
        return counts[inv_dev] if inv_np is not None else counts


TRAIN_GROUP = 4  # train images an XCD keeps resident while it streams query images past them (4 x 512 KiB of its 4 MiB L2)


def order_items_for_xcd(items, train_group=TRAIN_GROUP):
    """Reorder directed work items (n,4 int32: q_start,nq,t_start,nt) for the 8 XCD-private L2s: the kernel serves
    run position p on XCD p % 8, and each XCD gets one contiguous range of the list sorted by
    (group of `train_group` train images, query image, train image).  An XCD therefore keeps a group of train
    images resident and every query image it fetches serves up to `train_group` items in a row, instead of one
    new 512 KiB query image per item (train_group=1: the round-1 order, 142 GB of fabric traffic per launch of the
    512-image set).  Returns (items_interleaved (m,4), perm (m,) original index or -1 for padding)."""
    items = np.asarray(items, np.int32).reshape(-1, 4)
    n = len(items)
    if n == 0:
        return items, np.zeros((0,), np.int64)
    trank = np.unique(items[:, 2], return_inverse=True)[1].reshape(-1)  # dense rank of the train image
    order = np.lexsort((items[:, 2], items[:, 0], trank // max(1, int(train_group))))
    per = (n + 7) // 8
    perm = np.full((per, 8), -1, np.int64)
    for x in range(8):
        chunk = order[x * per:(x + 1) * per]
        perm[:len(chunk), x] = chunk
    perm = perm.reshape(-1)
    out = np.zeros((len(perm), 4), np.int32)
    valid = perm >= 0
    out[valid] = items[perm[valid]]
    return out, perm


def plan_items(items, device, xcd_order=True, train_group=TRAIN_GROUP):
    """Host-side launch plan for an item list: XCD-aware order, device copy, inverse permutation."""
    items = np.ascontiguousarray(np.asarray(items, np.int32).reshape(-1, 4))
    n = len(items)
    if xcd_order and n > 8:
        run_items, perm = order_items_for_xcd(items, train_group)
        valid = perm >= 0
        inv_np = np.empty((n,), np.int64)
        inv_np[perm[valid]] = np.nonzero(valid)[0]
        inv_dev = torch.from_numpy(inv_np).to(device)
    else:
        run_items, perm, inv_np, inv_dev = items, np.arange(n), None, None
    items_dev = torch.from_numpy(run_items).to(device)
    return run_items, perm, items_dev, inv_dev, inv_np


def knn2_l2_items(P, items, ratio, want_matches=False, xcd_order=True, plan=None, events=None, verify_all=False):
    """Run the fused kNN-2 + ratio kernel over directed work items.

    items: (n,4) int32 rows (q_start, nq, t_start, nt) into P's row space.
    plan: optional result of plan_items(items) (cached reordering + device copy of the item list).
    events: optional (start, end) torch.cuda.Event pair recorded immediately around the kernel launch
    on the launch stream (what bench.py uses for the per-launch kernel duration).
    want_matches=False runs the count-only kernel (isx_knn2_l2_count_items).
    Returns counts (n,) int32 device tensor [len(good_matches) per item] and, if want_matches,
    (idx, dist, out_starts): idx/dist are (sum nq, 2) device tensors, out_starts (n,) numpy int64.
    """
    L = _lib.lib()
    if plan is not None:
        n = len(plan[4]) if plan[4] is not None else len(plan[0])
    else:
        items = np.ascontiguousarray(np.asarray(items, np.int32).reshape(-1, 4))
        n = len(items)
    dev = P.device
    with torch.cuda.device(dev):
        if n == 0:
            z = torch.zeros((0,), dtype=torch.int32, device=dev)
            return (z, None) if want_matches else z
        if plan is None:
            plan = plan_items(items, dev, xcd_order)
        run_items, perm, items_dev, inv_dev, inv_np = plan
        reordered = inv_np is not None
        m = len(run_items)
        max_nq = int(run_items[:, 1].max())
        counts = torch.zeros((m,), dtype=torch.int32, device=dev)
        idx = dist = out_dev = None
        out_starts = None
        if want_matches:
            out_run = np.concatenate([[0], np.cumsum(run_items[:, 1].astype(np.int64))])
            total = int(out_run[-1])
            idx = torch.empty((total, 2), dtype=torch.int32, device=dev)
            dist = torch.empty((total, 2), dtype=torch.float32, device=dev)
            out_dev = torch.from_numpy(out_run[:-1].copy()).to(dev, non_blocking=True)
        if events is not None:
            events[0].record()
        if P.wide:  # norms beyond the MFMA kernels' exact range: the float-distance-ranking kernel, same outputs
            max_nt = int(run_items[:, 3].max())
            check(L.isx_knn2_l2_wide_items(_ptr(P.desc), _ptr(P.nrm), P.d_pad, _ptr(items_dev), m, max_nq, max_nt, float(ratio),
                                           _ptr(idx), _ptr(dist), _ptr(out_dev), _ptr(counts), _stream()))
        elif want_matches:
            check(L.isx_knn2_l2_items(_ptr(P.desc), _ptr(P.cbase), _ptr(P.nrm), P.d_pad, _ptr(items_dev), m,
                                      max_nq, float(ratio), _ptr(idx), _ptr(dist), _ptr(out_dev),
                                      _ptr(counts), _stream()))
        else:  # count-only kernel (verify_all: the test-hook `flags` of isx_knn2_l2_count_items, bits 0 to 2)
            check(L.isx_knn2_l2_count_items(_ptr(P.desc), _ptr(P.nrm), _ptr(P.cin), P.d_pad, _ptr(items_dev), m, max_nq,
                                            float(ratio), _ptr(counts), int(verify_all), _stream()))
        if events is not None:
            events[1].record()
        if reordered:
            counts_o = counts[inv_dev]
            if want_matches:
                out_starts = out_run[:-1][inv_np]
        else:
            counts_o = counts
            if want_matches:
                out_starts = out_run[:-1]
        if want_matches:
            return counts_o, (idx, dist, out_starts)
        return counts_o


def knn2_l2(q, t, ratio=0.65):
    """Single pair: cv2.BFMatcher().knnMatch(q, t, k=2) + Lowe ratio.
    q, t: (n,d) uint8 or integer-valued float32 (numpy or tensor).  Returns idx (nq,2) int32,
    dist (nq,2) float32, good_count (int32 0-d tensor), all on the device."""
    nq = 0 if q is None else q.shape[0]
    nt = 0 if t is None else t.shape[0]
    if nq == 0 or nt == 0:
        raise IsxError("knn2_l2 needs non-empty query and train sets (the reference returns [] before matching)")
    P = prepare_descriptors([q, t])
    items = np.array([[P.starts[0], nq, P.starts[1], nt]], np.int32)
    counts, (idx, dist, _) = knn2_l2_items(P, items, ratio, want_matches=True, xcd_order=False)
    return idx, dist, counts[0]


def _pair_outputs(nq, dev, want_matches=True):
    idx = torch.empty((nq, 2), dtype=torch.int32, device=dev)
    dist = torch.empty((nq, 2), dtype=torch.float32, device=dev)
    start = torch.zeros((1,), dtype=torch.int64, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    return idx, dist, start, cnt


def _as_dev(a, dtype, device):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    if t.dtype != dtype:
        raise IsxError(f"expected {dtype} descriptors, got {t.dtype}")
    return t.to(device).contiguous()


def knn2_hamming(q, t, ratio=0.65, device=None):
    """Opt-in: cv2.BFMatcher(cv2.NORM_HAMMING).knnMatch(q, t, k=2) + Lowe ratio for uint8 (binary)
    descriptors of up to 64 bytes.  Returns idx (nq,2) int32, dist (nq,2) float32, good_count tensor."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    nq, nt = q.shape[0], t.shape[0]
    if nq == 0 or nt == 0:
        raise IsxError("knn2_hamming needs non-empty query and train sets")
    d = q.shape[1]
    if t.shape[1] != d or d > 64:
        raise IsxError("descriptor widths differ or exceed 64 bytes")
    d_pad = 32 if d <= 32 else 64
    with torch.cuda.device(device):
        both = torch.cat([_as_dev(q, torch.uint8, device), _as_dev(t, torch.uint8, device)], 0)
        if d_pad != d:
            both = torch.nn.functional.pad(both, (0, d_pad - d))
        both = both.contiguous()
        items = torch.tensor([[0, nq, nq, nt]], dtype=torch.int32, device=device)
        idx, dist, start, cnt = _pair_outputs(nq, device)
        check(_lib.lib().isx_knn2_hamming_items(_ptr(both), d_pad, _ptr(items), 1, nq, nt, float(ratio), _ptr(idx),
                                                _ptr(dist), _ptr(start), _ptr(cnt), _stream()))
    return idx, dist, cnt[0]


def knn2_f32(q, t, ratio=0.65, device=None):
    """cv2.BFMatcher().knnMatch(q, t, k=2) + Lowe ratio for float32 descriptors that are not integer
    valued (fp32 MFMA path, 1e-4 relative distance gate).  Widths up to 128 (zero padded to 32/64/128)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    nq, nt = q.shape[0], t.shape[0]
    if nq == 0 or nt == 0:
        raise IsxError("knn2_f32 needs non-empty query and train sets")
    d = q.shape[1]
    if t.shape[1] != d or d > 128:
        raise IsxError("descriptor widths differ or exceed 128")
    d_pad = 32 if d <= 32 else (64 if d <= 64 else 128)
    with torch.cuda.device(device):
        both = torch.cat([_as_dev(q, torch.float32, device), _as_dev(t, torch.float32, device)], 0)
        if d_pad != d:
            both = torch.nn.functional.pad(both, (0, d_pad - d))
        both = both.contiguous()
        nrm = torch.empty((nq + nt,), dtype=torch.float32, device=device)
        L = _lib.lib()
        check(L.isx_f32_norms(_ptr(both), nq + nt, d_pad, _ptr(nrm), _stream()))
        items = torch.tensor([[0, nq, nq, nt]], dtype=torch.int32, device=device)
        idx, dist, start, cnt = _pair_outputs(nq, device)
        check(L.isx_knn2_f32_items(_ptr(both), _ptr(nrm), d_pad, _ptr(items), 1, nq, float(ratio), _ptr(idx),
                                   _ptr(dist), _ptr(start), _ptr(cnt), _stream()))
    return idx, dist, cnt[0]


def alt_item_counts(des_list, items_of, kind, ratio, device=None, hamming_impl="auto"):
    """Batched good-match counts with the Hamming ("hamming", uint8) or float32 ("f32") matcher.
    des_list: per-image (n_i, d) arrays (None/empty allowed); items_of(starts, counts) -> (m,4) int32 work
    items in the concatenated row space.  Returns an int32 device tensor of m counts.
    hamming_impl: "fp4" = bit vectors through the FP4 matrix pipe (descriptors of up to 32 bytes), "valu" =
    XOR/popcount kernel, "auto" = fp4 when the width allows it.  Both give the same counts."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    dtype = torch.uint8 if kind == "hamming" else torch.float32
    tens, counts, d = [], [], None
    for a in des_list:
        if a is None or len(a) == 0:
            counts.append(0)
            continue
        t = _as_dev(a, dtype, device)
        d = t.shape[1] if d is None else d
        if t.shape[1] != d:
            raise IsxError("all descriptor arrays must share their width")
        tens.append(t)
        counts.append(t.shape[0])
    if d is None:
        raise IsxError("no descriptors given")
    limit = 64 if kind == "hamming" else 128
    if d > limit:
        raise IsxError(f"descriptor width {d} exceeds {limit}")
    d_pad = 32 if d <= 32 else (64 if d <= 64 else 128)
    starts = np.concatenate([[0], np.cumsum(counts)])[:-1]
    items = np.ascontiguousarray(items_of(starts, np.asarray(counts)), np.int32).reshape(-1, 4)
    m = len(items)
    with torch.cuda.device(device):
        out = torch.zeros((m,), dtype=torch.int32, device=device)
        if m == 0:
            return out
        allrows = torch.cat(tens, 0)
        if d_pad != d:
            allrows = torch.nn.functional.pad(allrows, (0, d_pad - d))
        allrows = allrows.contiguous()
        items_dev = torch.from_numpy(items).to(device)
        L = _lib.lib()
        max_nq, max_nt = int(items[:, 1].max()), int(items[:, 3].max())
        if kind == "hamming" and hamming_impl not in ("auto", "fp4", "valu"):
            raise IsxError("hamming_impl must be 'auto', 'fp4' or 'valu'")
        if kind == "hamming" and (hamming_impl == "fp4" or (hamming_impl == "auto" and d <= 32)):
            if d > 32:
                raise IsxError("the FP4 Hamming path takes descriptors of up to 32 bytes")
            rows = allrows.shape[0]
            fp4 = torch.empty((rows, 128), dtype=torch.uint8, device=device)
            nrm = torch.empty((rows,), dtype=torch.int32, device=device)
            cin = torch.empty((rows,), dtype=torch.int32, device=device)
            check(L.isx_hamming_fp4_prep(_ptr(allrows), rows, d_pad, _ptr(fp4), _ptr(nrm), _ptr(cin), _stream()))
            run_items, perm, run_dev, inv_dev, _ = plan_items(items, device)  # XCD-aware order, as the L2 path
            cnt = torch.zeros((len(run_items),), dtype=torch.int32, device=device)
            check(L.isx_knn2_hamming_count_fp4(_ptr(fp4), _ptr(nrm), _ptr(cin), _ptr(run_dev), len(run_items), max_nq,
                                               float(ratio), _ptr(cnt), _stream()))
            out = cnt[inv_dev] if inv_dev is not None else cnt
        elif kind == "hamming":
            check(L.isx_knn2_hamming_items(_ptr(allrows), d_pad, _ptr(items_dev), m, max_nq, max_nt, float(ratio),
                                           None, None, None, _ptr(out), _stream()))
        else:
            nrm = torch.empty((allrows.shape[0],), dtype=torch.float32, device=device)
            check(L.isx_f32_norms(_ptr(allrows), allrows.shape[0], d_pad, _ptr(nrm), _stream()))
            check(L.isx_knn2_f32_items(_ptr(allrows), _ptr(nrm), d_pad, _ptr(items_dev), m, max_nq, float(ratio),
                                       None, None, None, _ptr(out), _stream()))
    return out


def ratio_compact(idx, dist, ratio):
    """filter_matches as device arrays: (good_q, good_t) int32 tensors in query order (one sync to
    learn the count)."""
    nq = idx.shape[0]
    dev = idx.device
    with torch.cuda.device(dev):
        gq = torch.empty((max(nq, 1),), dtype=torch.int32, device=dev)
        gt = torch.empty((max(nq, 1),), dtype=torch.int32, device=dev)
        cnt = torch.empty((1,), dtype=torch.int32, device=dev)
        check(_lib.lib().isx_ratio_compact(_ptr(idx), _ptr(dist), nq, float(ratio), _ptr(gq), _ptr(gt), _ptr(cnt),
                                           _stream()))
        g = int(cnt.item())
    return gq[:g], gt[:g]


# ---------------------------------------------------------------------------------------------
# image space
# ---------------------------------------------------------------------------------------------
def _img_args(t, name="image"):
    """(ptr, h, w, c, pitch) of a uint8 device image (H,W) or (H,W,C); rows may be strided (views)."""
    _require_cuda(t, name)
    if t.dtype != torch.uint8 or t.ndim not in (2, 3):
        raise IsxError(f"{name} must be a uint8 (H,W) or (H,W,C) tensor")
    h, w = t.shape[0], t.shape[1]
    c = 1 if t.ndim == 2 else t.shape[2]
    if c not in (1, 3):
        raise IsxError(f"{name}: channels must be 1 or 3 (got {c}); the reference's pad_image raises too")
    if h > 0 and w > 0:
        ok = (t.stride(1) == c) and (t.ndim == 2 or t.stride(2) == 1) and (t.stride(0) >= w * c or h == 1)
        if not ok:
            raise IsxError(f"{name} must have contiguous pixels within a row (strides {t.stride()})")
    pitch = t.stride(0) if h > 1 else max(w * c, t.stride(0))
    return _ptr(t), h, w, c, pitch


def bbox_nonzero(img, acc=None):
    """get_corners_from_image on the device: int32[5] tensor {x0,y0,x1,y1,found} (no sync).  acc: optional
    accumulator from new_bbox() to reuse (the result is its first 5 ints)."""
    p, h, w, c, pitch = _img_args(img)
    with torch.cuda.device(img.device):
        if acc is None:
            acc = torch.empty((_lib.BBOX_INTS,), dtype=torch.int32, device=img.device)
        check(_lib.lib().isx_bbox_nonzero(p, h, w, c, pitch, _ptr(acc), _stream()))
    return acc[:5]


def clear_rects(img, rects):
    """Zero up to four rectangles (y0, y1, x0, x1) (half-open, pixels) of a uint8 device image in one launch."""
    rects = [r for r in rects if r[1] > r[0] and r[3] > r[2]]
    if not rects:
        return 0
    p, h, w, c, pitch = _img_args(img)
    a = np.array([(r[0], r[1], r[2] * c, r[3] * c) for r in rects], np.int32)
    with torch.cuda.device(img.device):
        check(_lib.lib().isx_clear_rects(p, pitch, h, w * c, a.ctypes.data_as(ctypes.c_void_p), len(rects), _stream()))
    return len(rects)


def bbox_to_corners(bb_host):
    """numpy int32[5] -> ((x0,y0),(x1,y1)) or False, the reference's get_corners_from_image contract."""
    if not bb_host[4]:
        return False
    return ((int(bb_host[0]), int(bb_host[1])), (int(bb_host[2]), int(bb_host[3])))


WARP_ARITH = {"f32": _lib.WARP_LINEAR_F32, "classic": _lib.WARP_CLASSIC}


def warp(src, M, perspective, dst_rect, src_xy=(0, 0), out=None, bbox=None, arith="f32", canvas=None):
    """cv2.warpAffine / cv2.warpPerspective (INTER_LINEAR, constant 0) of `src` placed at src_xy of a
    virtual canvas, producing canvas rectangle dst_rect = (x0, y0, w, h).  M: forward 2x3 / 3x3 float64.
    bbox: optional accumulator from new_bbox() (gray != 0 bbox of the produced pixels in its first 5 ints).
    arith: "f32" = OpenCV 4.11's float32 linear kernels (the reference's pinned version), "classic" = the
    1/32-px fixed-point remap of OpenCV <= 4.10; canvas = (W, H) of the full destination, needed by the
    classic perspective path (defaults to the extent of dst_rect)."""
    p, sh, sw, c, spitch = _img_args(src, "src")
    x0, y0, dw, dh = (int(v) for v in dst_rect)
    M = np.ascontiguousarray(M, np.float64)
    if M.shape != ((3, 3) if perspective else (2, 3)):
        raise IsxError(f"M has shape {M.shape}")
    with torch.cuda.device(src.device):
        if out is None:
            out = torch.empty((dh, dw) + ((c,) if src.ndim == 3 else ()), dtype=torch.uint8, device=src.device)
        po, oh, ow, oc, opitch = _img_args(out, "out")
        if (oh, ow, oc) != (dh, dw, c):
            raise IsxError("out does not match dst_rect")
        if arith not in WARP_ARITH:
            raise IsxError(f"arith must be one of {sorted(WARP_ARITH)}")
        cw, ch = canvas if canvas is not None else (x0 + dw, y0 + dh)
        check(_lib.lib().isx_warp_arith(p, sh, sw, c, spitch, int(src_xy[0]), int(src_xy[1]),
                                        M.ctypes.data_as(ctypes.c_void_p),
                                        _lib.MODEL_HOMOGRAPHY if perspective else _lib.MODEL_AFFINE_PARTIAL,
                                        po, opitch, x0, y0, dw, dh, WARP_ARITH[arith], int(cw), int(ch),
                                        _ptr(bbox), _stream()))
    return out


def warp_alpha_blend(src, M, perspective, dst_rect, ref, contrib, src_xy=(0, 0), out=None, bbox=None, arith="f32",
                     canvas=None):
    """warp() followed by alpha_blend(warped, ref, contrib) in one pass: `ref` is the view of the mosaic
    canvas under dst_rect, `out` (default: ref itself, in place) receives the merged pixels; bbox as in warp()."""
    p, sh, sw, c, spitch = _img_args(src, "src")
    x0, y0, dw, dh = (int(v) for v in dst_rect)
    M = np.ascontiguousarray(M, np.float64)
    if M.shape != ((3, 3) if perspective else (2, 3)):
        raise IsxError(f"M has shape {M.shape}")
    if arith not in WARP_ARITH:
        raise IsxError(f"arith must be one of {sorted(WARP_ARITH)}")
    pr, rh, rw, rc, rpitch = _img_args(ref, "ref")
    if out is None:
        out = ref
    po, oh, ow, oc, opitch = _img_args(out, "out")
    if (rh, rw, rc) != (dh, dw, c) or (oh, ow, oc) != (dh, dw, c):
        raise IsxError("ref / out do not match dst_rect")
    cw, ch = canvas if canvas is not None else (x0 + dw, y0 + dh)
    with torch.cuda.device(src.device):
        check(_lib.lib().isx_warp_alpha_blend(p, sh, sw, c, spitch, int(src_xy[0]), int(src_xy[1]),
                                              M.ctypes.data_as(ctypes.c_void_p),
                                              _lib.MODEL_HOMOGRAPHY if perspective else _lib.MODEL_AFFINE_PARTIAL,
                                              pr, rpitch, po, opitch, x0, y0, dw, dh, WARP_ARITH[arith], int(cw), int(ch),
                                              float(contrib), _ptr(bbox), _stream()))
    return out


# ---- opt-in depth-preserving mode: single-channel uint16 images (BASELINE configs[3]; Stitcher(depth16=True)) ----
def _img16_args(t, name="image"):
    _require_cuda(t, name)
    if t.dtype != torch.uint16 or t.ndim != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise IsxError(f"{name} must be a 2-D uint16 device tensor with contiguous rows")
    return _ptr(t), int(t.shape[0]), int(t.shape[1]), int(t.stride(0)) * 2


def warp16(src, M, perspective, dst_rect, src_xy=(0, 0)):
    """warp() for a 2-D uint16 image (float32 linear arithmetic): canvas rectangle dst_rect = (x0, y0, w, h) as a new
    (h, w) uint16 tensor."""
    p, sh, sw, spitch = _img16_args(src, "src")
    x0, y0, dw, dh = (int(v) for v in dst_rect)
    M = np.ascontiguousarray(M, np.float64)
    if M.shape != ((3, 3) if perspective else (2, 3)):
        raise IsxError(f"M has shape {M.shape}")
    with torch.cuda.device(src.device):
        out = torch.empty((dh, dw), dtype=torch.uint16, device=src.device)
        check(_lib.lib().isx_warp16(p, sh, sw, spitch, int(src_xy[0]), int(src_xy[1]), M.ctypes.data_as(ctypes.c_void_p),
                                    _lib.MODEL_HOMOGRAPHY if perspective else _lib.MODEL_AFFINE_PARTIAL,
                                    _ptr(out), 2 * dw, x0, y0, dw, dh, _stream()))
    return out


def alpha_blend16(warped, ref, contrib, out=None):
    """seamless_merge(image1=warped, image2=ref, contrib) on 2-D uint16 images; out defaults to ref itself (in place)."""
    pw, h, w, wp = _img16_args(warped, "warped")
    pr, h2, w2, rp = _img16_args(ref, "ref")
    if (h, w) != (h2, w2):
        raise IsxError("warped and ref differ in shape")
    out = ref if out is None else out
    po, h3, w3, op = _img16_args(out, "out")
    if (h3, w3) != (h, w):
        raise IsxError("out differs in shape")
    with torch.cuda.device(ref.device):
        check(_lib.lib().isx_alpha_blend16(pw, wp, pr, rp, po, op, h, w, float(contrib), _stream()))
    return out


def bbox_nonzero16(img):
    """get_corners_from_image of a 2-D uint16 device image: ((x0, y0), (x1, y1)) or False (synchronous).  A pixel is
    non-zero iff one of its two bytes is: the uint8 reduction runs on the byte view and the columns are halved."""
    if img.shape[0] == 0 or img.shape[1] == 0:
        return False
    bb = bbox_nonzero(img.view(torch.uint8)).cpu().numpy()
    if not bb[4]:
        return False
    return ((int(bb[0]) >> 1, int(bb[1])), ((int(bb[2]) + 1) >> 1, int(bb[3])))


def new_bbox(device):
    """A bbox accumulator (include/isx.h: ISX_BBOX_INTS int32 -- 5 result ints, then the caller-owned partial
    table the reduction kernels use), reset.  Read the result as acc[:5]."""
    with torch.cuda.device(device):
        bb = torch.empty((_lib.BBOX_INTS,), dtype=torch.int32, device=device)
        check(_lib.lib().isx_bbox_reset(_ptr(bb), _stream()))
    return bb


def reset_bbox(bb):
    check(_lib.lib().isx_bbox_reset(_ptr(bb), _stream()))
    return bb


def alpha_blend(warped, ref, contrib, out=None):
    """seamless_merge(image1=warped, image2=ref, contrib); out may be ref itself (in place)."""
    pw, h, w, c, wp = _img_args(warped, "warped")
    pr, h2, w2, c2, rp = _img_args(ref, "ref")
    if (h, w, c) != (h2, w2, c2):
        raise IsxError("warped and ref differ in shape")
    with torch.cuda.device(ref.device):
        if out is None:
            out = torch.empty_like(ref, memory_format=torch.contiguous_format)
        po, h3, w3, c3, op = _img_args(out, "out")
        check(_lib.lib().isx_alpha_blend(pw, wp, pr, rp, po, op, h, w, c, float(contrib), _stream()))
    return out


def gradient_blend_roi(warped_roi, ref_roi, feather=10, ext=None):
    """ROI body of seamless_gradient_merge; ref_roi is updated in place.  ext: optional int32 scratch of
    2*h + 2*w elements (allocated when absent)."""
    pw, h, w, c, wp = _img_args(warped_roi, "warped_roi")
    pr, h2, w2, c2, rp = _img_args(ref_roi, "ref_roi")
    if (h, w, c) != (h2, w2, c2):
        raise IsxError("warped_roi and ref_roi differ in shape")
    with torch.cuda.device(ref_roi.device):
        if ext is None:
            ext = torch.empty((2 * h + 2 * w,), dtype=torch.int32, device=ref_roi.device)
        elif ext.numel() < 2 * h + 2 * w or ext.dtype != torch.int32:
            raise IsxError("ext scratch too small")
        check(_lib.lib().isx_gradient_blend(pw, wp, pr, rp, h, w, c, int(feather), _ptr(ext), _stream()))
    return ref_roi


def frame_download(ctx, img, out_hw=None, offset=(0, 0)):
    """uint8 device image (possibly a strided view) -> numpy VIEW of the ctx's pinned block holding a contiguous copy
    (one pitched DMA).  out_hw = (H, W) and offset = (y, x): the copy is placed at that position of a zero-filled H x W
    image (a slice of the reference's padded canvas that reaches beyond the mosaic).  Valid until the next
    frame_match / frame_transform / frame_download on this ctx."""
    p, h, w, c, pitch = _img_args(img)
    oh, ow = (h, w) if out_hw is None else (int(out_hw[0]), int(out_hw[1]))
    shape = (oh, ow) + tuple(img.shape[2:])
    out = ctypes.c_void_p()
    with torch.cuda.device(img.device):
        check(_lib.lib().isx_frame_download(ctx._h, p, pitch, h, w * c, oh, ow * c, int(offset[0]), int(offset[1]) * c,
                                            ctypes.byref(out), _stream()))
    n = oh * ow * c
    if n == 0:
        return np.zeros(shape, np.uint8)
    buf = (ctypes.c_uint8 * n).from_address(out.value)
    return np.frombuffer(buf, np.uint8).reshape(shape)


class FrameCtx:
    """isx_ctx (include/isx.h): the per-device workspace of the per-frame pipeline."""

    def __init__(self, device):
        self.device = torch.device(device)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(_lib.lib().isx_ctx_create(ctypes.byref(h)))
        self._h = h
        self._res = np.zeros((_lib.FRAME_RESULT_INTS,), np.int32)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().isx_ctx_destroy(h)
            except Exception:  # interpreter shutdown
                pass


def frame_match(ctx, q, t, ratio, hamming=False):
    """isx_frame_match: cv2.BFMatcher(L2 | HAMMING).knnMatch(q, t, k=2) + the Lowe ratio loop for one pair in ONE
    synchronous library call.  q, t: (n, d) numpy uint8 / float32 (integer valued).  Returns (good_q, good_t)
    int32 numpy arrays in query order; raises NotIntegerValuedError / NormLimitError like prepare_descriptors."""
    q = np.ascontiguousarray(q)
    t = np.ascontiguousarray(t)
    if q.ndim != 2 or t.ndim != 2 or q.shape[1] != t.shape[1] or q.dtype != t.dtype:
        raise IsxError("descriptor arrays must be 2-D and share width and dtype")
    if q.dtype not in (np.uint8, np.float32):
        raise IsxError(f"descriptor dtype must be uint8 or float32 (got {q.dtype})")
    nq, nt, d = q.shape[0], t.shape[0], q.shape[1]
    gq = np.empty((nq,), np.int32)
    gt = np.empty((nq,), np.int32)
    n_good, status = ctypes.c_int32(0), ctypes.c_int32(0)
    with torch.cuda.device(ctx.device):
        check(_lib.lib().isx_frame_match(ctx._h, q.ctypes.data_as(ctypes.c_void_p), nq, t.ctypes.data_as(ctypes.c_void_p), nt, d,
                                         _lib.U8 if q.dtype == np.uint8 else _lib.F32,
                                         _lib.NORM_HAMMING if hamming else _lib.NORM_L2, float(ratio),
                                         gq.ctypes.data_as(ctypes.c_void_p), gt.ctypes.data_as(ctypes.c_void_p),
                                         ctypes.byref(n_good), ctypes.byref(status), _stream()))
    if status.value == 1:
        raise NotIntegerValuedError("float descriptors are not integer valued in [0,255]")
    if status.value == 2:
        raise NormLimitError("descriptor norms too large for the exact integer path")
    return gq[:n_good.value], gt[:n_good.value]


def frame_transform(ctx, src_pts, dst_pts, perspective):
    """isx_frame_transform: cv2.estimateAffinePartial2D / cv2.findHomography(RANSAC, 5.0) in ONE synchronous library
    call.  Returns (M float64 (2,3)|(3,3) or None, mask uint8 (n,1)) like estimate_transform."""
    s = np.ascontiguousarray(np.asarray(src_pts, np.float32).reshape(-1, 2))
    d = np.ascontiguousarray(np.asarray(dst_pts, np.float32).reshape(-1, 2))
    n = len(s)
    if len(d) != n:
        raise IsxError("src_pts and dst_pts differ in length")
    M = np.zeros((9,), np.float64)
    mask = np.zeros((max(n, 1),), np.uint8)
    found = ctypes.c_int32(0)
    with torch.cuda.device(ctx.device):
        check(_lib.lib().isx_frame_transform(ctx._h, s.ctypes.data_as(ctypes.c_void_p), d.ctypes.data_as(ctypes.c_void_p), n,
                                             _lib.MODEL_HOMOGRAPHY if perspective else _lib.MODEL_AFFINE_PARTIAL,
                                             M.ctypes.data_as(ctypes.c_void_p), mask.ctypes.data_as(ctypes.c_void_p),
                                             ctypes.byref(found), _stream()))
    if not found.value:
        return None, np.zeros((n, 1), np.uint8)
    return (M.reshape(3, 3) if perspective else M[:6].reshape(2, 3)).copy(), mask[:n].reshape(-1, 1)


def _frame_result_dict(r):
    v = r.tolist()
    return {"wbbox": tuple(v[0:4]) if v[4] else False, "roi": tuple(v[5:9]), "mbbox": tuple(v[9:13]), "mfound": bool(v[13]),
            "zeroed": v[14], "dirty": v[15], "rows_ok": bool(v[16]), "done": bool(v[17])}


def frame_result(ctx):
    """Result block of the last frame_blend(..., fetch=False) on this ctx (synchronises the current stream)."""
    r = ctx._res
    with torch.cuda.device(ctx.device):
        check(_lib.lib().isx_frame_result(ctx._h, r.ctypes.data_as(ctypes.c_void_p), _stream()))
    return _frame_result_dict(r)


def frame_blend(ctx, src, M, perspective, src_xy, canvas_wh, arena_t, canvas_to_arena, q_rect, blender,
                contrib=0.5, feather=10, arith="f32", fetch=True, clip_rows=None):
    """isx_frame_blend: warp `src` (pasted at src_xy of the virtual canvas) by M and blend it IN PLACE into the
    mosaic held by arena_t (canvas (X, Y) is arena pixel (X + canvas_to_arena[0], Y + canvas_to_arena[1])), only over
    the canvas rectangle q_rect = (x0, y0, w, h).  Synchronous; returns the int32 result block as a dict of python
    ints / tuples.  fetch=False only queues the kernels (returns None; frame_result() reads the block later).
    clip_rows = (y0, y1): arena_t holds only the canvas rows [y0, y1) of the mosaic (one strip of a mosaic partitioned over
    several GPUs): only those rows are read or written and only they must be addressable in arena_t."""
    p, sh, sw, c, spitch = _img_args(src, "src")
    pa, ah, aw, ac, apitch = _img_args(arena_t, "arena")
    if ac != c:
        raise IsxError("arena and frame differ in channels")
    qx0, qy0, qw, qh = (int(v) for v in q_rect)
    ax0, ay0 = qx0 + int(canvas_to_arena[0]), qy0 + int(canvas_to_arena[1])
    cy0, cy1 = (0, int(canvas_wh[1])) if clip_rows is None else (int(clip_rows[0]), int(clip_rows[1]))
    ry0, ry1 = max(qy0, cy0), min(qy0 + qh, cy1)  # rows of Q this arena must hold
    if qw > 0 and ry1 > ry0 and (ax0 < 0 or ax0 + qw > aw or ry0 + int(canvas_to_arena[1]) < 0 or ry1 + int(canvas_to_arena[1]) > ah):
        raise IsxError("frame_blend: the rows of Q inside the clip lie outside the arena")
    M = np.ascontiguousarray(M, np.float64)
    if M.shape != ((3, 3) if perspective else (2, 3)):
        raise IsxError(f"M has shape {M.shape}")
    mq = ctypes.c_void_p(arena_t.data_ptr() + ay0 * apitch + ax0 * c)  # (may lie outside the arena when clipped: never dereferenced there)
    r = ctx._res
    with torch.cuda.device(src.device):
        check(_lib.lib().isx_frame_blend(ctx._h, p, sh, sw, c, spitch, int(src_xy[0]), int(src_xy[1]),
                                         M.ctypes.data_as(ctypes.c_void_p),
                                         _lib.MODEL_HOMOGRAPHY if perspective else _lib.MODEL_AFFINE_PARTIAL,
                                         WARP_ARITH[arith], int(canvas_wh[0]), int(canvas_wh[1]), mq, apitch, qx0, qy0, qw, qh, cy0, cy1,
                                         int(blender), float(contrib), int(feather),
                                         r.ctypes.data_as(ctypes.c_void_p) if fetch else None, _stream()))
    return _frame_result_dict(r) if fetch else None


def frame_warped_view(ctx, q_rect, c, ndim3):
    """The warped image staged by the last gradient-path frame_blend, as a uint8 tensor view of Q's shape."""
    out = ctypes.c_void_p()
    check(_lib.lib().isx_frame_warped(ctx._h, ctypes.byref(out)))
    qw, qh = int(q_rect[2]), int(q_rect[3])
    n = qw * qh * c

    class _Raw:  # __cuda_array_interface__ holder
        pass
    raw = _Raw()
    raw.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (out.value, False), "version": 2}
    t = torch.as_tensor(raw, device=ctx.device)
    return t.view((qh, qw, c) if ndim3 else (qh, qw))


def copy2d(src, dst):
    ps, h, w, c, sp = _img_args(src, "src")
    pd, h2, w2, c2, dp = _img_args(dst, "dst")
    if (h, w, c) != (h2, w2, c2):
        raise IsxError("copy2d shape mismatch")
    with torch.cuda.device(dst.device):
        check(_lib.lib().isx_copy2d(ps, sp, pd, dp, h, w * c, _stream()))
    return dst


# ---------------------------------------------------------------------------------------------
# on-GPU ORB-style detection (opt-in; SURVEY 8f-4, reference stitcher.py:92-100)
# ---------------------------------------------------------------------------------------------
ORB_TILE, ORB_SLOTS, ORB_EDGE = 32, 64, 16


def orb_workspace_bytes(h, w):
    n = int(_lib.lib().isx_orb_workspace_bytes(int(h), int(w)))
    if n <= 0:
        raise IsxError(f"orb: image size {w}x{h} out of range")
    return n


def orb_candidates(img, workspace, tiles, fast_threshold=20, clean=16):
    """isx_orb_candidates: queue gray / blur / FAST + Harris / NMS for a device image.  `workspace` = uint8 device tensor
    of at least orb_workspace_bytes(h, w); `tiles` = int32 device tensor of n_tiles * (1 + 2 * 64) ints laid out as
    [counts | positions | keys].  Nothing is synchronised."""
    p, h, w, c, pitch = _img_args(img, "image")
    nt = ((h + ORB_TILE - 1) // ORB_TILE) * ((w + ORB_TILE - 1) // ORB_TILE)
    if workspace.numel() < orb_workspace_bytes(h, w) or tiles.numel() < nt * (1 + 2 * ORB_SLOTS):
        raise IsxError("orb: workspace or tile buffer too small")
    base = tiles.data_ptr()
    with torch.cuda.device(img.device):
        check(_lib.lib().isx_orb_candidates(p, h, w, c, pitch, int(fast_threshold), int(clean), _ptr(workspace),
                                            ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * nt),
                                            ctypes.c_void_p(base + 4 * nt * (1 + ORB_SLOTS)), _stream()))
    return nt


def orb_select_describe(workspace, h, w, tiles, nfeatures, pattern, sel_out, des_out, ang_out):
    """isx_orb_select_describe: device-side selection of the nfeatures best survivors + their descriptors (2 kernels, no
    synchronisation).  `tiles` as filled by orb_candidates; sel_out int32 [2 + 2*nfeatures]."""
    nt = ((h + ORB_TILE - 1) // ORB_TILE) * ((w + ORB_TILE - 1) // ORB_TILE)
    base = tiles.data_ptr()
    with torch.cuda.device(tiles.device):
        check(_lib.lib().isx_orb_select_describe(_ptr(workspace), int(h), int(w), ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * nt),
                                                 ctypes.c_void_p(base + 4 * nt * (1 + ORB_SLOTS)), int(nfeatures), _ptr(pattern),
                                                 _ptr(sel_out), _ptr(des_out), _ptr(ang_out), _stream()))


def orb_describe(workspace, h, w, pos, pattern, des_out, ang_out):
    """isx_orb_describe for n = len(pos) selected positions ((y << 16) | x, int32 device tensor)."""
    n = int(pos.numel())
    with torch.cuda.device(pos.device):
        check(_lib.lib().isx_orb_describe(_ptr(workspace), int(h), int(w), _ptr(pos), n, _ptr(pattern), _ptr(des_out), _ptr(ang_out),
                                          _stream()))


# ---------------------------------------------------------------------------------------------
# robust transform (Stitcher.get_transform, reference stitcher.py:135-141)
# ---------------------------------------------------------------------------------------------
# cv2 defaults of the two calls the reference makes
RANSAC_DEFAULTS = {
    False: dict(thr=3.0, max_iters=2000, confidence=0.99, refine_iters=10),   # estimateAffinePartial2D
    True: dict(thr=5.0, max_iters=2000, confidence=0.995, refine_iters=10),   # findHomography(RANSAC, 5.0)
}


def ransac_sample(src, dst, perspective, max_iters=2000):
    """Host: the subset sequence cv's RANSAC would draw (k, 4|2) int32."""
    model = _lib.MODEL_HOMOGRAPHY if perspective else _lib.MODEL_AFFINE_PARTIAL
    s = np.ascontiguousarray(np.asarray(src, np.float32).reshape(-1, 2))
    d = np.ascontiguousarray(np.asarray(dst, np.float32).reshape(-1, 2))
    mp = 4 if perspective else 2
    out = np.zeros((max(max_iters, 1), mp), np.int32)
    k = check(_lib.lib().isx_ransac_sample(model, s.ctypes.data_as(ctypes.c_void_p), d.ctypes.data_as(ctypes.c_void_p),
                                           len(s), max_iters, out.ctypes.data_as(ctypes.c_void_p)))
    return out[:k]


def ransac_hypotheses(src_dev, dst_dev, subsets_dev, perspective):
    model = _lib.MODEL_HOMOGRAPHY if perspective else _lib.MODEL_AFFINE_PARTIAL
    k = subsets_dev.shape[0]
    dev = src_dev.device
    with torch.cuda.device(dev):
        models = torch.empty((k, 9 if perspective else 6), dtype=torch.float64, device=dev)
        valid = torch.empty((k,), dtype=torch.int32, device=dev)
        check(_lib.lib().isx_ransac_hypotheses(model, _ptr(src_dev), _ptr(dst_dev), src_dev.shape[0], _ptr(subsets_dev),
                                               k, _ptr(models), _ptr(valid), _stream()))
    return models, valid


def ransac_score(models_dev, valid_dev, src_dev, dst_dev, perspective, thr):
    model = _lib.MODEL_HOMOGRAPHY if perspective else _lib.MODEL_AFFINE_PARTIAL
    k = models_dev.shape[0]
    dev = src_dev.device
    with torch.cuda.device(dev):
        counts = torch.empty((k,), dtype=torch.int32, device=dev)
        check(_lib.lib().isx_ransac_score(model, _ptr(models_dev), k, _ptr(valid_dev), _ptr(src_dev), _ptr(dst_dev),
                                          src_dev.shape[0], float(thr), _ptr(counts), _stream()))
    return counts


def ransac_select(counts, valid, n_points, perspective, confidence, max_iters):
    model = _lib.MODEL_HOMOGRAPHY if perspective else _lib.MODEL_AFFINE_PARTIAL
    counts = np.ascontiguousarray(counts, np.int32)
    valid = np.ascontiguousarray(valid, np.int32)
    best = ctypes.c_int32(-1)
    iters = ctypes.c_int32(0)
    check(_lib.lib().isx_ransac_select(model, counts.ctypes.data_as(ctypes.c_void_p),
                                       valid.ctypes.data_as(ctypes.c_void_p), len(counts), n_points,
                                       float(confidence), max_iters, ctypes.byref(best), ctypes.byref(iters)))
    return best.value, iters.value


def ransac_refine(M_dev, src_dev, dst_dev, perspective, thr, refit, max_iters):
    """In place on M_dev (9|6 float64).  Returns (mask uint8 (n,), info int32 (1,)) device tensors."""
    model = _lib.MODEL_HOMOGRAPHY if perspective else _lib.MODEL_AFFINE_PARTIAL
    dev = src_dev.device
    n = src_dev.shape[0]
    with torch.cuda.device(dev):
        mask = torch.empty((n,), dtype=torch.uint8, device=dev)
        info = torch.zeros((1,), dtype=torch.int32, device=dev)
        check(_lib.lib().isx_ransac_refine(model, _ptr(M_dev), _ptr(src_dev), _ptr(dst_dev), n, float(thr),
                                           int(bool(refit)), int(max_iters), _ptr(mask), _ptr(info), _stream()))
    return mask, info


def estimate_transform(src_pts, dst_pts, perspective, device=None, return_trace=False):
    """cv2.estimateAffinePartial2D(src, dst) (perspective=False) or
    cv2.findHomography(src, dst, cv2.RANSAC, 5.0) (perspective=True).
    src_pts/dst_pts: host float32 arrays of n points (any shape ending in 2, e.g. (n,1,2)).
    Returns (M float64 (2,3)|(3,3) or None, mask uint8 (n,1))."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    cfg = RANSAC_DEFAULTS[bool(perspective)]
    s = np.ascontiguousarray(np.asarray(src_pts, np.float32).reshape(-1, 2))
    d = np.ascontiguousarray(np.asarray(dst_pts, np.float32).reshape(-1, 2))
    n = len(s)
    if len(d) != n:
        raise IsxError("src_pts and dst_pts differ in length")
    mp = 4 if perspective else 2
    shape = (3, 3) if perspective else (2, 3)
    none = (None, np.zeros((n, 1), np.uint8))
    if n < mp:
        return none
    subsets = ransac_sample(s, d, perspective, cfg["max_iters"])
    if len(subsets) == 0:
        return none
    with torch.cuda.device(device):
        sd = torch.from_numpy(s).to(device)
        dd = torch.from_numpy(d).to(device)
        sub_dev = torch.from_numpy(np.ascontiguousarray(subsets)).to(device)
        models, valid = ransac_hypotheses(sd, dd, sub_dev, perspective)
        if n == mp:
            # cv: direct solve, mask of ones, no refinement (findHomography: npoints == 4; affine: count == 2)
            ok = bool(valid[0].item())
            M = models[0].cpu().numpy().reshape(shape) if ok else None
            return (M, np.ones((n, 1), np.uint8)) if ok else none
        counts = ransac_score(models, valid, sd, dd, perspective, cfg["thr"])
        counts_h = counts.cpu().numpy()
        valid_h = valid.cpu().numpy()
        best, iters = ransac_select(counts_h, valid_h, n, perspective, cfg["confidence"], cfg["max_iters"])
        if best < 0:
            return none
        M_dev = models[best].clone()
        mask, info = ransac_refine(M_dev, sd, dd, perspective, cfg["thr"], refit=perspective,
                                   max_iters=cfg["refine_iters"])
        M = M_dev.cpu().numpy().reshape(shape)
        mask_h = mask.cpu().numpy().reshape(-1, 1)
    if return_trace:
        return M, mask_h, dict(models=models, valid=valid_h, counts=counts_h, best=best, iters=iters,
                               subsets=subsets)
    return M, mask_h

"""All-pairs relation graph of ArbitraryStitcher.get_collections (reference stitcher.py:280-304),
sharded over the GPUs of one node.

The reference evaluates every ORDERED pair (i, j), i != j (optionally |i-j| <= consecutive_range,
stitcher.py:289-291) with knnMatch + ratio test and unions i, j when len(good) >= min_match_count.
Here the unordered pair list {i<j} is dealt round-robin to the ranks (pair p -> rank p % world),
every rank runs both directions of its pairs in ONE kernel launch, and only the int32 good-match
counts travel back.  Collectives (RCCL when the tensors are on GPU, gloo in the CPU tests):
    all_gather  of the per-rank descriptor shards   (only when detection was sharded)
    gather      of the per-pair counts to rank 0
"""
import numpy as np
import torch

from ._lib import check, lib
from .utils.structs import DynamicConnectivity


def unordered_pairs(n, consecutive_range=None):
    """(p,2) int64 array of i<j pairs in the reference's visiting order restricted to i<j."""
    i, j = np.triu_indices(n, 1)
    if consecutive_range is not None:
        keep = (j - i) <= consecutive_range
        i, j = i[keep], j[keep]
    return np.stack([i, j], 1).astype(np.int64)


def shard_pairs(pairs, rank, world):
    """pair p -> rank p % world (uniform cost mix on every rank)."""
    return pairs[rank::world]


def overlap_shards(n, world, consecutive_range=None):
    """The pair list split so that the descriptor all_gather can hide behind work: images are held in `world` equal shards
    (image k on rank k // (n // world), as after sharded detection); a rank's LOCAL pairs (both images its own) need no
    gather and run first, the CROSS pairs are dealt round-robin as in shard_pairs.  Returns (local, cross): two lists of
    `world` (p, 2) arrays of global image indices; every unordered pair appears exactly once, and every rank gets the same
    number of each kind (+-1 cross pair)."""
    if n % world:
        raise ValueError("overlap_shards needs equal image shards")
    pairs = unordered_pairs(n, consecutive_range)
    per = n // world
    own_i, own_j = pairs[:, 0] // per, pairs[:, 1] // per
    loc = own_i == own_j
    cross = pairs[~loc]
    return [pairs[loc & (own_i == r)] for r in range(world)], [cross[r::world] for r in range(world)]


def directed_items(pairs, starts, counts):
    """Both directions of every unordered pair as kernel work items (q_start, nq, t_start, nt).
    Row 2p is i->j, row 2p+1 is j->i."""
    i, j = pairs[:, 0], pairs[:, 1]
    it = np.empty((2 * len(pairs), 4), np.int32)
    it[0::2, 0], it[0::2, 1], it[0::2, 2], it[0::2, 3] = starts[i], counts[i], starts[j], counts[j]
    it[1::2, 0], it[1::2, 1], it[1::2, 2], it[1::2, 3] = starts[j], counts[j], starts[i], counts[i]
    return it


def pair_items(pairs, starts, counts):
    """One work item per unordered pair (i, j) for ops.knn2_l2_pairs: (starts[i], counts[i], starts[j], counts[j]); its
    result row is (len(good) of i -> j, len(good) of j -> i), the layout of directed_items' counts viewed as (p, 2)."""
    i, j = pairs[:, 0], pairs[:, 1]
    return np.stack([starts[i], counts[i], starts[j], counts[j]], 1).astype(np.int32)


def components_from_counts(n, pairs, counts_ij, counts_ji, min_match_count, n_union_find=None):
    """Host union-find exactly as stitcher.py:283-300.  The reference visits the ORDERED pairs (i, j) in
    lexicographic order and unions where len(good) >= min_match_count; the same unions are replayed here in
    the same order (so `connections` ends in the reference's internal state, not only with the same
    component lists), but only over the pairs that pass -- a few thousand of 261,632 at BASELINE's size.
    n_union_find: the reference sizes the union-find BEFORE dropping empty images (stitcher.py:283)."""
    dc = DynamicConnectivity(n if n_union_find is None else n_union_find)
    pairs = np.asarray(pairs).reshape(-1, 2)
    e_ij = pairs[np.flatnonzero(np.asarray(counts_ij) >= min_match_count)]
    e_ji = pairs[np.flatnonzero(np.asarray(counts_ji) >= min_match_count)]
    first = np.concatenate([e_ij[:, 0], e_ji[:, 1]])
    second = np.concatenate([e_ij[:, 1], e_ji[:, 0]])
    order = np.lexsort((second, first))
    first = np.ascontiguousarray(first[order], np.int32)
    second = np.ascontiguousarray(second[order], np.int32)
    parent = np.arange(dc.n, dtype=np.int32)
    size = np.ones(dc.n, np.int32)
    check(lib().isx_union_find(dc.n, first.ctypes.data, second.ctypes.data, len(first), parent.ctypes.data, size.ctypes.data))
    dc.parent, dc.size = parent.tolist(), size.tolist()
    comps = dc.get_connected_components()
    comps = sorted(comps, key=len, reverse=True)
    return comps, dc


def all_gather_descriptors(local_des, group=None, always=False):
    """local_des: (n_local, kp, d) uint8 tensor (equal shapes on every rank).  Returns (n, kp, d).
    always=True issues the collective even in a world of one (used to exercise the RCCL path on one GPU)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1 and not always:
        return local_des
    # concatenated (not stacked) output shape: the form both RCCL and gloo accept
    dev = local_des.device
    if _gloo_on_device(local_des, group):
        local_des = local_des.cpu()
    out = torch.empty((world * local_des.shape[0],) + tuple(local_des.shape[1:]), dtype=local_des.dtype,
                      device=local_des.device)
    dist.all_gather_into_tensor(out, local_des.contiguous(), group=group)
    return out.to(dev)


def all_gather_prepared(Pl, starts, counts, group=None, always=False, async_op=False):
    """Every rank has prepared (ops.prepare_descriptor_tensor, optionally ops.sort_rows_by_norm) the rows of ITS images --
    equal row counts on every rank, images in rank order; returns the PreparedDescriptors of all images.  ONE all_gather of
    one uint8 buffer per rank ([desc | cbase | nrm | cin] as bytes: uint8 is the element type every backend carries, and a
    step keeps a single collective on the way in), then one copy per field to make the fields contiguous again.
    async_op=True: the collective is only started (on the backend's own stream) and a function is returned that waits for
    it -- the current stream waits, not the host -- and hands over the result; work launched in between overlaps it."""
    from . import ops
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1 and not always:
        P = ops.PreparedDescriptors(Pl.desc, Pl.cbase, Pl.nrm, Pl.cin, starts, counts, Pl.d, Pl.d_pad, Pl.max_norm2, Pl.wide)
        return (lambda: P) if async_op else P
    rows = Pl.desc.shape[0]
    parts = [Pl.desc.contiguous().view(torch.uint8).reshape(-1)] + \
            [t.contiguous().view(torch.uint8).reshape(-1) for t in (Pl.cbase, Pl.nrm, Pl.cin)]
    sizes = [int(t.numel()) for t in parts]
    buf = torch.cat(parts)
    dev = buf.device
    if _gloo_on_device(buf, group):
        buf = buf.cpu()  # (rehearsal backend: host tensors all the way, see _gloo_on_device)
    out = torch.empty((world, buf.numel()), dtype=torch.uint8, device=buf.device)   # (world, bytes per rank)
    # (input (1, bytes) -> output (world, bytes): the concatenated form that both RCCL and gloo accept)
    work = dist.all_gather_into_tensor(out, buf.view(1, -1), group=group, async_op=async_op)
    offs = np.concatenate([[0], np.cumsum(sizes)])

    def finish():
        if async_op:
            work.wait()
        keep = buf  # noqa: F841  (the send buffer lives until the collective has been waited for)
        got = out.to(dev)
        field = lambda k: got[:, int(offs[k]):int(offs[k + 1])].contiguous().view(-1)
        desc = field(0).view(torch.int8).view(world * rows, Pl.d_pad)
        cbase, nrm, cin = (field(k).view(torch.int32) for k in (1, 2, 3))
        return ops.PreparedDescriptors(desc, cbase, nrm, cin, starts, counts, Pl.d, Pl.d_pad, Pl.max_norm2, Pl.wide)
    return finish if async_op else finish()


def gather_rank_lists(local_counts, max_len, rank, world, group=None):
    """local_counts: (p_r, 2) int32, p_r <= max_len, one list per rank in any partition of the pairs.  Rank 0 returns
    (world, max_len, 2) -- row r = rank r's list, zero-padded; others None."""
    import torch.distributed as dist
    pad = torch.zeros((max_len, 2), dtype=torch.int32, device=local_counts.device)
    pad[:local_counts.shape[0]] = local_counts
    if world == 1:
        return pad[None]
    dev = pad.device
    if _gloo_on_device(pad, group):
        pad = pad.cpu()
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0, group=group)
    return torch.stack(bufs, 0).to(dev) if rank == 0 else None


def _gloo_on_device(t, group):
    """gloo (the rehearsal backend: several ranks on one GPU) carries device tensors through the host by itself and was seen to
    stall on gathers of a few hundred KB of them; stage such a gather through a host tensor explicitly.  RCCL is not touched."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) == "gloo"


def gather_counts(local_counts, n_pairs_total, rank, world, group=None, always=False):
    """local_counts: (p_local, 2) int32 for pairs rank::world.  Rank 0 returns (n_pairs_total, 2); others None."""
    import torch.distributed as dist
    if world == 1 and not always:
        return local_counts
    per = (n_pairs_total + world - 1) // world
    dev = local_counts.device
    pad = torch.zeros((per, 2), dtype=torch.int32, device=dev)
    pad[:local_counts.shape[0]] = local_counts
    if _gloo_on_device(pad, group):
        pad = pad.cpu()
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0, group=group)
    if rank != 0:
        return None
    allc = torch.stack(bufs, 1).reshape(-1, 2).to(dev)  # row q*world + r  <- rank r's q-th pair = pair q*world + r
    return allc[:n_pairs_total]

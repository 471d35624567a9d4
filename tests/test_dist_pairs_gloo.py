"""world_size-2 gloo test (CPU) of the multi-GPU get_collections plumbing in image_stitcher_amd.pairs:
pair sharding (pair p -> rank p % world), all_gather of descriptor shards, gather of the per-pair counts,
host union-find.  The per-pair counts come from the CPU oracle here (there is no GPU in this test); on the
GPU box the same functions run over RCCL with the HIP kernel producing the counts (bench.py --gpus N)."""
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


def _worker(rank, world, port, n_img, kp, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from image_stitcher_amd import pairs as prs
        from image_stitcher_amd.synthetic import overlapping_crops_descriptors
        from oracle import orc
        des, scene_of = overlapping_crops_descriptors(seed=5, n_scenes=2, imgs_per_scene=n_img // 2, kp=kp, step=kp // 6)
        per = n_img // world
        local = torch.from_numpy(des[rank * per:(rank + 1) * per].copy())       # this rank's "detected" shard
        full = prs.all_gather_descriptors(local).numpy()
        assert np.array_equal(full, des)
        all_pairs = prs.unordered_pairs(n_img)
        mine = prs.shard_pairs(all_pairs, rank, world)
        c = np.array([[orc.match_count(full[i], full[j], 0.65), orc.match_count(full[j], full[i], 0.65)] for i, j in mine],
                     np.int32).reshape(-1, 2)
        allc = prs.gather_counts(torch.from_numpy(c), len(all_pairs), rank, world)
        if rank == 0:
            comps, _ = prs.components_from_counts(n_img, all_pairs, allc[:, 0].numpy(), allc[:, 1].numpy(), 5)
            np.save(out_path, np.array([allc.numpy(), np.array(comps, dtype=object)], dtype=object), allow_pickle=True)
        else:
            assert allc is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_get_collections_matches_single_process(tmp_path, orc, world):
    from image_stitcher_amd import pairs as prs
    from image_stitcher_amd.synthetic import overlapping_crops_descriptors
    n_img, kp = 6, 96
    out = str(tmp_path / "r0.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_img, kp, out), nprocs=world, join=True)
    allc, comps = np.load(out, allow_pickle=True)  # written by this test itself
    des, scene_of = overlapping_crops_descriptors(seed=5, n_scenes=2, imgs_per_scene=n_img // 2, kp=kp, step=kp // 6)
    all_pairs = prs.unordered_pairs(n_img)
    exp = np.array([[orc.match_count(des[i], des[j], 0.65), orc.match_count(des[j], des[i], 0.65)] for i, j in all_pairs], np.int32)
    assert np.array_equal(np.asarray(allc, np.int32), exp)
    exp_comps, _ = prs.components_from_counts(n_img, all_pairs, exp[:, 0], exp[:, 1], 5)
    assert [list(c) for c in comps] == exp_comps
    truth = sorted(sorted(np.nonzero(scene_of == s)[0].tolist()) for s in range(2))
    assert sorted(sorted(c) for c in exp_comps) == truth


def test_pair_sharding_covers_every_pair_once():
    from image_stitcher_amd import pairs as prs
    p = prs.unordered_pairs(37)
    assert len(p) == 37 * 36 // 2
    for world in (1, 2, 4, 8):
        got = np.concatenate([prs.shard_pairs(p, r, world) for r in range(world)])
        assert len(got) == len(p) and len({(a, b) for a, b in got}) == len(p)
        sizes = [len(prs.shard_pairs(p, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
    q = prs.unordered_pairs(10, consecutive_range=2)
    assert all(0 < b - a <= 2 for a, b in q) and len(q) == 9 + 8
    it = prs.directed_items(p[:3], np.arange(37) * 10, np.full(37, 10))
    assert it.shape == (6, 4) and (it[0] == [0, 10, 10, 10]).all() and (it[1] == [10, 10, 0, 10]).all()


def test_overlap_shards_split_the_pair_list_into_own_and_cross_pairs():
    """pairs.overlap_shards (bench.py at N > 1): every unordered pair exactly once; a rank's own pairs lie inside its image
    shard and number C(n / N, 2); the cross pairs are spread evenly."""
    from image_stitcher_amd import pairs as prs
    for n, world in ((16, 2), (24, 4), (64, 8), (6, 3), (8, 8)):
        loc, cross = prs.overlap_shards(n, world)
        per = n // world
        assert len(loc) == len(cross) == world
        seen = set()
        for r in range(world):
            assert len(loc[r]) == per * (per - 1) // 2
            assert all(r * per <= a < b < (r + 1) * per for a, b in loc[r])
            assert all(a < b and a // per != b // per for a, b in cross[r])
            seen |= {(int(a), int(b)) for a, b in loc[r]} | {(int(a), int(b)) for a, b in cross[r]}
        assert len(seen) == n * (n - 1) // 2 == sum(len(a) + len(b) for a, b in zip(loc, cross))
        sizes = [len(c) for c in cross]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        prs.overlap_shards(10, 4)


def _prepared_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from image_stitcher_amd import ops, pairs as prs
        per, kp, d_pad = 3, 50, 128
        n = per * world
        rng = np.random.default_rng(100 + rank)
        rows = per * kp
        # a rank's PREPARED shard (on the GPU box: ops.prepare_descriptor_tensor + sort_rows_by_norm of its own images)
        Pl = ops.PreparedDescriptors(torch.from_numpy(rng.integers(-128, 128, (rows, d_pad), dtype=np.int8)),
                                     torch.from_numpy(rng.integers(0, 2 ** 31 - 1, (rows,)).astype(np.int32)),
                                     torch.from_numpy(rng.integers(0, 2 ** 22, (rows,)).astype(np.int32)),
                                     torch.from_numpy(rng.integers(-2 ** 22, 2 ** 22, (rows,)).astype(np.int32)),
                                     np.arange(per) * kp, np.full(per, kp), 128, d_pad, 12345, False)
        starts, counts = np.arange(n) * kp, np.full(n, kp)
        P = prs.all_gather_prepared(Pl, starts, counts)
        # the asynchronous form (bench.py at N > 1: the gather hides behind the rank's own-shard pairs) hands over the same set
        landed = prs.all_gather_prepared(Pl, starts, counts, async_op=True)
        _own = (Pl.desc.to(torch.int32).sum(), Pl.nrm.sum())   # (work queued while the collective travels)
        Pa = landed()
        for a, b in ((P.desc, Pa.desc), (P.cbase, Pa.cbase), (P.nrm, Pa.nrm), (P.cin, Pa.cin)):
            assert torch.equal(a, b)
        # ... and the per-rank result lists come back to rank 0 row by row
        mine = torch.full((2 + rank, 2), rank + 1, dtype=torch.int32)
        got = prs.gather_rank_lists(mine, 2 + world, rank, world)
        if rank == 0:
            assert got.shape == (world, 2 + world, 2)
            for r in range(world):
                assert int(got[r, :2 + r].min()) == r + 1 == int(got[r, :2 + r].max()) and int(got[r, 2 + r:].abs().max()) == 0
        else:
            assert got is None
        assert P.desc.shape == (n * kp, d_pad) and P.desc.dtype == torch.int8 and P.desc.is_contiguous()
        assert P.nrm.shape == (n * kp,) and P.nrm.dtype == torch.int32
        np.savez(out_path + f".{rank}.npz", desc=Pl.desc.numpy(), cbase=Pl.cbase.numpy(), nrm=Pl.nrm.numpy(), cin=Pl.cin.numpy(),
                 g_desc=P.desc.numpy(), g_cbase=P.cbase.numpy(), g_nrm=P.nrm.numpy(), g_cin=P.cin.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_all_gather_prepared_concatenates_the_ranks_fields(tmp_path, world):
    """pairs.all_gather_prepared (bench.py's collective on the way in since round 3): the four fields of every rank's prepared
    shard come back concatenated in rank order, on every rank, from ONE uint8 all_gather."""
    out = str(tmp_path / "prep")
    mp.spawn(_prepared_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    parts = [np.load(out + f".{r}.npz") for r in range(world)]
    for f in ("desc", "cbase", "nrm", "cin"):
        want = np.concatenate([p[f] for p in parts])
        for p in parts:
            assert np.array_equal(p["g_" + f], want), f

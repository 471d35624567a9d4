"""The collectives of the sharded get_collections path issued through RCCL (backend "nccl") on device tensors
of the shapes and dtypes bench.py uses, in a process group of one rank: checks that RCCL accepts the calls
(uint8 all_gather_into_tensor with a concatenated output -- raw descriptors and the packed prepared rows --, int32 gather to rank 0) and returns the data
unchanged.  The multi-rank arithmetic is covered on CPU by tests/test_dist_pairs_gloo.py."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_collectives_world_of_one(cuda):
    import torch
    import torch.distributed as dist
    from image_stitcher_amd import pairs as prs
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rng = np.random.default_rng(0)
        local = torch.from_numpy(rng.integers(0, 256, (4, 512, 128), dtype=np.uint8)).to(dev)
        full = prs.all_gather_descriptors(local, always=True)
        assert full.shape == local.shape and torch.equal(full, local)
        # bench.py's collective on the way in since round 3: the rank's PREPARED rows as one uint8 buffer
        from image_stitcher_amd import ops
        st, ct = np.arange(4, dtype=np.int64) * 512, np.full((4,), 512, np.int64)
        Pl = ops.sort_rows_by_norm(ops.prepare_descriptor_tensor(local.reshape(-1, 128), st, ct))
        P = prs.all_gather_prepared(Pl, st, ct, always=True)
        assert P.desc.shape == Pl.desc.shape and P.desc.dtype == torch.int8 and P.desc.data_ptr() % 16 == 0
        for a, b in ((P.desc, Pl.desc), (P.cbase, Pl.cbase), (P.nrm, Pl.nrm), (P.cin, Pl.cin)):
            assert a.dtype == b.dtype and torch.equal(a, b)
        # bench.py at N > 1: the same collective started asynchronously, the rank's own-shard pairs counted while it
        # travels on RCCL's stream, then the current stream (not the host) waits for it
        from image_stitcher_amd import pairs as prs2
        own = prs2.pair_items(prs2.unordered_pairs(4), st, ct)
        want = ops.knn2_l2_pairs(Pl, own, 0.65)
        landed = prs.all_gather_prepared(Pl, st, ct, always=True, async_op=True)
        got_own = ops.knn2_l2_pairs(Pl, own, 0.65)
        Pa = landed()
        got_all = ops.knn2_l2_pairs(Pa, own, 0.65)
        torch.cuda.synchronize()
        for a, b in ((Pa.desc, Pl.desc), (Pa.cbase, Pl.cbase), (Pa.nrm, Pl.nrm), (Pa.cin, Pl.cin)):
            assert a.dtype == b.dtype and torch.equal(a, b)
        assert torch.equal(got_own, want) and torch.equal(got_all, want)
        counts = torch.from_numpy(rng.integers(0, 4096, (37, 2)).astype(np.int32)).to(dev)
        allc = prs.gather_counts(counts, 37, 0, 1, always=True)
        assert allc.shape == (37, 2) and torch.equal(allc, counts)
        dist.barrier()
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's max-over-ranks timing
        assert float(t.item()) == 1.5
    finally:
        dist.destroy_process_group()

"""bench.py's multi-rank path rehearsed on one GPU: two ranks launched by torch.distributed.run share cuda:0 and
exchange through gloo (ISX_BENCH_BACKEND=gloo), on a small set.  Checks the sharded step end to end --
all_gather of the descriptor shards, one count launch per rank, gather of the counts, union-find -- and the
shape of the JSON line the driver parses.  (The measured configuration is one rank per GPU over RCCL.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_two_ranks_share_one_gpu(cuda, overlap):
    """overlap 1 (bench.py's default at N > 1): the gather of the prepared rows travels behind the ranks' own-shard pairs;
    0: gather first, then one launch over pairs p % N."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, ISX_BENCH_BACKEND="gloo", ISX_BENCH_OVERLAP=overlap)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--scenes", "2", "--imgs-per-scene", "8", "--kp", "1024"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["metric"] == "image pairs matched/sec (SIFT 4k kp)" and d["unit"] == "image pairs/s" and d["higher_is_better"] is True
    assert d["config"]["components_correct"] is True and d["config"]["pairs_per_step"] == 120
    assert ("own-shard pairs first" in d["config"]["parallelism"]) == (overlap == "1")
    assert d["value"] > 0 and abs(d["value"] - 120 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d and "stitch" not in d           # rank-0-at-N=1 extras only
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    b = d["step_breakdown"]                                            # per-phase attribution over the ranks
    for k in ("kernel_ms", "allgather_ms", "prep_plus_kernel_ms", "gather_ms"):
        assert 0 <= b[k]["min"] <= b[k]["max"]
    assert b["kernel_ms"]["max"] > 0 and b["host_ms_rank0"] >= 0

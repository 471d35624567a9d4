"""SURVEY 8f-2 on hardware: the strip-partitioned mosaic (image_stitcher_amd.strips) on TWO ranks that share one GPU
(gloo between them; on a node it is one rank per GPU).  Every rank holds only its rows; the gathered strips must equal,
bit for bit, what the oracle's single-canvas pipeline (tests/oracle_stitch.py) produces from the same frames and
transforms -- which is also what the single-GPU resident arena produces (tests/test_gpu_stitch.py) -- and the
blend_i.png files of a strip-partitioned stitch_collections must equal the single-GPU run's."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle_stitch import Cfg, oracle_stitch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_worker(tmp_path, *extra):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "strip_worker.py"), str(tmp_path), *extra]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, (r.stdout[-1500:] + r.stderr[-3000:])


def _check_against_the_oracle(scenario_list, results):
    """every frame of every scenario: the gathered strips equal the oracle's single canvas, the corners too, and the ranks'
    rows tile the mosaic.  Returns how many frames ended on a mosaic that both ranks hold rows of."""
    pts = np.zeros((5, 1, 2), np.float32)
    straddled = 0
    for name, frames, Ms, type_, blender, contrib in scenario_list:
        per_frame, stats = results[name]
        cfg = Cfg(type=type_, blender=blender, ref_image_contrib=contrib, consecutive_range=1)
        ref, corners = frames[0], None
        assert len(per_frame) == len(Ms)
        for k, (f, M) in enumerate(zip(frames[1:], Ms)):
            o = oracle_stitch(ref, f, None, cfg, tf=(np.asarray(M, np.float64), pts, pts), corners=corners)
            full, got_corners, owns = per_frame[k]
            assert full.shape == o[0].shape and np.array_equal(full, o[0]), f"{name} frame {k}: {(full != o[0]).sum()} bytes differ"
            assert got_corners == o[1], f"{name} frame {k}"
            # the two ranks' rows tile the mosaic without overlap
            spans = sorted(s for s in owns if s is not None and s[1] > s[0])
            assert spans[0][0] == 0 and spans[-1][1] == full.shape[0]
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            straddled += len(spans) == 2
            ref, corners = o[0], o[1]
    return straddled


def test_two_ranks_stitch_uhd_frames_across_the_strip_boundary(cuda, orc, tmp_path):
    """BASELINE configs[4]'s frame size on the strip partition: 3840 x 2160 frames, the mosaic growing to 5680 rows."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from strip_scenarios import uhd_scenarios
    _run_worker(tmp_path, "uhd")
    results, _ = np.load(tmp_path / "results_uhd.npy", allow_pickle=True)  # written by the worker of this test
    sc = uhd_scenarios()
    assert _check_against_the_oracle(sc, results) == 2
    per_frame, stats = results["uhd_down_b1"]
    assert per_frame[-1][0].shape[0] > 5500 and per_frame[-1][0].shape[1] >= 3840
    assert stats["rows_slow"] == 0 and stats["rescans"] == 0


def test_two_ranks_hold_strips_that_concatenate_to_the_single_gpu_mosaic(cuda, orc, tmp_path):
    from PIL import Image
    from image_stitcher_amd import ArbitraryStitcher
    from image_stitcher_amd.detect import NumpyOrbLite
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from strip_scenarios import collection_images, scenarios
    os.makedirs(tmp_path / "strips")
    _run_worker(tmp_path)
    results, cols = np.load(tmp_path / "results.npy", allow_pickle=True)  # written by the worker of this test
    straddled = _check_against_the_oracle(scenarios(), results)
    for name, frames, Ms, type_, blender, contrib in scenarios():
        per_frame, stats = results[name]
        if name == "rows_slow":
            assert stats["rows_slow"] == 1
        if name == "alpha_wrap_b0":
            assert stats["rescans"] >= 1 and stats["clears"] >= 1 and per_frame[0][0].shape[1] == 180
        if name.startswith("march_down"):
            assert stats["rescans"] == 0 and stats["rows_slow"] == 0 and stats["arena_allocs"] <= 6
            assert per_frame[-1][0].shape[0] > 200 * 2.5
            # the mosaic grew 2.5x downward and then up-left: the bounds were re-split on the way (rows moved between the
            # two ranks), so no strip ends up more than twice as tall as the other
            tall = [[b - a for a, b in sorted(s for s in owns if s is not None)] for _, _, owns in per_frame]
            assert stats.get("rebalances", 0) >= 1 and all(len(t) == 2 and max(t) <= 2 * min(t) + 2 for t in tall[2:]), tall
    assert straddled >= 15, "the scenarios must really spread the mosaic over both ranks"
    # stitch_collections: same files as the single-GPU run
    images = collection_images()
    one = ArbitraryStitcher(images=list(images), with_tqdm=False, log_level=3, algorithm=1, detector=NumpyOrbLite(nfeatures=500),
                            min_match_count=40, output_dir=str(tmp_path / "one"))
    os.makedirs(tmp_path / "one")
    cols1 = one.get_collections()
    assert [list(c) for c in cols1] == [list(c) for c in cols] and sorted(map(sorted, cols1)) == [[0, 2, 3], [1, 4]]
    assert one.stitch_collections(cols1) == []
    for i in range(len(cols1)):
        a = np.asarray(Image.open(tmp_path / "one" / f"blend_{i}.png"))
        b = np.asarray(Image.open(tmp_path / "strips" / f"blend_{i}.png"))
        assert a.shape == b.shape and np.array_equal(a, b) and a.shape[1] > 320 * 1.1

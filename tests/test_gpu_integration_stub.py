"""INTEGRATION.md section B shows the ctypes stub a maintainer of the reference would add to bind the matcher
seam to libisx_hip.so.  This test extracts that code block from the document and runs it, so the documented
binding cannot rot."""
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_documented_binding_runs(cuda, orc, monkeypatch):
    from image_stitcher_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n# image_stitcher/_isx\.py.*?\n(.*?)\n# image_stitcher/stitcher\.py", text, re.S).group(1)
    block = block.replace('C.CDLL("libisx_hip.so")', f'C.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    rng = np.random.default_rng(4)
    for q, t in ((rng.integers(0, 180, (300, 128), dtype=np.uint8), rng.integers(0, 180, (450, 128), dtype=np.uint8)),
                 (rng.integers(0, 150, (64, 128)).astype(np.float32), rng.integers(0, 150, (90, 128)).astype(np.float32))):
        idx, dist, n_good = ns["knn2_ratio"](q, t, 0.65)
        ridx, rdist = orc.knn2(q.astype(np.uint8), t.astype(np.uint8))
        assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(dist.cpu().numpy(), rdist)
        assert n_good == len(orc.ratio_filter(ridx, rdist, 0.65)[0])

"""Host-side RANSAC logic of libisx_hip.so (sampler, adaptive-stop selection): runs without a GPU.
Checked against the CPU oracle's independent restatement."""
import numpy as np


def _pts(n, seed):
    rng = np.random.default_rng(seed)
    src = rng.uniform(0, 1200, (n, 2)).astype(np.float32)
    th = 0.05
    dst = (src @ np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]]) + (30, -12)).astype(np.float32)
    dst[: n // 3] = rng.uniform(0, 1200, (n // 3, 2)).astype(np.float32)
    return src, dst


def test_sampler_matches_oracle(orc):
    from image_stitcher_amd import ops
    for n, seed in [(5, 0), (9, 1), (60, 2), (700, 3)]:
        src, dst = _pts(n, seed)
        for persp in (False, True):
            got = ops.ransac_sample(src, dst, persp, 300)
            exp = orc.ransac_sample(orc.MODEL_HOMOGRAPHY if persp else orc.MODEL_AFFINE_PARTIAL, src, dst, 300)
            assert np.array_equal(got, exp)
    # degenerate: identical points -> every subset rejected -> no subsets
    p = np.full((20, 2), 3.0, np.float32)
    assert len(ops.ransac_sample(p, p, True, 10)) == 0
    # n == model points: the identity subset
    src, dst = _pts(4, 9)
    assert ops.ransac_sample(src, dst, True, 10).tolist() == [[0, 1, 2, 3]]
    assert len(ops.ransac_sample(src[:3], dst[:3], True, 10)) == 0


def test_select_replays_sequential_ransac(orc):
    """Feeding the oracle's own per-iteration counts to isx_ransac_select must reproduce the iteration at
    which the sequential loop found its best model and the number of iterations it ran."""
    from image_stitcher_amd import ops
    for n, seed, persp in [(40, 0, True), (300, 1, True), (300, 2, False), (25, 3, False)]:
        src, dst = _pts(n, seed)
        model = orc.MODEL_HOMOGRAPHY if persp else orc.MODEL_AFFINE_PARTIAL
        cfg = ops.RANSAC_DEFAULTS[persp]
        ok, hyps, cnt, valid, best_it = orc.ransac_trace(model, src, dst, cfg["thr"], cfg["max_iters"], cfg["confidence"])
        assert ok
        # give select MORE hypotheses than the loop ran (as the GPU path does): pad with huge counts that
        # must never be looked at
        pad = 50
        counts = np.concatenate([cnt, np.full(pad, n, np.int32)])
        val = np.concatenate([valid, np.ones(pad, np.int32)])
        best, iters = ops.ransac_select(counts, val, n, persp, cfg["confidence"], cfg["max_iters"])
        assert best == best_it and iters == len(cnt)

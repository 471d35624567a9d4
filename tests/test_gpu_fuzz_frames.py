"""Randomised parity of the frame pipeline (Stitcher.stitch with a given transform on a resident arena) against the
oracle pipeline: sizes from 2 pixels up, affine and homography, both blenders, 1 and 3 channels, sparse / dark / solid
content, frames partly off the canvas.  tools/fuzz_frames.py holds the generator; campaigns of 20,000 small and 150 large
(up to 1560 x 1980) frame cases and 1,500 matcher cases were run on the round's final kernels without a mismatch."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_frames_against_the_oracle(cuda, orc):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import fuzz_frames
    assert fuzz_frames.main(150, 31000) == 0
    assert fuzz_frames.main(12, 32000, 5) == 0
    # several frames onto the same resident arena (growth, crop updates, clears); 1500 sequences ran clean
    assert fuzz_frames.sequences(40, 33000) == 0


def test_fuzz_frame_match_against_the_oracle(cuda, orc):
    """tools/fuzz_match.py: isx_frame_match (train segments merged per query) against the oracle's knnMatch + ratio loop
    on random sizes, widths and value ranges with many exact ties."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import fuzz_match
    assert fuzz_match.main(60) == 0


def test_fuzz_count_kernels_against_the_oracle(cuda, orc):
    """tools/fuzz_count.py: the count kernels of get_collections (default, both A/B kernels, the forced-exact path, the
    index kernel) against the oracle on random ragged image sets (400 cases ran clean; a short campaign here)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import fuzz_count
    assert fuzz_count.main(25) == 0


def test_fuzz_pair_kernel_against_the_oracle(cuda, orc):
    """tools/fuzz_pairs.py: isx_knn2_l2_count_pairs (one distance matrix per unordered pair: filter, exact recount, overflow to
    the directed kernel, early exit) against the oracle on ragged 128-d image sets of up to 4096 rows, sorted and unsorted,
    with planted near-duplicates from a handful to most of an image (250 cases ran clean on the round's final kernel; a short
    campaign on other seeds here)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import fuzz_pairs
    assert fuzz_pairs.main(20, 180000) == 0


def test_fuzz_robust_transform_against_the_oracle(cuda, orc):
    """tools/fuzz_ransac.py: get_transform (both models) against the oracle's sequential RANSAC on random point counts, outlier
    fractions and noise levels: the same inlier masks everywhere, M within 1e-3 px on well-determined sets (700 cases ran with
    the same verdicts on the round's final build and on the build before the refit / LM changes)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import fuzz_ransac
    assert fuzz_ransac.main(40, 330000) == 0

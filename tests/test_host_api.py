"""The frozen Python surface: constructor keywords and defaults of the three classes equal the
reference's (stitcher.py:37-50, :233, :378-383), state attributes exist with the reference's initial values,
error conventions.  No GPU is touched: constructing a stitcher must not need one."""
import inspect
import os
import sys

import numpy as np
import pytest


def _defaults(cls):
    sig = inspect.signature(cls.__init__)
    return {k: p.default for k, p in sig.parameters.items()
            if k != "self" and p.kind is inspect.Parameter.POSITIONAL_OR_KEYWORD}


def test_constructor_signatures_match_the_reference():
    from image_stitcher_amd import ArbitraryStitcher, ConsecutiveStitcher, Stitcher
    d = _defaults(Stitcher)
    assert list(d) == ["log_level", "output_dir", "min_match_count", "lowes_ratio_threshold", "algorithm", "matcher",
                       "type", "resize", "consecutive_range", "blend_processor", "blender", "ref_image_contrib"]
    bp = d.pop("blend_processor")
    assert bp("x") == "x"
    assert d == dict(log_level=2, output_dir="stitches/", min_match_count=5, lowes_ratio_threshold=0.65, algorithm=0,
                     matcher=0, type=0, resize=None, consecutive_range=None, blender=1, ref_image_contrib=0.5)
    assert _defaults(ArbitraryStitcher) == dict(images=None, filepaths=None, with_tqdm=True)
    assert _defaults(ConsecutiveStitcher) == dict(ref_image_contrib=0.2, consecutive_range=1, backup_interval=False,
                                                  consecutive_volatility_threshold=4)
    # the build's extensions are keyword-only, so positional calls written for the reference keep their meaning
    kwonly = [k for k, p in inspect.signature(Stitcher.__init__).parameters.items() if p.kind is inspect.Parameter.KEYWORD_ONLY]
    assert set(kwonly) == {"detector", "device", "resident", "norm", "async_save", "warp_arith", "parallel_detect", "depth16"}


def test_initial_state_and_error_conventions(tmp_path):
    from image_stitcher_amd import ArbitraryStitcher, ConsecutiveStitcher, TransformDetails
    st = ConsecutiveStitcher(log_level=3, output_dir=str(tmp_path), consecutive_range=3)
    assert st.stitch_count == 0 and st.image_count == 0 and st.current_image_count == 0 and st.corners is None
    assert len(st.refs) == 1 and st.refs[-1] is None and st.ref_image_contrib == 0.2 and st.consecutive_range == 3
    with pytest.raises(ValueError, match="No images specified"):      # stitcher.py:248
        ArbitraryStitcher(log_level=3)
    ar = ArbitraryStitcher(images=[np.zeros((4, 4, 3), np.uint8)], with_tqdm=False, log_level=3)
    assert ar.filepaths is None and len(ar.images) == 1
    ar2 = ArbitraryStitcher(filepaths=["a.png", "b.txt", "c.jpg"], with_tqdm=False, log_level=3)
    assert ar2.filepaths == ["a.png", "c.jpg"] and len(ar2.images) == 2           # stitcher.py:241-245
    ConsecutiveStitcher(log_level=0)       # the reference raises here (stitcher.py:69-70); the build switches logging off
    tf = TransformDetails("M", "s", "d")
    assert tf.get() == ("M", "s", "d") and (tf.M, tf.src_pts, tf.dst_pts) == ("M", "s", "d")   # stitcher.py:23-33
    for bad in (dict(norm="l1"), dict(warp_arith="nearest")):
        with pytest.raises(ValueError):
            ConsecutiveStitcher(log_level=3, **bad)


def test_image_stitcher_alias_package():
    """`image-stitcher_amd/compat` makes the reference's import lines resolve to this package's modules."""
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd")
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "from image_stitcher.stitcher import Stitcher, ArbitraryStitcher, ConsecutiveStitcher, TransformDetails\n"
            "from image_stitcher.utils.structs import DynamicConnectivity\n"
            "from image_stitcher.utils import proc, logger, utils\n"
            "import image_stitcher, image_stitcher_amd.stitcher as s\n"
            "assert ArbitraryStitcher is s.ArbitraryStitcher and image_stitcher.stitcher is s\n"
            "assert 'torch.cuda' not in sys.modules or not __import__('torch').cuda.is_initialized()\n"
            "print('ok')\n") % (root, os.path.join(root, "compat"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr


def test_load_image_round_trip(tmp_path):
    """load_image(write_image(a)) == a for BGR and gray PNGs (cv2.imread semantics: always 3 channels, BGR;
    unreadable file -> None), with or without cv2."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd"))
    from image_stitcher_amd.async_save import write_image
    from image_stitcher_amd.utils.proc import load_image
    rng = np.random.default_rng(3)
    bgr = rng.integers(0, 256, (17, 23, 3), dtype=np.uint8)
    p = str(tmp_path / "a.png")
    write_image(p, bgr)
    assert np.array_equal(load_image(p), bgr)
    gray = rng.integers(0, 256, (9, 11), dtype=np.uint8)
    write_image(p, gray)
    got = load_image(p)
    assert got.shape == (9, 11, 3) and all(np.array_equal(got[..., k], gray) for k in range(3))
    (tmp_path / "bad.png").write_bytes(b"not a png")
    assert load_image(str(tmp_path / "bad.png")) is None

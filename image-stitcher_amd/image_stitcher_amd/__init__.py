"""MI355X-native stitching core: drop-in for codegallivant/image-stitcher's hot path.

    from image_stitcher_amd import ArbitraryStitcher, ConsecutiveStitcher, TransformDetails

Importing the package does not touch the GPU; the stitcher classes load libisx_hip.so on first use and
fail loudly if it (or a HIP device) is missing -- there is no CPU fallback.
"""
__version__ = "0.1.0"


def __getattr__(name):
    if name in ("Stitcher", "ArbitraryStitcher", "ConsecutiveStitcher", "TransformDetails"):
        from . import stitcher
        return getattr(stitcher, name)
    raise AttributeError(name)

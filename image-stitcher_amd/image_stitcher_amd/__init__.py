"""MI355X-native stitching core: drop-in for codegallivant/image-stitcher's hot path.

    from image_stitcher_amd import ArbitraryStitcher, ConsecutiveStitcher, TransformDetails
"""
__version__ = "0.1.0"

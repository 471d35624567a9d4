"""Opt-in alias: put `image-stitcher_amd/` and `image-stitcher_amd/compat/` on sys.path (ahead of the
reference checkout) and code written as `from image_stitcher.stitcher import ArbitraryStitcher` runs on
the MI355X path unchanged.  Every submodule here IS the image_stitcher_amd module of the same name
(reference layout: image_stitcher/{stitcher.py, utils/{logger,proc,structs,utils}.py})."""
import importlib
import sys

import image_stitcher_amd as _amd

__version__ = _amd.__version__

for _name in ("stitcher", "utils", "utils.logger", "utils.proc", "utils.structs", "utils.utils"):
    sys.modules[__name__ + "." + _name] = importlib.import_module("image_stitcher_amd." + _name)
stitcher = sys.modules[__name__ + ".stitcher"]
utils = sys.modules[__name__ + ".utils"]

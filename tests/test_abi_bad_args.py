"""Every entry point of include/isx.h refuses unusable arguments with a negative status and a message, before it
touches the device: first with every size negative and every pointer NULL, then with plausible sizes and NULL pointers.
Nothing is computed, so the same cases run with and without a GPU (the -m gpu twin runs them on the box, where a
missing check would show up as a fault instead of a HIP error)."""
import ctypes
import os
import re

import pytest

from image_stitcher_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# entry points without arguments to get wrong, or whose answer for these arguments is a value, not a status
NOT_STATUS = {"isx_version", "isx_last_error", "isx_device_count", "isx_desc_pad", "isx_orb_workspace_bytes",
              "isx_ctx_destroy"}  # (isx_ctx_destroy(NULL) is a no-op by contract, like free)
PLAUSIBLE = {"c": 3, "d": 32, "d_pad": 32, "model": 0, "arith": 0, "blender": 1, "dtype": 0, "src_dtype": 0, "norm": 0,
             "feather": 10, "n_seg": 2, "clean": 1, "fast_threshold": 20}


def _prototypes():
    text = open(os.path.join(ROOT, "include", "isx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char \*)\s*(isx_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        out[m.group(1)] = [re.split(r"[\s\*]+", p.strip())[-1] for p in m.group(2).replace("\n", " ").split(",")]
    return out


def _call(name, plausible):
    _res, argtypes = _lib.SIGNATURES[name]
    names = _prototypes()[name]
    assert len(names) == len(argtypes), (name, names)
    vals = []
    for pn, a in zip(names, argtypes):
        if a is ctypes.c_void_p:
            vals.append(None)
        elif a is ctypes.c_double:
            vals.append(0.5)
        elif not plausible:
            vals.append(-1)
        elif a is ctypes.c_int64:
            vals.append(4096)
        else:
            vals.append(PLAUSIBLE.get(pn, 64))
    L = _lib.lib()
    rc = getattr(L, name)(*vals)
    return rc, L.isx_last_error().decode()


CASES = [(n, p) for n in _lib.SIGNATURES if n not in NOT_STATUS for p in (False, True)]


def _check(name, plausible):
    rc, msg = _call(name, plausible)
    assert rc < 0, (name, rc)
    assert msg, name  # isx_last_error() names the argument


@pytest.mark.parametrize("name,plausible", CASES)
def test_bad_arguments_are_refused(name, plausible):
    _check(name, plausible)


@pytest.mark.gpu
@pytest.mark.parametrize("name,plausible", CASES)
def test_bad_arguments_are_refused_on_the_gpu_box(name, plausible):
    _check(name, plausible)


def test_ctx_destroy_null_is_a_no_op():
    assert _lib.lib().isx_ctx_destroy(None) == 0

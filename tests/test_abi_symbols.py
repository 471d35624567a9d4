"""The C-ABI library loads without a GPU and exports every entry point include/isx.h declares; the ctypes
binding table of the host layer names exactly those symbols.  No compute call is made."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "isx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"^\s*(?:int|int64_t|const char \*)\s*(isx_\w+)\s*\(", text, flags=re.M)))


def test_header_symbols_are_exported():
    from image_stitcher_amd import _lib
    names = _declared()
    assert len(names) >= 20 and "isx_knn2_l2_count_items" in names and "isx_warp" in names
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/isx.h but not exported: {missing}"


def test_binding_table_matches_header():
    from image_stitcher_amd import _lib
    bound = set(_lib.SIGNATURES) if hasattr(_lib, "SIGNATURES") else None
    if bound is None:
        table = [v for v in vars(_lib).values() if isinstance(v, dict) and any(str(k).startswith("isx_") for k in v)]
        assert table, "no ctypes signature table found in _lib"
        bound = set(table[0])
    declared = set(_declared())
    assert bound <= declared, f"bound but not declared: {sorted(bound - declared)}"
    assert declared - bound <= {"isx_version", "isx_last_error", "isx_device_count"}, sorted(declared - bound)


def test_version_and_error_string_without_gpu():
    from image_stitcher_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.isx_version.restype = ctypes.c_int
    lib.isx_last_error.restype = ctypes.c_char_p
    assert lib.isx_version() > 0
    assert lib.isx_desc_pad(61) == 64 and lib.isx_desc_pad(128) == 128
    assert lib.isx_desc_pad(0) < 0 and b"descriptor width" in lib.isx_last_error()

"""isx_bbox_nonzero_static (include/isx.h): the one entry point that allocates nothing -- its per-block
partial table is a static device array inside the library.  Called through the C ABI and checked against
isx_bbox_nonzero (bit-exact: integer work) on two calls in a row, the second with a smaller input, so stale
partials of the first call would show."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _call(fn, img, out, stream):
    from image_stitcher_amd import _lib
    h, w, c = img.shape
    _lib.check(fn(ctypes.c_void_p(img.data_ptr()), h, w, c, img.stride(0), ctypes.c_void_p(out.data_ptr()),
                  ctypes.c_void_p(stream)))


def test_static_scratch_bbox_matches_the_allocating_entry_point(cuda, orc):
    import torch
    from image_stitcher_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(11)
    big = np.zeros((3100, 5200, 3), np.uint8)          # 21 x 194 tiles = more tiles than the 2048-block grid
    big[17:3050, 33:5100] = rng.integers(0, 255, (3033, 5067, 3), dtype=np.uint8)
    big[17, 33] = big[3049, 5099] = (0, 1, 0)           # gray != 0 corners
    small = np.zeros((70, 300, 3), np.uint8)            # 5 x 2 tiles: uses 10 entries of the table
    small[20:31, 100:140] = 200
    small[5, 7] = (4, 0, 0)                             # gray == 0: must not count
    stream = torch.cuda.current_stream().cuda_stream
    for img in (big, small):                            # second call: smaller input, table keeps the first call's partials
        d = torch.from_numpy(img).to(cuda)
        acc = torch.empty((_lib.BBOX_INTS,), dtype=torch.int32, device=cuda)
        res = torch.full((5,), -7, dtype=torch.int32, device=cuda)
        _call(L.isx_bbox_nonzero, d, acc, stream)
        _call(L.isx_bbox_nonzero_static, d, res, stream)
        torch.cuda.synchronize()
        exp, got = acc[:5].cpu().numpy(), res.cpu().numpy()
        assert np.array_equal(got, exp), (got, exp)
        (x0, y0), (x1, y1) = orc.bbox_nonzero(img)
        assert tuple(got) == (x0, y0, x1, y1, 1)
    # an all-black image after that: found == 0, and no stale box leaks in
    z = torch.zeros((64, 64, 3), dtype=torch.uint8, device=cuda)
    res = torch.full((5,), -7, dtype=torch.int32, device=cuda)
    _call(L.isx_bbox_nonzero_static, z, res, stream)
    torch.cuda.synchronize()
    assert int(res[4]) == 0


def test_static_scratch_bbox_rejects_oversized_images(cuda):
    import torch
    from image_stitcher_amd import _lib
    L = _lib.lib()
    d = torch.zeros((4, 4, 3), dtype=torch.uint8, device=cuda)
    res = torch.zeros((5,), dtype=torch.int32, device=cuda)
    rc = L.isx_bbox_nonzero_static(ctypes.c_void_p(d.data_ptr()), 4, 65537, 3, 65537 * 3, ctypes.c_void_p(res.data_ptr()), None)
    assert rc == -1 and b"up to 65536" in L.isx_last_error()

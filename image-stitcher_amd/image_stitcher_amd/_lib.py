"""ctypes binding of libisx_hip.so (include/isx.h).  No CPU fallback: if the library is missing the
import of any compute op fails loudly."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ISX_LIB_PATH") or os.path.join(HERE, "lib", "libisx_hip.so")  # override: diagnostic builds only

U8, F32 = 0, 1
NORM_L2, NORM_HAMMING = 0, 1
WARP_LINEAR_F32, WARP_CLASSIC = 0, 1  # enum isx_warp_arith
MODEL_AFFINE_PARTIAL, MODEL_HOMOGRAPHY = 0, 1
BLEND_ALPHA, BLEND_GRADIENT = 0, 1
L2_EXACT_LIMIT = 4197200
FRAME_RESULT_INTS = 32
BBOX_PARTIAL_BLOCKS, BBOX_HEAD = 2048, 8
BBOX_INTS = BBOX_HEAD + 4 * BBOX_PARTIAL_BLOCKS  # include/isx.h ISX_BBOX_INTS

c_void_p, c_int, c_int64, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double

# name -> (restype, argtypes); every symbol include/isx.h declares
SIGNATURES = {
    "isx_version": (c_int, []),
    "isx_last_error": (ctypes.c_char_p, []),
    "isx_device_count": (c_int, []),
    "isx_union_find": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "isx_desc_pad": (c_int, [c_int]),
    "isx_desc_prep": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "isx_knn2_l2_items": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_double,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "isx_knn2_l2_count_items": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_double, c_void_p,
                                        c_int, c_void_p]),
    "isx_knn2_l2_count_pairs": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_double, c_void_p,
                                        c_int, c_void_p]),
    "isx_sort_rows_by_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "isx_knn2_l2_wide_items": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "isx_knn2_hamming_items": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "isx_hamming_fp4_prep": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "isx_knn2_hamming_count_fp4": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_double, c_void_p, c_void_p]),
    "isx_f32_norms": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "isx_knn2_f32_items": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_double, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "isx_ratio_compact": (c_int, [c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "isx_ratio_compact_segments": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p,
                                           c_void_p]),
    "isx_bbox_nonzero": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "isx_bbox_reset": (c_int, [c_void_p, c_void_p]),
    "isx_bbox_nonzero_static": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "isx_warp": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_int64,
                         c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "isx_warp_arith": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_int64,
                               c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "isx_warp_alpha_blend": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_int64,
                                     c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_double,
                                     c_void_p, c_void_p]),
    "isx_warp16": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_int,
                           c_int, c_int, c_void_p]),
    "isx_alpha_blend16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_double, c_void_p]),
    "isx_alpha_blend": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int,
                                c_double, c_void_p]),
    "isx_gradient_blend": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p,
                                   c_void_p]),
    "isx_frame_download": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_int64, c_int, c_int64, c_void_p,
                                   c_void_p]),
    "isx_orb_workspace_bytes": (c_int64, [c_int, c_int]),
    "isx_orb_candidates": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    "isx_orb_describe": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "isx_orb_select_describe": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p]),
    "isx_ctx_create": (c_int, [c_void_p]),
    "isx_ctx_destroy": (c_int, [c_void_p]),
    "isx_frame_blend": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_void_p, c_int, c_int,
                                c_int, c_int, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_double,
                                c_int, c_void_p, c_void_p]),
    "isx_frame_match": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_double, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p]),
    "isx_frame_transform": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "isx_frame_result": (c_int, [c_void_p, c_void_p, c_void_p]),
    "isx_frame_warped": (c_int, [c_void_p, c_void_p]),
    "isx_copy2d": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int64, c_void_p]),
    "isx_clear_rects": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_int, c_void_p]),
    "isx_ransac_sample": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "isx_ransac_hypotheses": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "isx_ransac_score": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p,
                                 c_void_p]),
    "isx_ransac_select": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_double, c_int, c_void_p, c_void_p]),
    "isx_ransac_refine": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_double, c_int, c_int, c_void_p,
                                  c_void_p, c_void_p]),
}


class IsxError(RuntimeError):
    pass


class NotIntegerValuedError(IsxError):
    """float32 descriptors that are not integer valued in [0, 255] (found on the device by isx_desc_prep): the
    exact int8 path does not apply; the caller uses the fp32 matcher."""


class NormLimitError(IsxError):
    """2*max||v||^2 reaches ISX_L2_EXACT_LIMIT: squared distances may leave the range where sqrtf is injective,
    so the integer ranking could differ from cv2's float ranking; the caller falls back to the fp32 matcher."""


ABI_VERSION = 3  # include/isx.h ISX_ABI_VERSION
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IsxError(
                f"{LIB_PATH} is missing: build it with `python image-stitcher_amd/build.py` "
                "(there is no CPU fallback for the stitching core)")
        # torch bundles its own libamdhip64; it must be the HIP runtime this process binds first,
        # otherwise two runtimes fight over the device ("no ROCm-capable device is detected").
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.isx_version() != ABI_VERSION:
            raise IsxError(f"{LIB_PATH} was built for ABI version {L.isx_version()}, this package binds version {ABI_VERSION} "
                           "(include/isx.h ISX_ABI_VERSION): rebuild with `python image-stitcher_amd/build.py --force`")
        _lib = L
    return _lib


def check(status):
    if status < 0:
        raise IsxError(f"isx error {status}: {lib().isx_last_error().decode()}")
    return status

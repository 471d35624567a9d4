/*
 * include/isx.h -- C ABI of libisx_hip.so, the MI355X (gfx950) stitching core.
 *
 * The reference (codegallivant/image-stitcher v0.9.1) is pure Python and has NO plugin / FFI
 * interface: its hot path is seven overridable `Stitcher` methods that call cv2 and numpy.  Each entry
 * point below therefore cites the reference *method* (file:line, relative to
 * /root/reference/image_stitcher/) whose arithmetic it replaces; INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer argument named *_dev is a raw DEVICE address (e.g. torch.Tensor.data_ptr());
 *     every pointer named *_host is a host address;
 *   - the caller owns every buffer; the library allocates nothing it returns;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is queued on it and
 *     the call returns without synchronising unless stated otherwise;
 *   - every function returns 0 on success or a negative isx_status; isx_last_error() returns a
 *     thread-local human readable message for the last failure on the calling thread;
 *   - calls may be issued from any thread and on any stream.  Device state is held in exactly two places: an isx_ctx
 *     (the per-stitcher workspace of the isx_frame_* calls: staging image, tables, pinned mirrors; created and
 *     destroyed by the caller, not re-entrant within one ctx) and the 32 KiB file-scope table of
 *     isx_bbox_nonzero_static (one per GPU, shared by every call on that GPU).  Every other entry point keeps no
 *     state between calls;
 *   - there is no CPU fallback: without a HIP device the compute entry points fail with ISX_E_HIP.
 */
#ifndef ISX_H
#define ISX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum isx_status {
    ISX_OK = 0,
    ISX_E_ARG = -1,         /* invalid argument */
    ISX_E_HIP = -2,         /* a HIP runtime call failed (message has the hipError string) */
    ISX_E_UNSUPPORTED = -3, /* valid request this build has no kernel for (message says which) */
    ISX_E_RANGE = -4        /* data outside the exact-arithmetic range of the requested path */
};

enum isx_dtype { ISX_U8 = 0, ISX_F32 = 1 };
enum isx_norm { ISX_NORM_L2 = 0, ISX_NORM_HAMMING = 1 };
enum isx_model { ISX_MODEL_AFFINE_PARTIAL = 0, ISX_MODEL_HOMOGRAPHY = 1 }; /* stitcher.py:137-140 `type` */
/* which INTER_LINEAR arithmetic cv2.warpAffine / cv2.warpPerspective is taken to run:
 * the float32 linear kernels of OpenCV 4.11 (the reference's pinned opencv-python==4.11.0.86,
 * requirements.txt:6) or the classic 1/32-px fixed-point remap of OpenCV <= 4.10 */
enum isx_warp_arith { ISX_WARP_LINEAR_F32 = 0, ISX_WARP_CLASSIC = 1 };
enum isx_blender { ISX_BLEND_ALPHA = 0, ISX_BLEND_GRADIENT = 1 };          /* stitcher.py:153-156 `blender` */

/* ABI version of this header.  isx_version() returns the value the library was built with; a caller compares the two
 * before anything else (image_stitcher_amd._lib does).  History: 1 = round 1; 2 = round 2 (bbox_out_dev of isx_bbox_nonzero /
 * isx_warp / isx_warp_arith / isx_warp_alpha_blend grew from int32[5] to int32[ISX_BBOX_INTS] under unchanged names: a
 * round-1 caller passing 5 ints would be overwritten -- that is what this number is for); 3 = round 3 (additions only:
 * isx_knn2_l2_count_pairs, isx_sort_rows_by_norm, isx_warp16, isx_alpha_blend16). */
#define ISX_ABI_VERSION 3
int isx_version(void);
const char *isx_last_error(void);
/* number of HIP devices visible (0 if none / no driver).  Never fails. */
int isx_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Descriptor matching.   Replaces Stitcher.get_matches + Stitcher.filter_matches
 * (stitcher.py:106-116 `matcher_obj.knnMatch(des1, des2, k=2)` with matcher_obj = cv2.BFMatcher()
 * built at stitcher.py:85, and stitcher.py:118-127 Lowe ratio test) and the all-pairs loop body of
 * ArbitraryStitcher.get_collections (stitcher.py:285-296).
 * ------------------------------------------------------------------------------------------------ */

/* Padded descriptor width the integer L2 path uses for a descriptor of `d` bytes (32, 64 or 128;
 * negative status if d is unsupported). */
int isx_desc_pad(int d);

/* Prepare `rows` descriptors for the exact integer-L2 matcher.
 *   src_dev   : rows x d, row-major, uint8 (ISX_U8) or float32 (ISX_F32, must be integer valued in
 *               [0,255] -- what cv2.SIFT returns; anything else raises flag bit 0)
 *   desc_dev  : rows x isx_desc_pad(d) int8   (value - 128, zero padded)
 *   cbase_dev : rows uint32                   ((BIAS - ||v-128||^2) << 8, the per-train-row key base)
 *   nrm_dev   : rows int32                    (||v-128||^2)
 *   cin_dev   : rows int32                    (2^22 - ceil(nrm/2): MFMA start value of the count-only kernel)
 *   stats_dev : 2 int32, caller-zeroed: [0] |= 1 if a float value was not an integer in [0,255];
 *               [1] = max over rows of ||v||^2 (atomicMax) -- the caller checks 2*max < ISX_L2_EXACT_LIMIT
 */
#define ISX_L2_EXACT_LIMIT 4197200 /* sqrtf is injective on integers below this (verified exhaustively) */
int isx_desc_prep(const void *src_dev, int src_dtype, int64_t rows, int d, int8_t *desc_dev,
                  uint32_t *cbase_dev, int32_t *nrm_dev, int32_t *cin_dev, int32_t *stats_dev, void *stream);

/* One matching work item: query rows [q_start, q_start+nq) against train rows [t_start, t_start+nt)
 * of the prepared arrays. */
typedef struct isx_knn_item {
    int32_t q_start, nq, t_start, nt;
} isx_knn_item;

/* Batched exact kNN-2 (L2 over integer descriptors, distances = sqrtf(int), ties -> lowest train index
 * in both slots, exactly cv2.BFMatcher(NORM_L2).knnMatch(k=2)) fused with the Lowe ratio test.
 *   items_dev     : n_items work items (device)
 *   max_nq        : max over items of nq (host value; sizes the grid)
 *   ratio         : lowes_ratio_threshold, compared in float64 as the reference's Python does
 *   idx_out_dev   : optional (NULL = counts only; prefer isx_knn2_l2_count_items then) int32 [(out_start[e]+i)*2 + k], -1 = absent
 *   dist_out_dev  : optional float32, same indexing (FLT_MAX where absent)
 *   out_start_dev : int64 per item, first output row of the item (required iff idx/dist given)
 *   count_out_dev : optional int32 per item, MUST be zeroed by the caller; receives len(good_matches)
 * An item list of any length is accepted: lists whose grid would pass 2^31 threads are run as several
 * launches on `stream` (true of every *_items entry point below).
 */
int isx_knn2_l2_items(const int8_t *desc_dev, const uint32_t *cbase_dev, const int32_t *nrm_dev,
                      int d_pad, const isx_knn_item *items_dev, int n_items, int max_nq, double ratio,
                      int32_t *idx_out_dev, float *dist_out_dev, const int64_t *out_start_dev,
                      int32_t *count_out_dev, void *stream);

/* Count-only form of the above: the loop body of ArbitraryStitcher.get_collections
 * (stitcher.py:292-296 uses only len(good_matches) of every ordered pair).  Same counts, bit for bit,
 * as isx_knn2_l2_items; it skips the train indices, which more than halves the per-distance vector work.
 *   count_out_dev : int32 per item, MUST be zeroed by the caller
 *   flags         : bit 0 = decide EVERY query by the exact per-query scan (test hook, slow);
 *                   bit 1 = run 128- and 64-byte rows on the 32x32x32 MFMA kernel instead of the (faster) 16x16x64 ones;
 *                   bit 2 = run 128-byte rows on the 16x16x64 kernel with 64 queries per wave instead of the (faster)
 *                   one with 128 (A/B hooks: all give the same counts); other bits 0 */
int isx_knn2_l2_count_items(const int8_t *desc_dev, const int32_t *nrm_dev, const int32_t *cin_dev, int d_pad,
                            const isx_knn_item *items_dev, int n_items, int max_nq, double ratio,
                            int32_t *count_out_dev, int flags, void *stream);

/* Both directions of UNORDERED image pairs from one distance matrix (round 3): reference stitcher.py:285-296 visits
 * (i, j) and (j, i); this entry point forms the products of a pair of 128- or 64-byte-row images once (half the matrix instructions of
 * two isx_knn2_l2_count_items items) and settles almost every query with a max-only filter that proves "not a good
 * match"; the few candidates are recounted exactly in the kernel.  One work item = one pair: {q_start, nq} = image j,
 * {t_start, nt} = image i.
 *   count_out_dev[2 e]     = len(good_matches) of (query = image j, train = image i)
 *   count_out_dev[2 e + 1] = len(good_matches) of (query = image i, train = image j)
 *   -1 = that direction has more than 128 candidates (the images overlap), or the other direction of the pair had and
 *        the kernel left the pair early: the caller runs it as an item of isx_knn2_l2_count_items.  Every other value
 *        is the exact count (same as isx_knn2_l2_count_items).
 * count_out_dev need not be zeroed.  d_pad must be 128 (SIFT) or 64 (AKAZE's 61 bytes, padded).  max_rows = max over items of nq and nt: up to 4096 the kernel keeps
 * four group maxima per row; beyond, two (a few more candidates are recounted), the smaller image of a pair is made the train
 * side, and a pair whose smaller image has more than 8192 rows or whose larger one more than 16384 comes back as (-1, -1).  Rows sorted
 * by nrm inside each image keep the candidate lists short (counts do not depend on the row order); any order is exact.
 * flags bit 0 = test hook: every query is a candidate. */
int isx_knn2_l2_count_pairs(const int8_t *desc_dev, const int32_t *nrm_dev, const int32_t *cin_dev, int d_pad,
                            const isx_knn_item *items_dev, int n_items, int max_rows, double ratio,
                            int32_t *count_out_dev, int flags, void *stream);

/* The row order isx_knn2_l2_count_pairs likes: the prepared rows of every image (isx_desc_prep's four outputs) sorted by
 * nrm, ties by row, and rotated by 64 where an image has more than 256 rows (its 64 smallest norms follow its 64 largest:
 * one class of widely spread norms per image instead of two).  No counterpart in the reference: stitcher.py:292-296 uses only
 * len(good_matches) of a pair, which does not depend on the row order of either image -- use the sorted set for the count
 * entry points only (isx_knn2_l2_items reports row indices).
 *   images back to back: image i = rows [starts_dev[i], starts_dev[i + 1]) (n_images + 1 int32 on the device), or
 *   starts_dev = NULL and every image has uniform_count rows; max_count = the largest image (<= ISX_SORT_MAX_ROWS, else
 *   ISX_E_UNSUPPORTED).  Out of place: the four outputs are as large as the inputs and none may alias its input. */
#define ISX_SORT_MAX_ROWS 16384
int isx_sort_rows_by_norm(const int8_t *desc_in_dev, const uint32_t *cbase_in_dev, const int32_t *nrm_in_dev,
                          const int32_t *cin_in_dev, int d_pad, const int32_t *starts_dev, int n_images, int uniform_count,
                          int max_count, int8_t *desc_out_dev, uint32_t *cbase_out_dev, int32_t *nrm_out_dev,
                          int32_t *cin_out_dev, void *stream);

/* The same matcher without a norm limit: exact L2 kNN-2 + ratio for byte descriptors whose squared distances can
 * leave the range where sqrtf is injective (2*max||v||^2 >= ISX_L2_EXACT_LIMIT: saturated AKAZE / ORB rows).  Ranks on
 * the float32 distances cv2.BFMatcher compares (stitcher.py:85, :115), so squared distances that round to the same
 * float tie to the lowest train index in both slots, as in cv2.  Inputs as isx_knn2_l2_items (the prepared rows and
 * norms of isx_desc_prep; cbase is not used); VALU dot products, about 10x the MFMA kernel's time.  count_out_dev
 * optional (caller-zeroed); max_nt < 2^18. */
int isx_knn2_l2_wide_items(const int8_t *desc_dev, const int32_t *nrm_dev, int d_pad, const isx_knn_item *items_dev,
                           int n_items, int max_nq, int max_nt, double ratio, int32_t *idx_out_dev, float *dist_out_dev,
                           const int64_t *out_start_dev, int32_t *count_out_dev, void *stream);

/* Opt-in Hamming matcher (XOR + popcount): the parity target is cv2.BFMatcher(cv2.NORM_HAMMING).knnMatch(k=2)
 * -- NOT what the reference runs by default (stitcher.py:85 builds cv2.BFMatcher(), i.e. L2, for ORB too).
 *   desc_dev : rows x d_pad bytes (d_pad 32 or 64; shorter descriptors zero padded on both sides)
 *   distance = popcount as float32; ties -> lowest train index; outputs as isx_knn2_l2_items
 *   max_nt   : max over items of nt (< 2^18, cv2's own limit) */
int isx_knn2_hamming_items(const uint8_t *desc_dev, int d_pad, const isx_knn_item *items_dev, int n_items,
                           int max_nq, int max_nt, double ratio, int32_t *idx_out_dev, float *dist_out_dev,
                           const int64_t *out_start_dev, int32_t *count_out_dev, void *stream);

/* Hamming good-match counts on the matrix pipe (opt-in norm = "hamming", binary descriptors of up to 256 bits):
 * isx_hamming_fp4_prep expands every bit into one FP4 nibble (rows x 128 bytes), nrm = popcount, cin = float
 * bits of 512 - popcount/2; isx_knn2_hamming_count_fp4 is isx_knn2_l2_count_items on those rows with
 * v_mfma_scale_f32_32x32x64_f8f6f4: popcount(a ^ b) = |a| + |b| - 2 a.b, exact in float32.  Same counts as
 * isx_knn2_hamming_items (cv2.BFMatcher(NORM_HAMMING).knnMatch + the ratio loop of stitcher.py:119-127).
 * No counterpart call in the reference, whose matcher is L2 for every detector (stitcher.py:85). */
int isx_hamming_fp4_prep(const uint8_t *src_dev, int64_t rows, int d, uint8_t *desc_dev, int32_t *nrm_dev,
                         int32_t *cin_dev, void *stream);
int isx_knn2_hamming_count_fp4(const uint8_t *desc_dev, const int32_t *nrm_dev, const int32_t *cin_dev,
                               const isx_knn_item *items_dev, int n_items, int max_nq, double ratio,
                               int32_t *count_out_dev, void *stream);

/* float32 L2 matcher for descriptors that are not integer valued (stitcher.py:106-116 with float
 * descriptors; cv2.SIFT's are integer valued and take the exact path above).  fp32 MFMA, ranking on
 * na + nb - 2ab, winners' distances recomputed directly.  Gate: 1e-4 relative distance.
 *   desc_dev : rows x d float32 (d = 32, 64 or 128), nrm_dev: rows float32 from isx_f32_norms */
int isx_f32_norms(const float *desc_dev, int64_t rows, int d, float *nrm_out_dev, void *stream);
int isx_knn2_f32_items(const float *desc_dev, const float *nrm_dev, int d, const isx_knn_item *items_dev,
                       int n_items, int max_nq, double ratio, int32_t *idx_out_dev, float *dist_out_dev,
                       const int64_t *out_start_dev, int32_t *count_out_dev, void *stream);

/* Stitcher.filter_matches as a list (stitcher.py:118-127): stable compaction, in query order, of the rows
 * of a kNN-2 result that pass the ratio test (single-match rows are kept unconditionally, :125-126).
 * good_q/good_t: nq int32 each (capacity); count_out_dev: 1 int32 = len(good_matches). */
int isx_ratio_compact(const int32_t *idx_dev, const float *dist_dev, int nq, double ratio,
                      int32_t *good_q_out_dev, int32_t *good_t_out_dev, int32_t *count_out_dev, void *stream);
/* The same list from a kNN-2 result computed in n_seg pieces: result s (entry offset s * nq * 2) holds the two nearest
 * rows of train segment s = rows [s * seg_rows, (s + 1) * seg_rows) with indices relative to the segment; the pieces
 * are merged per query under (distance, train index), which is BFMatcher's order (stitcher.py:106-116), and the list
 * is built as above (n_seg <= 16; idx_dev / dist_dev 8-byte aligned).  Lets one image pair spread over the whole chip: isx_frame_match runs the index kernel on
 * (query block x train segment) work items instead of query blocks only. */
int isx_ratio_compact_segments(const int32_t *idx_dev, const float *dist_dev, int nq, int n_seg, int seg_rows, double ratio,
                               int32_t *good_q_out_dev, int32_t *good_t_out_dev, int32_t *count_out_dev, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Image space.  All images: uint8, interleaved channels (c = 1 or 3), PITCHED: `pitch` is the row
 * stride in bytes, so a crop / ROI of a device image is a pointer to its first pixel + the parent pitch.
 * ------------------------------------------------------------------------------------------------ */

/* A bbox ACCUMULATOR is ISX_BBOX_INTS int32 of caller-owned device memory: ints 0..4 are the result
 * {x0, y0, x1, y1 (exclusive), found}, ints ISX_BBOX_HEAD.. are the per-workgroup partial table the reduction
 * kernels use as scratch (caller-owned, so bbox-producing calls on different accumulators may overlap on one GPU). */
#define ISX_BBOX_PARTIAL_BLOCKS 2048
#define ISX_BBOX_HEAD 8
#define ISX_BBOX_INTS (ISX_BBOX_HEAD + 4 * ISX_BBOX_PARTIAL_BLOCKS)

/* Replaces get_corners_from_image (utils/proc.py:34-45: cv2.cvtColor(BGR2GRAY) -> cv2.findNonZero ->
 * cv2.boundingRect), also the bbox half of crop_image (utils/proc.py:307-312).
 * bbox_out_dev: a bbox accumulator (ISX_BBOX_INTS int32); found == 0 <=> the reference returns False. */
int isx_bbox_nonzero(const uint8_t *img_dev, int h, int w, int c, int64_t pitch, int32_t *bbox_out_dev,
                     void *stream);
/* Re-initialise a bbox accumulator (for isx_warp's accumulating bbox output). */
int isx_bbox_reset(int32_t *bbox_dev, void *stream);

/* isx_bbox_nonzero for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc, code that prepares a HIP graph).  NO COUNTERPART IN THE REFERENCE.  bbox_out_dev is just the 5-int result
 * {x0, y0, x1, y1, found}; the partial table is a static device array inside the library (32 KiB per GPU, held until
 * the process ends), so the entry point allocates nothing and needs no workspace.
 * Limits: h, w <= ISX_BBOX_STATIC_MAX_DIM (larger images are refused with ISX_E_ARG); the image's device must be
 * current and own `stream`.  ONE STREAM PER GPU: every call on a GPU uses that GPU's one table, so two calls that can
 * overlap on a GPU (two streams, two threads) race and return garbage boxes; serialise them or use isx_bbox_nonzero.
 * The first call on a device resolves the table's address (hipGetSymbolAddress, under a mutex) and every call queries
 * the current device and the stream's device: make one eager call per device BEFORE a stream capture begins -- a first
 * call issued during capture may be refused by the runtime.  Capturing later calls is untested here. */
#define ISX_BBOX_STATIC_MAX_DIM 65536
int isx_bbox_nonzero_static(const uint8_t *img_dev, int h, int w, int c, int64_t pitch, int32_t *bbox_out_dev,
                            void *stream);

/* Replaces Stitcher.warp_target (stitcher.py:143-149: cv2.warpAffine / cv2.warpPerspective with
 * INTER_LINEAR, BORDER_CONSTANT 0, dsize = padded canvas) for a source that is the UN-padded frame
 * sitting at (src_x, src_y) of the virtual padded canvas of get_padded_images (utils/proc.py:295-304).
 * Produces only the canvas rectangle [dx0,dx0+dw) x [dy0,dy0+dh) into dst_dev (its first pixel).
 *   M_host : forward map, 6 doubles (model ISX_MODEL_AFFINE_PARTIAL, 2x3) or 9 doubles (3x3)
 *   bbox_out_dev : optional bbox accumulator (ISX_BBOX_INTS int32, see isx_bbox_reset) that receives the bbox, in canvas
 *                  coordinates, of produced pixels whose gray != 0  (stitcher.py:220) */
int isx_warp(const uint8_t *src_dev, int sh, int sw, int c, int64_t spitch, int src_x, int src_y,
             const double *M_host, int model, uint8_t *dst_dev, int64_t dpitch, int dx0, int dy0, int dw,
             int dh, int32_t *bbox_out_dev, void *stream);
/* isx_warp with the arithmetic chosen explicitly (isx_warp = ISX_WARP_LINEAR_F32).  canvas_w/canvas_h =
 * the dsize of the reference's call (stitcher.py:146, :148: the padded canvas); the classic perspective
 * path needs it because cv::WarpPerspectiveInvoker evaluates the map per 64-column block of that canvas. */
int isx_warp_arith(const uint8_t *src_dev, int sh, int sw, int c, int64_t spitch, int src_x, int src_y,
                   const double *M_host, int model, uint8_t *dst_dev, int64_t dpitch, int dx0, int dy0, int dw,
                   int dh, int arith, int canvas_w, int canvas_h, int32_t *bbox_out_dev, void *stream);

/* Opt-in depth-preserving mode (BASELINE configs[3]; Stitcher(depth16=True)).  The reference's canvases are uint8 whatever
 * the frame's dtype (utils/proc.py:283-291), so uint16 depth wraps modulo 256 there -- the default here as well.  These two
 * entry points are Stitcher.warp_target (stitcher.py:143-149) and seamless_merge (utils/proc.py:48-82) as the same code
 * computes them when the canvases take the frame's dtype: single-channel uint16 images, pitches in BYTES (even).
 * isx_warp16: arguments as isx_warp (float32 linear arithmetic only), dst = the rectangle's own buffer (dw x dh).
 * isx_alpha_blend16: mask = v > 10, overlap products truncated in float64, uint16 sum wraps modulo 65536; out may alias ref. */
int isx_warp16(const uint16_t *src_dev, int sh, int sw, int64_t spitch_bytes, int src_x, int src_y, const double *M_host,
               int model, uint16_t *dst_dev, int64_t dpitch_bytes, int dx0, int dy0, int dw, int dh, void *stream);
int isx_alpha_blend16(const uint16_t *warped_dev, int64_t wpitch_bytes, const uint16_t *ref_dev, int64_t rpitch_bytes,
                      uint16_t *out_dev, int64_t opitch_bytes, int h, int w, double ref_contrib, void *stream);

/* Replaces seamless_merge (utils/proc.py:48-82; Stitcher.merge blender=0, stitcher.py:153-154).
 * warped/ref/out are same-size views; out may alias ref. */
int isx_alpha_blend(const uint8_t *warped_dev, int64_t wpitch, const uint8_t *ref_dev, int64_t rpitch,
                    uint8_t *out_dev, int64_t opitch, int h, int w, int c, double ref_contrib, void *stream);

/* Stitcher.warp_target followed by Stitcher.merge with blender = 0 (stitcher.py:216-217 -> utils/proc.py:48-82)
 * in ONE pass over the destination rectangle: the warped pixel is produced as isx_warp_arith produces it,
 * counted into bbox_out_dev (gray != 0, stitcher.py:220) and alpha-blended with the mosaic pixel under it
 * (ref_dev; out_dev may alias it).  ref_dev / out_dev point at the rectangle's first pixel. */
int isx_warp_alpha_blend(const uint8_t *src_dev, int sh, int sw, int c, int64_t spitch, int src_x, int src_y,
                         const double *M_host, int model, const uint8_t *ref_dev, int64_t rpitch,
                         uint8_t *out_dev, int64_t opitch, int dx0, int dy0, int dw, int dh, int arith,
                         int canvas_w, int canvas_h, double ref_contrib, int32_t *bbox_out_dev, void *stream);

/* Replaces the ROI body of seamless_gradient_merge (utils/proc.py:207-253; Stitcher.merge blender=1).
 * warped/ref are the two ROI views the reference cuts at proc.py:199-205 (bbox(warped) +- 50 px clipped
 * to the canvas); ref is updated in place.  ext_dev: caller scratch of 2*h + 2*w int32. */
int isx_gradient_blend(const uint8_t *warped_dev, int64_t wpitch, uint8_t *ref_dev, int64_t rpitch, int h, int w,
                       int c, int feather, int32_t *ext_dev, void *stream);

/* Pitched device-to-device copy (mosaic arena growth / get_padded_images paste, utils/proc.py:290-291). */
int isx_copy2d(const uint8_t *src_dev, int64_t spitch, uint8_t *dst_dev, int64_t dpitch, int h,
               int64_t row_bytes, void *stream);
/* Zero up to four rectangles {row0, row1, byte0, byte1} (half-open, rects_host = 4 int32 each, HOST memory) of a pitched
 * image in one launch: the strips between the merged canvas's crop (stitcher.py:225, utils/proc.py:307-312 drops them)
 * and what the blend wrote on a resident mosaic arena. */
int isx_clear_rects(uint8_t *dst_dev, int64_t dpitch, int h, int64_t row_bytes, const int32_t *rects_host, int n, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Per-frame pipeline on a resident mosaic.  Replaces, in one call, the tail of Stitcher.stitch
 * (stitcher.py:216-225): warp_target -> merge -> get_corners_from_image(warped) -> crop_image, i.e.
 * cv2.warpAffine | cv2.warpPerspective on the padded canvas of get_padded_images (utils/proc.py:295-304),
 * seamless_merge (utils/proc.py:48-82) or seamless_gradient_merge (utils/proc.py:192-254), and the two
 * cv2.findNonZero + cv2.boundingRect scans (utils/proc.py:34-45, :307-312).
 *
 * isx_ctx is the per-device workspace handle of SURVEY 8b: it owns the scratch of the frame kernels (the staged
 * warped image, extent tables, a small result block in device and pinned host memory) and grows it on demand,
 * so a steady stream allocates nothing per frame.  Create it with the device current; not re-entrant (one call
 * at a time per ctx); destroy it explicitly.
 * ------------------------------------------------------------------------------------------------ */
typedef struct isx_ctx isx_ctx;
int isx_ctx_create(isx_ctx **ctx_out);
int isx_ctx_destroy(isx_ctx *ctx);

#define ISX_FRAME_RESULT_INTS 32
#define ISX_FRAME_MAX_Q 32768
/* The mosaic lives in a caller-owned image ("arena") that is ZERO outside the mosaic, exactly as the reference's
 * padded canvas is; coordinates below are those of that virtual canvas (ref pasted at (img_w, img_h), the frame at
 * (src_x, src_y) = ((ref_w+img_w)//2, (ref_h+img_h)//2), utils/proc.py:286-291).
 *   Q = [q_x0, q_x0+q_w) x [q_y0, q_y0+q_h) : a canvas rectangle that contains every warped pixel that can be
 *       non-zero, plus >= 52 px on every side for the gradient blender's ROI (clipped to the canvas);
 *   mosaic_q_dev / mpitch : the arena pixel under canvas (q_x0, q_y0); all of Q must lie inside the arena;
 *   clip_y0 / clip_y1 : canvas rows [clip_y0, clip_y1) of the mosaic that THIS arena holds (0, canvas_h: all of it).  A
 *       mosaic partitioned into row strips over several GPUs is blended by one call per GPU with the same
 *       frame and M: the warp, bbox(warped), ROI and extents are computed for the whole frame on every GPU (they do not
 *       depend on the mosaic), only rows inside the clip are read or written, and [9..15] then describe this strip only;
 *   blender / ref_contrib / feather : stitcher.py:153-156, utils/proc.py:48 `image1_contribution`, :192 (10).
 * The mosaic is updated IN PLACE.  SYNCHRONOUS when result_host is given: returns after the kernels finished
 * (one small device-to-host copy + stream synchronisation).  With result_host == NULL the kernels are only
 * queued (graph capture, back-to-back timing) and isx_frame_result fetches the block later from the same stream.
 * result_host (ISX_FRAME_RESULT_INTS int32) =
 *   [0..4]   bbox of the warped image's gray != 0 pixels {x0,y0,x1,y1,found}            (stitcher.py:220)
 *   [5..8]   the gradient blender's ROI {x0,y0,x1,y1}                                  (utils/proc.py:199-201)
 *   [9..13]  bbox of the gray != 0 pixels written into the mosaic + found flag: the merged canvas's bbox
 *            (stitcher.py:225 crop_image) is its union with the bbox the mosaic had before ...
 *   [14]     ... provided this flag is 0: some pixel's gray was != 0 and became 0 (else rescan the arena)
 *   [15]     != 0: some pixel was written with gray == 0 but a non-zero byte: if it lies outside the new bbox the
 *            caller must clear it (the reference's crop drops it, the next padded canvas has 0 there)
 *   [16]     gradient: 1 unless a channel-0 != 0 pixel of an ROI row lies outside the ROI's columns; when 0
 *            the mosaic was NOT touched ([17] == 0) and the caller blends the ROI with isx_gradient_blend from
 *            the staged warped image (isx_frame_warped)
 *   [17]     1 if the blend ran */
int isx_frame_blend(isx_ctx *ctx, const uint8_t *src_dev, int sh, int sw, int c, int64_t spitch, int src_x, int src_y,
                    const double *M_host, int model, int arith, int canvas_w, int canvas_h, uint8_t *mosaic_q_dev,
                    int64_t mpitch, int q_x0, int q_y0, int q_w, int q_h, int clip_y0, int clip_y1, int blender,
                    double ref_contrib, int feather, int32_t *result_host, void *stream);
int isx_frame_result(isx_ctx *ctx, int32_t *result_host, void *stream);
/* Device address of the warped image staged by the last gradient-path isx_frame_blend (Q-sized, pitch q_w * c). */
int isx_frame_warped(isx_ctx *ctx, const uint8_t **wbuf_dev_out);

/* Composite, SYNCHRONOUS forms of the matcher and of the robust transform for the per-frame path: each stages its
 * host inputs through the ctx (one upload), queues the same kernels as the fine-grained entry points below /
 * above and reads the results back in one or two copies.  Same results as calling those one by one.
 *
 * isx_frame_match = Stitcher.get_matches + Stitcher.filter_matches (stitcher.py:106-127) for one pair:
 *   q_host / t_host : nq x d and nt x d descriptors, uint8 or float32 (ISX_NORM_L2: the exact integer path, as
 *                     isx_desc_prep + isx_knn2_l2_items + isx_ratio_compact; ISX_NORM_HAMMING: uint8 only)
 *   good_q/good_t_out_host : capacity nq int32 each; *n_good_out = len(good_matches), query order
 *   *status_out : 0 ok; 1 = float descriptors are not integer valued in [0,255]; 2 = 2*max||v||^2 reaches
 *                 ISX_L2_EXACT_LIMIT -- in both cases nothing is returned and the caller uses isx_knn2_f32_items */
int isx_frame_match(isx_ctx *ctx, const void *q_host, int nq, const void *t_host, int nt, int d, int dtype, int norm,
                    double ratio, int32_t *good_q_out_host, int32_t *good_t_out_host, int32_t *n_good_out,
                    int32_t *status_out, void *stream);
/* isx_frame_transform = Stitcher.get_transform (stitcher.py:135-141) with cv2's defaults: sample (host) ->
 * hypotheses -> score -> select (host) -> refine.  M_out_host: 6 (2x3) or 9 (3x3) doubles; mask_out_host: optional
 * n uint8 inlier mask; *found_out = 0 <=> cv2 returns None. */
int isx_frame_transform(isx_ctx *ctx, const float *src_host, const float *dst_host, int n, int model, double *M_out_host,
                        uint8_t *mask_out_host, int32_t *found_out, void *stream);
/* Pitched device-to-host download of a sub-rectangle into the ctx's pinned block, SYNCHRONOUS: the region of the
 * resident mosaic the CPU detector is run on (Stitcher.detectAndComputeFromCorners, stitcher.py:159-165 slices the
 * padded canvas there).  The block is out_h rows of out_row_bytes; the h x row_bytes source rectangle lands at row
 * off_y, byte off_x and the rest is zero (the canvas's padding).  *host_out stays valid until the next staging call
 * on this ctx. */
int isx_frame_download(isx_ctx *ctx, const uint8_t *src_dev, int64_t spitch, int h, int64_t row_bytes, int out_h,
                       int64_t out_row_bytes, int off_y, int64_t off_x, const uint8_t **host_out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * On-GPU ORB-style feature detection.  Stands where the reference calls
 * cv2.ORB_create().detectAndCompute (stitcher.py:92-100, :163, :193-194, :263), as an OPT-IN detector: it is not
 * cv2.ORB (cv2's learned pattern, pyramid and float arithmetic cannot be restated without cv2 -- parity with cv2 is
 * unpinned by construction); what it computes is defined in integer arithmetic by tests/np_orb.py and matched bit for
 * bit: FAST-9 corners, Harris ranking (k = 1/25), radius-3 NMS, intensity-centroid orientation in 32 bins, 256 steered
 * BRIEF bits on a 7x7-blurred image, single scale.
 *
 * isx_orb_workspace_bytes: device scratch the two calls below share for an h x w image (about 8.2 bytes per pixel;
 *   0 if the size is out of range: h, w <= 32767).
 * isx_orb_candidates: queues 4 kernels.  Outputs per 32 x 32 tile t (tiles row-major, ceil(w/32) per row):
 *   tile_count[t] survivors of the non-maximum suppression, their positions tile_pos[t*64 + i] = (y << 16) | x and
 *   ranking keys tile_key[t*64 + i] = bits of float32(Harris response) (order inside a tile is unspecified).
 *   fast_threshold: FAST-9 threshold (cv2.ORB default 20); clean: a corner is dropped when a gray == 0 pixel lies within
 *   `clean` pixels (0..16; 0 = off) -- keeps keypoints off the black padding of canvases and mosaics.
 *   The caller picks the keypoints (the nfeatures largest keys, ties to the lower y*w + x) and passes them back:
 * isx_orb_describe: one kernel; pos_dev[n] positions as above (must come from isx_orb_candidates on the same workspace
 *   contents; others yield angle -1 and zero bits), pattern_dev int8 [32][256][4] = the (x1,y1,x2,y2) test pairs for
 *   each orientation bin, |coordinate| <= 15.  Writes des_out_dev[n][32] (bit of pair 8j is bit 7 of byte j) and
 *   angle_out_dev[n] (bin 0..31 of 11.25 degrees).
 * ------------------------------------------------------------------------------------------------ */
int64_t isx_orb_workspace_bytes(int h, int w);
int isx_orb_candidates(const uint8_t *img_dev, int h, int w, int c, int64_t pitch, int fast_threshold, int clean,
                       void *workspace_dev, int32_t *tile_count_dev, int32_t *tile_pos_dev, uint32_t *tile_key_dev,
                       void *stream);
int isx_orb_describe(const void *workspace_dev, int h, int w, const int32_t *pos_dev, int n, const int8_t *pattern_dev,
                     uint8_t *des_out_dev, int32_t *angle_out_dev, void *stream);
/* The selection on the device followed by isx_orb_describe, without a host round trip in between (3 kernels).
 * sel_out_dev: int32 [2 + 2*nfeatures] = {n selected, survivors in total, n positions ((y << 16) | x) in rank order
 * padded to nfeatures, n keys padded to nfeatures}; des_out_dev [nfeatures][32] and angle_out_dev [nfeatures] are
 * written for the first n.  n = -1 when there are more than 65536 survivors (the selection ranks by counting, n^2
 * compares): the caller then selects on the host from the tile slots and calls isx_orb_describe.  nfeatures in
 * [1, 65536]; h, w >= 33.  Overwrites the workspace's blur scratch (isx_orb_candidates must run again before the next
 * isx_orb_select_describe; isx_orb_describe keeps working). */
int isx_orb_select_describe(const void *workspace_dev, int h, int w, const int32_t *tile_count_dev,
                            const int32_t *tile_pos_dev, const uint32_t *tile_key_dev, int nfeatures,
                            const int8_t *pattern_dev, int32_t *sel_out_dev, uint8_t *des_out_dev, int32_t *angle_out_dev,
                            void *stream);

/* ------------------------------------------------------------------------------------------------
 * Robust transform.  Replaces Stitcher.get_transform (stitcher.py:135-141):
 *   type 0: cv2.estimateAffinePartial2D(src, dst)       (RANSAC thr 3.0, 2000 iters, conf 0.99, 10 LM iters)
 *   type 1: cv2.findHomography(src, dst, cv2.RANSAC, 5.0) (2000 iters, conf 0.995, DLT refit + 10 LM iters)
 * Points are float32 (x, y) pairs, n of them (the (G,1,2) arrays of Stitcher.get_pts, stitcher.py:129-133).
 * The sequential cv loop's CONTROL FLOW is replayed by: sample (host) -> hypotheses (GPU) -> score (GPU) ->
 * select (host, the in-order "strictly better wins" + adaptive stop) -> refine (GPU).  The arithmetic follows cv's to
 * tolerance (models within 1e-6 relative of the CPU oracle's, same inlier counts on the same hypotheses), not bit for
 * bit: against cv2 itself this stage is "parity unpinned" (DESIGN.md section 2).
 * ------------------------------------------------------------------------------------------------ */

/* HOST.  cv::RNG(-1)-faithful subset list: up to max_iters subsets of 4 (homography) or 2 (affine
 * partial) distinct indices, each passing cv's checkSubset.  Returns the number of subsets written
 * (>= 0; < max_iters only if 10000 attempts failed, as cv gives up) or a negative status. */
int isx_ransac_sample(int model, const float *src_host, const float *dst_host, int n, int max_iters,
                      int32_t *subsets_out_host);

/* GPU.  One minimal model per subset: models_out_dev n_hyp x (9|6) float64 row-major, valid_out_dev n_hyp. */
int isx_ransac_hypotheses(int model, const float *src_dev, const float *dst_dev, int n,
                          const int32_t *subsets_dev, int n_hyp, double *models_out_dev,
                          int32_t *valid_out_dev, void *stream);

/* GPU.  Inlier count of every hypothesis (float32 reprojection error <= thr^2, cv computeError /
 * findInliers).  valid_dev may be NULL. */
int isx_ransac_score(int model, const double *models_dev, int n_hyp, const int32_t *valid_dev,
                     const float *src_dev, const float *dst_dev, int n, double thr, int32_t *counts_out_dev,
                     void *stream);

/* HOST.  Replays RANSACPointSetRegistrator::run over the counts: best_out = index of the winning
 * hypothesis (-1 = none), iters_out = iterations cv would have run. */
int isx_ransac_select(int model, const int32_t *counts_host, const int32_t *valid_host, int n_hyp,
                      int n_points, double confidence, int max_iters, int32_t *best_out, int32_t *iters_out);

/* GPU.  M_io_dev (9|6 float64): in = winning hypothesis, out = refined model.  Writes the winner's
 * inlier mask (n uint8) and info_out_dev[0] = inlier count.  refit != 0: all-inlier DLT first
 * (findHomography); max_iters: LM iterations (10 in the reference's calls; 0 = none). */
int isx_ransac_refine(int model, double *M_io_dev, const float *src_dev, const float *dst_dev, int n,
                      double thr, int refit, int max_iters, uint8_t *mask_out_dev, int32_t *info_out_dev,
                      void *stream);

/* HOST.  Union step of ArbitraryStitcher.get_collections (stitcher.py:283-296): applies union(first[e],
 * second[e]) for e = 0..m-1 to a DynamicConnectivity state (utils/structs.py:16-42: weighted quick-union,
 * path halving, ties hang q's root under p's).  parent_io / size_io: n int32 each, initialised by the caller
 * (parent[i] = i, size[i] = 1) and left in the state the reference's object would have. */
int isx_union_find(int n, const int32_t *first_host, const int32_t *second_host, int64_t m,
                   int32_t *parent_io_host, int32_t *size_io_host);

#ifdef __cplusplus
}
#endif
#endif

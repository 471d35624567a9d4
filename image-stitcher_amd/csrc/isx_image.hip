// Image-space kernels for gfx950: bbox of non-black pixels, warpAffine/warpPerspective (INTER_LINEAR,
// constant-0 border), the reference's two blenders.  All images are uint8, interleaved, PITCHED
// (row stride in bytes), c in {1,3}; a sub-rectangle of a larger device image is just a pointer to its
// first pixel plus the parent's pitch, so crops and ROIs cost nothing.
//
// These are HBM-bandwidth-bound byte kernels: one pass, coalesced row-major access, no LDS staging
// (there is no reuse to exploit except the 2x2 bilinear footprint, which L1/L2 serve).
#include "isx_common.h"
#include "isx_image_dev.h"

// bbox accumulator layout: {x0, y0, x1 (exclusive), y1 (exclusive), found}
__global__ void bbox_init_kernel(int32_t *bb) {
    if (threadIdx.x == 0) {
        bb[0] = INT_MAX;
        bb[1] = INT_MAX;
        bb[2] = -1;
        bb[3] = -1;
        bb[4] = 0;
    }
}

// Bounding boxes are reduced in two kernels: every workgroup stores its partial box with plain stores
// into a table, and a one-workgroup kernel on the same stream folds the table into the 5-int result.
// (Atomics on the accumulator itself serialise: all workgroups finish together, read the same untouched
// bounds and then queue ~10^4 atomics on one cache line -- measured +50 us on a 13 us warp.)  The table
// is CALLER-OWNED: it is the tail of the bbox accumulator (ints ISX_BBOX_HEAD .. ISX_BBOX_INTS of
// include/isx.h), so calls on different accumulators may overlap freely on one GPU.  The one exception is
// isx_bbox_nonzero_static (isx_bbox_static.hip), whose table is a library-owned static array.
#define ISX_BBOX_MAX_BLOCKS ISX_BBOX_PARTIAL_BLOCKS

// EVERY thread of a 256-thread block must call it
__device__ __forceinline__ void bbox_block_partial(int32_t *__restrict__ partial, int minx, int miny, int maxx,
                                                   int maxy) {
    __shared__ int red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a = wave_min(minx), b = wave_min(miny), c = wave_max(maxx), d = wave_max(maxy);
    if (lane == 0) {
        red[wave][0] = a;
        red[wave][1] = b;
        red[wave][2] = c;
        red[wave][3] = d;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int x0 = red[0][0], y0 = red[0][1], x1 = red[0][2], y1 = red[0][3];
        for (int w = 1; w < 4; w++) {
            x0 = min(x0, red[w][0]);
            y0 = min(y0, red[w][1]);
            x1 = max(x1, red[w][2]);
            y1 = max(y1, red[w][3]);
        }
        int32_t *o = partial + 4 * blockIdx.x;
        o[0] = x0, o[1] = y0, o[2] = x1, o[3] = y1;
    }
}

__global__ void __launch_bounds__(256) bbox_finish_kernel(int n_blocks, const int32_t *__restrict__ partial,
                                                          int32_t *__restrict__ bb) {
    int x0 = INT_MAX, y0 = INT_MAX, x1 = -1, y1 = -1;
    for (int i = threadIdx.x; i < n_blocks; i += 256) {
        const int32_t *o = partial + 4 * i;
        x0 = min(x0, o[0]);
        y0 = min(y0, o[1]);
        x1 = max(x1, o[2]);
        y1 = max(y1, o[3]);
    }
    __shared__ int red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    x0 = wave_min(x0), y0 = wave_min(y0), x1 = wave_max(x1), y1 = wave_max(y1);
    if (lane == 0) red[wave][0] = x0, red[wave][1] = y0, red[wave][2] = x1, red[wave][3] = y1;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) {
            x0 = min(x0, red[w][0]);
            y0 = min(y0, red[w][1]);
            x1 = max(x1, red[w][2]);
            y1 = max(y1, red[w][3]);
        }
        if (x1 >= 0) { // merge into the caller's accumulator
            bb[0] = min(bb[0], x0);
            bb[1] = min(bb[1], y0);
            bb[2] = max(bb[2], x1);
            bb[3] = max(bb[3], y1);
            bb[4] = 1;
        }
    }
}

// reference utils/proc.py:34-45 get_corners_from_image (cv2.findNonZero + cv2.boundingRect on gray).
// Tile = 256 pixels x 16 rows (wave w takes rows w, w+4, w+8, w+12; four 768-byte loads in flight).
template <int C>
__global__ void __launch_bounds__(256) bbox_nonzero_kernel(const uint8_t *__restrict__ img, int h, int w,
                                                           int64_t pitch, int tiles_x, int n_tiles,
                                                           int32_t *__restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int minx = INT_MAX, miny = INT_MAX, maxx = -1, maxy = -1;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int x = (t % tiles_x) * 256 + 4 * lane, y0 = (t / tiles_x) * 16 + wave;
        uint32_t px[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int y = y0 + 4 * i;
            quad_load<C>(img + (int64_t)min(y, h - 1) * pitch, x, y < h ? w : 0, px[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) bbox_quad<C>(px[i], x, y0 + 4 * i, minx, miny, maxx, maxy);
    }
    bbox_block_partial(partial, minx, miny, maxx, maxy);
}

int isx_bbox_nonzero_launch(const uint8_t *img_dev, int h, int w, int c, int64_t pitch, int32_t *result_dev,
                               int32_t *partial_dev, hipStream_t st) {
    ISX_REQUIRE(c == 1 || c == 3, "channels must be 1 or 3 (got %d)", c);
    ISX_REQUIRE(h >= 0 && w >= 0 && pitch >= (int64_t)w * c, "bad image geometry h=%d w=%d pitch=%lld", h, w, (long long)pitch);
    ISX_REQUIRE(result_dev && partial_dev, "null bbox accumulator");
    hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, st, result_dev);
    if (h > 0 && w > 0) {
        ISX_REQUIRE(img_dev, "null img_dev");
        const int tiles_x = (w + 255) / 256;
        const int64_t n_tiles = (int64_t)tiles_x * ((h + 15) / 16);
        ISX_REQUIRE(n_tiles < ((int64_t)1 << 31), "image too large");
        const unsigned grid = (unsigned)(n_tiles < ISX_BBOX_MAX_BLOCKS ? n_tiles : ISX_BBOX_MAX_BLOCKS);
        if (c == 3)
            hipLaunchKernelGGL((bbox_nonzero_kernel<3>), dim3(grid), dim3(256), 0, st, img_dev, h, w, pitch, tiles_x, (int)n_tiles, partial_dev);
        else
            hipLaunchKernelGGL((bbox_nonzero_kernel<1>), dim3(grid), dim3(256), 0, st, img_dev, h, w, pitch, tiles_x, (int)n_tiles, partial_dev);
        hipLaunchKernelGGL(bbox_finish_kernel, dim3(1), dim3(256), 0, st, (int)grid, partial_dev, result_dev);
    }
    return isx_launch_status("bbox_nonzero_kernel");
}

extern "C" int isx_bbox_nonzero(const uint8_t *img_dev, int h, int w, int c, int64_t pitch, int32_t *bbox_out_dev,
                                void *stream) {
    ISX_REQUIRE(bbox_out_dev, "null bbox_out_dev");
    return isx_bbox_nonzero_launch(img_dev, h, w, c, pitch, bbox_out_dev, bbox_out_dev + ISX_BBOX_HEAD, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// warp.  reference stitcher.py:143-149: cv2.warpAffine / cv2.warpPerspective(img, M, (W, H)) with the
// defaults INTER_LINEAR, BORDER_CONSTANT 0; M is the forward map.  [OCV-K] OpenCV 4.11 linear kernels:
// inverse map in double -> float32; coordinates, weights and the three lerps are float32 FMAs; result
// rounded to nearest even.  The source is the UN-padded frame placed at (src_x, src_y) of the virtual
// padded canvas (reference utils/proc.py:295-304), which is identical to sampling the zero-padded
// canvas; only the destination rectangle [dx0,dx0+dw) x [dy0,dy0+dh) of the canvas is produced.
// Interior pixels fetch the two taps of a row with ONE unaligned 8-byte (C=3) / 2-byte (C=1) load.
// ------------------------------------------------------------------------------------------------
// Tile = 256 pixels x 4 rows (one row per wave, a lane produces 4 adjacent pixels: eight gathers in
// flight per lane, one 12-byte store).  Capped, strided grid; per-block partial bbox (see above).
template <int C, bool PERSP, bool CLASSIC = false>
__global__ void __launch_bounds__(256) warp_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                   const WarpParams P, int tiles_x, int n_tiles,
                                                   int32_t *__restrict__ bbox_partial) {
    const bool want_bbox = bbox_partial != nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int minx = INT_MAX, miny = INT_MAX, maxx = -1, maxy = -1;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int lx = (t % tiles_x) * 256 + 4 * lane;
        const int ly = (t / tiles_x) * 4 + wave;
        if (ly >= P.dh) continue; // wave-uniform
        uint32_t px[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
            px[i] = lx + i >= P.dw ? 0u
                    : CLASSIC  ? warp_pixel_classic<C, PERSP>(src, P, P.dx0 + lx + i, P.dy0 + ly)
                               : warp_pixel<C, PERSP>(src, P, P.dx0 + lx + i, P.dy0 + ly);
        if (lx < P.dw) quad_store<C>(dst + (int64_t)ly * P.dpitch, lx, P.dw, px);
        if (want_bbox) bbox_quad<C>(px, P.dx0 + lx, P.dy0 + ly, minx, miny, maxx, maxy);
    }
    if (want_bbox) bbox_block_partial(bbox_partial, minx, miny, maxx, maxy);
}

extern "C" int isx_warp_arith(const uint8_t *src_dev, int sh, int sw, int c, int64_t spitch, int src_x, int src_y,
                              const double *M_host, int model, uint8_t *dst_dev, int64_t dpitch, int dx0, int dy0,
                              int dw, int dh, int arith, int canvas_w, int canvas_h, int32_t *bbox_out_dev,
                              void *stream) {
    ISX_REQUIRE(c == 1 || c == 3, "channels must be 1 or 3 (got %d)", c);
    ISX_REQUIRE(model == ISX_MODEL_AFFINE_PARTIAL || model == ISX_MODEL_HOMOGRAPHY, "bad model %d", model);
    ISX_REQUIRE(arith == ISX_WARP_LINEAR_F32 || arith == ISX_WARP_CLASSIC, "bad warp arithmetic %d", arith);
    ISX_REQUIRE(M_host, "null M_host");
    ISX_REQUIRE(sh > 0 && sw > 0 && spitch >= (int64_t)sw * c, "bad source geometry");
    ISX_REQUIRE(dw >= 0 && dh >= 0 && dpitch >= (int64_t)dw * c, "bad destination geometry");
    if (dw == 0 || dh == 0) return ISX_OK;
    ISX_REQUIRE(src_dev && dst_dev, "null device pointer");
    WarpParams P;
    double iM[9];
    if (model == ISX_MODEL_HOMOGRAPHY)
        invert_3x3_cv(M_host, iM);
    else
        invert_affine_cv(M_host, iM);
    for (int i = 0; i < 9; i++) P.m[i] = (float)iM[i], P.im[i] = iM[i];
    P.bw0 = 64;
    if (arith == ISX_WARP_CLASSIC && model == ISX_MODEL_HOMOGRAPHY) {
        // WarpPerspectiveInvoker's blocks on the FULL destination of the reference's call (the canvas)
        ISX_REQUIRE(canvas_w >= dx0 + dw && canvas_h >= dy0 + dh && dx0 >= 0 && dy0 >= 0,
                    "classic perspective warp needs the size of the full canvas (got %dx%d)", canvas_w, canvas_h);
        const int bh0 = canvas_h < 16 ? canvas_h : 16;
        P.bw0 = 1024 / bh0 < canvas_w ? 1024 / bh0 : canvas_w;
    }
    P.src_x = src_x; P.src_y = src_y; P.sh = sh; P.sw = sw;
    P.dx0 = dx0; P.dy0 = dy0; P.dw = dw; P.dh = dh;
    P.spitch = spitch; P.dpitch = dpitch;
    hipStream_t st = (hipStream_t)stream;
    const int tiles_x = (dw + 255) / 256;
    const int64_t n_tiles64 = (int64_t)tiles_x * ((dh + 3) / 4);
    ISX_REQUIRE(n_tiles64 < ((int64_t)1 << 31), "destination rectangle too large");
    const int n_tiles = (int)n_tiles64;
    const unsigned nblk = (unsigned)(n_tiles < ISX_BBOX_MAX_BLOCKS ? n_tiles : ISX_BBOX_MAX_BLOCKS);
    dim3 grid(nblk), block(256);
    const bool persp = model == ISX_MODEL_HOMOGRAPHY;
    int32_t *const wb = bbox_out_dev ? bbox_out_dev + ISX_BBOX_HEAD : nullptr;
#define ISX_WARP_LAUNCH(CC, PP, AA) \
    hipLaunchKernelGGL((warp_kernel<CC, PP, AA>), grid, block, 0, st, src_dev, dst_dev, P, tiles_x, n_tiles, wb)
    if (arith == ISX_WARP_CLASSIC) {
        if (c == 3 && persp) ISX_WARP_LAUNCH(3, true, true);
        else if (c == 3) ISX_WARP_LAUNCH(3, false, true);
        else if (persp) ISX_WARP_LAUNCH(1, true, true);
        else ISX_WARP_LAUNCH(1, false, true);
    } else {
        if (c == 3 && persp) ISX_WARP_LAUNCH(3, true, false);
        else if (c == 3) ISX_WARP_LAUNCH(3, false, false);
        else if (persp) ISX_WARP_LAUNCH(1, true, false);
        else ISX_WARP_LAUNCH(1, false, false);
    }
#undef ISX_WARP_LAUNCH
    if (wb) hipLaunchKernelGGL(bbox_finish_kernel, dim3(1), dim3(256), 0, st, (int)nblk, wb, bbox_out_dev);
    return isx_launch_status("warp_kernel");
}

extern "C" int isx_warp(const uint8_t *src_dev, int sh, int sw, int c, int64_t spitch, int src_x, int src_y,
                        const double *M_host, int model, uint8_t *dst_dev, int64_t dpitch, int dx0, int dy0,
                        int dw, int dh, int32_t *bbox_out_dev, void *stream) {
    return isx_warp_arith(src_dev, sh, sw, c, spitch, src_x, src_y, M_host, model, dst_dev, dpitch, dx0, dy0, dw, dh,
                          ISX_WARP_LINEAR_F32, 0, 0, bbox_out_dev, stream);
}

extern "C" int isx_bbox_reset(int32_t *bbox_dev, void *stream) {
    ISX_REQUIRE(bbox_dev, "null bbox_dev");
    hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, bbox_dev);
    return isx_launch_status("bbox_init_kernel");
}

// ------------------------------------------------------------------------------------------------
// alpha blender.  reference utils/proc.py:48-82 seamless_merge(image1=warped, image2=ref, contrib):
// overlap = gray1>10 & gray2>10; there ref' = (u8)trunc(c*ref), warped' = (u8)trunc((1-c)*warped) in
// float64; out = warped' + ref' as uint8 (wraps modulo 256).  out may alias ref.
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) alpha_blend_kernel(const uint8_t *__restrict__ warped, int64_t wpitch,
                                                          const uint8_t *ref, int64_t rpitch, uint8_t *out,
                                                          int64_t opitch, int h, int w, double contrib,
                                                          double one_minus) {
    // tile = 256 pixels x 8 rows; wave wv takes rows wv and wv + 4
    const int x = blockIdx.x * 256 + 4 * (threadIdx.x & 63);
    const int y0 = blockIdx.y * 8 + (threadIdx.x >> 6);
    if (x >= w) return;
    uint32_t pa[2][4], pb[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int y = y0 + 4 * i, yc = min(y, h - 1);
        quad_load<C>(warped + yc * wpitch, x, w, pa[i]);
        quad_load<C>(ref + yc * rpitch, x, w, pb[i]);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int y = y0 + 4 * i;
        if (y >= h) break;
        uint32_t po[4];
#pragma unroll
        for (int k = 0; k < 4; k++) po[k] = alpha_px<C>(pa[i][k], pb[i][k], contrib, one_minus);
        quad_store<C>(out + y * opitch, x, w, po);
    }
}

extern "C" int isx_alpha_blend(const uint8_t *warped_dev, int64_t wpitch, const uint8_t *ref_dev, int64_t rpitch,
                               uint8_t *out_dev, int64_t opitch, int h, int w, int c, double ref_contrib,
                               void *stream) {
    ISX_REQUIRE(c == 1 || c == 3, "channels must be 1 or 3 (got %d)", c);
    ISX_REQUIRE(h >= 0 && w >= 0, "negative size");
    if (h == 0 || w == 0) return ISX_OK;
    ISX_REQUIRE(warped_dev && ref_dev && out_dev, "null device pointer");
    const int64_t rb = (int64_t)w * c;
    ISX_REQUIRE(wpitch >= rb && rpitch >= rb && opitch >= rb, "pitch smaller than a row");
    const double one_minus = 1 - ref_contrib; // Python: (1-ref_image_contrib)
    dim3 grid((w + 255) / 256, (h + 7) / 8), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (c == 3)
        hipLaunchKernelGGL((alpha_blend_kernel<3>), grid, block, 0, st, warped_dev, wpitch, ref_dev, rpitch, out_dev, opitch, h, w, ref_contrib, one_minus);
    else
        hipLaunchKernelGGL((alpha_blend_kernel<1>), grid, block, 0, st, warped_dev, wpitch, ref_dev, rpitch, out_dev, opitch, h, w, ref_contrib, one_minus);
    return isx_launch_status("alpha_blend_kernel");
}

// Fused warp + alpha blend (Stitcher.warp_target + Stitcher.merge with blender = 0, stitcher.py:216-217):
// the warped pixel never goes to memory -- it is produced, its gray != 0 bbox accumulated (stitcher.py:220),
// blended with the mosaic pixel under it and stored in place.  One pass over the destination rectangle.
template <int C, bool PERSP, bool CLASSIC>
__global__ void __launch_bounds__(256) warp_alpha_kernel(const uint8_t *__restrict__ src, const uint8_t *ref,
                                                         int64_t rpitch, uint8_t *out, int64_t opitch,
                                                         const WarpParams P, int tiles_x, int n_tiles,
                                                         int32_t *__restrict__ bbox_partial, double contrib,
                                                         double one_minus) {
    const bool want_bbox = bbox_partial != nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int minx = INT_MAX, miny = INT_MAX, maxx = -1, maxy = -1;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int lx = (t % tiles_x) * 256 + 4 * lane;
        const int ly = (t / tiles_x) * 4 + wave;
        if (ly >= P.dh) continue; // wave-uniform
        uint32_t pb[4], px[4], po[4];
        quad_load<C>(ref + (int64_t)ly * rpitch, lx, P.dw, pb);
#pragma unroll
        for (int i = 0; i < 4; i++)
            px[i] = lx + i >= P.dw ? 0u
                    : CLASSIC  ? warp_pixel_classic<C, PERSP>(src, P, P.dx0 + lx + i, P.dy0 + ly)
                               : warp_pixel<C, PERSP>(src, P, P.dx0 + lx + i, P.dy0 + ly);
#pragma unroll
        for (int i = 0; i < 4; i++) po[i] = alpha_px<C>(px[i], pb[i], contrib, one_minus);
        if (lx < P.dw) quad_store<C>(out + (int64_t)ly * opitch, lx, P.dw, po);
        if (want_bbox) bbox_quad<C>(px, P.dx0 + lx, P.dy0 + ly, minx, miny, maxx, maxy);
    }
    if (want_bbox) bbox_block_partial(bbox_partial, minx, miny, maxx, maxy);
}

extern "C" int isx_warp_alpha_blend(const uint8_t *src_dev, int sh, int sw, int c, int64_t spitch, int src_x, int src_y,
                                    const double *M_host, int model, const uint8_t *ref_dev, int64_t rpitch,
                                    uint8_t *out_dev, int64_t opitch, int dx0, int dy0, int dw, int dh, int arith,
                                    int canvas_w, int canvas_h, double ref_contrib, int32_t *bbox_out_dev,
                                    void *stream) {
    ISX_REQUIRE(c == 1 || c == 3, "channels must be 1 or 3 (got %d)", c);
    ISX_REQUIRE(model == ISX_MODEL_AFFINE_PARTIAL || model == ISX_MODEL_HOMOGRAPHY, "bad model %d", model);
    ISX_REQUIRE(arith == ISX_WARP_LINEAR_F32 || arith == ISX_WARP_CLASSIC, "bad warp arithmetic %d", arith);
    ISX_REQUIRE(M_host, "null M_host");
    ISX_REQUIRE(sh > 0 && sw > 0 && spitch >= (int64_t)sw * c, "bad source geometry");
    ISX_REQUIRE(dw >= 0 && dh >= 0 && rpitch >= (int64_t)dw * c && opitch >= (int64_t)dw * c, "bad destination geometry");
    if (dw == 0 || dh == 0) return ISX_OK;
    ISX_REQUIRE(src_dev && ref_dev && out_dev, "null device pointer");
    WarpParams P;
    double iM[9];
    if (model == ISX_MODEL_HOMOGRAPHY)
        invert_3x3_cv(M_host, iM);
    else
        invert_affine_cv(M_host, iM);
    for (int i = 0; i < 9; i++) P.m[i] = (float)iM[i], P.im[i] = iM[i];
    P.bw0 = 64;
    if (arith == ISX_WARP_CLASSIC && model == ISX_MODEL_HOMOGRAPHY) {
        ISX_REQUIRE(canvas_w >= dx0 + dw && canvas_h >= dy0 + dh && dx0 >= 0 && dy0 >= 0,
                    "classic perspective warp needs the size of the full canvas (got %dx%d)", canvas_w, canvas_h);
        const int bh0 = canvas_h < 16 ? canvas_h : 16;
        P.bw0 = 1024 / bh0 < canvas_w ? 1024 / bh0 : canvas_w;
    }
    P.src_x = src_x; P.src_y = src_y; P.sh = sh; P.sw = sw;
    P.dx0 = dx0; P.dy0 = dy0; P.dw = dw; P.dh = dh;
    P.spitch = spitch; P.dpitch = opitch;
    hipStream_t st = (hipStream_t)stream;
    const int tiles_x = (dw + 255) / 256;
    const int64_t n_tiles64 = (int64_t)tiles_x * ((dh + 3) / 4);
    ISX_REQUIRE(n_tiles64 < ((int64_t)1 << 31), "destination rectangle too large");
    const int n_tiles = (int)n_tiles64;
    const unsigned nblk = (unsigned)(n_tiles < ISX_BBOX_MAX_BLOCKS ? n_tiles : ISX_BBOX_MAX_BLOCKS);
    dim3 grid(nblk), block(256);
    const bool persp = model == ISX_MODEL_HOMOGRAPHY, classic = arith == ISX_WARP_CLASSIC;
    int32_t *const wb = bbox_out_dev ? bbox_out_dev + ISX_BBOX_HEAD : nullptr;
    const double one_minus = 1 - ref_contrib;
#define ISX_WA_LAUNCH(CC, PP, AA)                                                                                    \
    hipLaunchKernelGGL((warp_alpha_kernel<CC, PP, AA>), grid, block, 0, st, src_dev, ref_dev, rpitch, out_dev, opitch, P, \
                       tiles_x, n_tiles, wb, ref_contrib, one_minus)
    if (c == 3) {
        if (persp && classic) ISX_WA_LAUNCH(3, true, true);
        else if (persp) ISX_WA_LAUNCH(3, true, false);
        else if (classic) ISX_WA_LAUNCH(3, false, true);
        else ISX_WA_LAUNCH(3, false, false);
    } else {
        if (persp && classic) ISX_WA_LAUNCH(1, true, true);
        else if (persp) ISX_WA_LAUNCH(1, true, false);
        else if (classic) ISX_WA_LAUNCH(1, false, true);
        else ISX_WA_LAUNCH(1, false, false);
    }
#undef ISX_WA_LAUNCH
    if (wb) hipLaunchKernelGGL(bbox_finish_kernel, dim3(1), dim3(256), 0, st, (int)nblk, wb, bbox_out_dev);
    return isx_launch_status("warp_alpha_kernel");
}

// ------------------------------------------------------------------------------------------------
// gradient (feather) blender.  reference utils/proc.py:192-254 seamless_gradient_merge, applied to the
// ROI the caller cut out (bbox(warped) +- 50 px, proc.py:199-205): `warped` and `ref` are the two ROI
// views, ref is updated in place (proc.py:252-253).
//   mask = float64(warped > 0) per channel                                  (:207-208)
//   horizontal pass per row, edges on channel 0: x in [L, min(L+f, W)) *= (x-L)/f ;
//                                                x in [max(0,R-f), R]  *= (R-x)/f       (:211-227)
//   vertical pass per column ON THE MODIFIED MASK (:230-246): channel 0 of the modified mask is > 0
//   exactly where the original is > 0 and x is neither L(y) nor R(y) (their factors are 0/f), so the
//   column extents are found without materialising the float64 mask.
//   result = trunc(warped*mask + ref*(1-mask)) in float64                    (:249, :253)
// ext layout (int32): left[h], right[h], top[w], bottom[w]   (left/top = INT_MAX, right/bottom = -1 if none)
// ------------------------------------------------------------------------------------------------
__global__ void ext_init_kernel(int32_t *ext, int h, int w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < h) {
        ext[i] = INT_MAX;
        ext[h + i] = -1;
    }
    if (i < w) {
        ext[2 * h + i] = INT_MAX;
        ext[2 * h + w + i] = -1;
    }
}

template <int C>
__global__ void __launch_bounds__(256) row_extent_kernel(const uint8_t *__restrict__ warped, int64_t wpitch, int h,
                                                         int w, int32_t *__restrict__ ext) {
    // one wave per (row, 1024-pixel span): 4 quads per lane
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (y >= h) return;
    int lo = INT_MAX, hi = -1;
    const uint8_t *row = warped + y * wpitch;
    uint32_t px[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) quad_load<C>(row, blockIdx.x * 1024 + i * 256 + 4 * lane, w, px[i]);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = blockIdx.x * 1024 + i * 256 + 4 * lane;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((px[i][k] & 255u) != 0u) { // channel 0; pixels beyond w read as 0
                lo = min(lo, x + k);
                hi = max(hi, x + k);
            }
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if (lane == 0 && hi >= 0) {
        atomicMin(&ext[y], lo);
        atomicMax(&ext[h + y], hi);
    }
}

template <int C>
__global__ void __launch_bounds__(256) col_extent_kernel(const uint8_t *__restrict__ warped, int64_t wpitch, int h,
                                                         int w, int32_t *__restrict__ ext) {
    // block = 256 columns x 32 rows; wave wv takes rows wv, wv+4, ... (8 row loads in flight, a lane owns 4
    // adjacent columns); the four waves are merged through LDS: one atomic pair per column and block
    __shared__ int slo[4][256], shi[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 256 + 4 * lane;
    const int ys = blockIdx.y * 32 + wave;
    int lo[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX}, hi[4] = {-1, -1, -1, -1};
    if (x < w) {
        uint32_t px[8][4];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int y = ys + 4 * i;
            quad_load<C>(warped + (int64_t)min(y, h - 1) * wpitch, x, y < h ? w : 0, px[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int y = ys + 4 * i, yc = min(y, h - 1);
            const int L = ext[yc], R = ext[h + yc];
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((px[i][k] & 255u) != 0u && x + k != L && x + k != R) { // rows beyond h read as 0
                    lo[k] = min(lo[k], y);
                    hi[k] = max(hi[k], y);
                }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        slo[wave][4 * lane + k] = lo[k];
        shi[wave][4 * lane + k] = hi[k];
    }
    __syncthreads();
    const int col = threadIdx.x, xx = blockIdx.x * 256 + col;
    const int l = min(min(slo[0][col], slo[1][col]), min(slo[2][col], slo[3][col]));
    const int u = max(max(shi[0][col], shi[1][col]), max(shi[2][col], shi[3][col]));
    if (xx < w && u >= 0) {
        atomicMin(&ext[2 * h + xx], l);
        atomicMax(&ext[2 * h + w + xx], u);
    }
}

template <int C>
__global__ void __launch_bounds__(256) gradient_blend_kernel(const uint8_t *__restrict__ warped, int64_t wpitch,
                                                             uint8_t *ref, int64_t rpitch, int h, int w, int feather,
                                                             const int32_t *__restrict__ ext) {
    // tile = 256 pixels x 8 rows; wave wv takes rows wv and wv + 4; a lane owns 4 adjacent pixels
    const int x = blockIdx.x * 256 + 4 * (threadIdx.x & 63);
    const int y0 = blockIdx.y * 8 + (threadIdx.x >> 6);
    if (x >= w) return;
    uint32_t pw[2][4], pr[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int yc = min(y0 + 4 * i, h - 1);
        quad_load<C>(warped + yc * wpitch, x, w, pw[i]);
        quad_load<C>(ref + yc * rpitch, x, w, pr[i]);
    }
    const double f = (double)feather;
    int T[4], B[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int xc = min(x + k, w - 1);
        T[k] = ext[2 * h + xc];
        B[k] = ext[2 * h + w + xc];
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int y = y0 + 4 * i;
        if (y >= h) break;
        const int L = ext[y], R = ext[h + y];
        uint32_t po[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int xx = x + k;
            // the factor chain is the same for every channel of the pixel; applied to the mask=1 channels
            double m = 1.0;
            if (R >= 0) {
                const int xe = L + feather < w ? L + feather : w;
                if (xx >= L && xx < xe) m *= (double)(xx - L) / f;
                const int xb = R - feather > 0 ? R - feather : 0;
                if (xx >= xb && xx <= R) m *= (double)(R - xx) / f;
            }
            if (B[k] >= 0) {
                const int ye = T[k] + feather < h ? T[k] + feather : h;
                if (y >= T[k] && y < ye) m *= (double)(y - T[k]) / f;
                const int yb = B[k] - feather > 0 ? B[k] - feather : 0;
                if (y >= yb && y <= B[k]) m *= (double)(B[k] - y) / f;
            }
            po[k] = 0;
#pragma unroll
            for (int c = 0; c < C; c++) {
                const int wv = (pw[i][k] >> (8 * c)) & 255, rv = (pr[i][k] >> (8 * c)) & 255;
                const double mk = wv > 0 ? m : 0.0;
                const double r = (double)wv * mk + (double)rv * (1 - mk);
                po[k] |= (uint32_t)(uint8_t)(int)r << (8 * c);
            }
        }
        quad_store<C>(ref + y * rpitch, x, w, po);
    }
}

extern "C" int isx_gradient_blend(const uint8_t *warped_dev, int64_t wpitch, uint8_t *ref_dev, int64_t rpitch, int h,
                                  int w, int c, int feather, int32_t *ext_dev, void *stream) {
    ISX_REQUIRE(c == 1 || c == 3, "channels must be 1 or 3 (got %d)", c);
    ISX_REQUIRE(h >= 0 && w >= 0 && feather > 0, "bad size / feather");
    if (h == 0 || w == 0) return ISX_OK;
    ISX_REQUIRE(warped_dev && ref_dev && ext_dev, "null device pointer");
    const int64_t rb = (int64_t)w * c;
    ISX_REQUIRE(wpitch >= rb && rpitch >= rb, "pitch smaller than a row");
    hipStream_t st = (hipStream_t)stream;
    const int n = h > w ? h : w;
    hipLaunchKernelGGL(ext_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ext_dev, h, w);
    dim3 gr((w + 1023) / 1024, (h + 3) / 4), gc((w + 255) / 256, (h + 31) / 32), gb((w + 255) / 256, (h + 7) / 8);
    if (c == 3) {
        hipLaunchKernelGGL((row_extent_kernel<3>), gr, dim3(256), 0, st, warped_dev, wpitch, h, w, ext_dev);
        hipLaunchKernelGGL((col_extent_kernel<3>), gc, dim3(256), 0, st, warped_dev, wpitch, h, w, ext_dev);
        hipLaunchKernelGGL((gradient_blend_kernel<3>), gb, dim3(256), 0, st, warped_dev, wpitch, ref_dev, rpitch, h, w, feather, ext_dev);
    } else {
        hipLaunchKernelGGL((row_extent_kernel<1>), gr, dim3(256), 0, st, warped_dev, wpitch, h, w, ext_dev);
        hipLaunchKernelGGL((col_extent_kernel<1>), gc, dim3(256), 0, st, warped_dev, wpitch, h, w, ext_dev);
        hipLaunchKernelGGL((gradient_blend_kernel<1>), gb, dim3(256), 0, st, warped_dev, wpitch, ref_dev, rpitch, h, w, feather, ext_dev);
    }
    return isx_launch_status("gradient_blend kernels");
}

// pitched device-to-device copy used by the mosaic arena
// Zero up to four byte rectangles of a pitched image in ONE launch (blockIdx.y = rectangle).
struct ClearRects {
    int32_t r[4][4]; // {row0, row1, byte0, byte1} per rectangle
};
__global__ void __launch_bounds__(256) clear_rects_kernel(uint8_t *__restrict__ dst, int64_t pitch, const ClearRects R) {
    const int y0 = R.r[blockIdx.y][0], y1 = R.r[blockIdx.y][1], b0 = R.r[blockIdx.y][2], b1 = R.r[blockIdx.y][3];
    for (int y = y0 + blockIdx.x; y < y1; y += gridDim.x) {
        uint8_t *row = dst + (int64_t)y * pitch;
        // bytes up to the first 4-byte boundary and after the last one singly, dwords in between
        const int a0 = min(b1, b0 + (int)((4 - ((uintptr_t)(row + b0) & 3)) & 3)), a1 = a0 + ((b1 - a0) & ~3);
        for (int b = b0 + threadIdx.x; b < a0; b += 256) row[b] = 0;
        for (int b = a0 + 4 * threadIdx.x; b < a1; b += 1024) *(uint32_t *)(row + b) = 0u;
        for (int b = a1 + threadIdx.x; b < b1; b += 256) row[b] = 0;
    }
}

extern "C" int isx_clear_rects(uint8_t *dst_dev, int64_t dpitch, int h, int64_t row_bytes, const int32_t *rects_host, int n,
                               void *stream) {
    ISX_REQUIRE(n >= 0 && n <= 4, "at most four rectangles per call (got %d)", n);
    if (n == 0) return ISX_OK;
    ISX_REQUIRE(dst_dev && rects_host && h >= 0 && row_bytes >= 0 && dpitch >= row_bytes && row_bytes < ((int64_t)1 << 31),
                "bad clear_rects arguments");
    ClearRects R;
    int rows = 0;
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 4; k++) R.r[i][k] = 0;
    for (int i = 0; i < n; i++) {
        const int32_t *q = rects_host + 4 * i;
        ISX_REQUIRE(q[0] >= 0 && q[0] <= q[1] && q[1] <= h && q[2] >= 0 && q[2] <= q[3] && q[3] <= row_bytes,
                    "rectangle %d = rows %d..%d, bytes %d..%d outside the %d x %lld image", i, q[0], q[1], q[2], q[3], h,
                    (long long)row_bytes);
        for (int k = 0; k < 4; k++) R.r[i][k] = q[k];
        rows = rows > q[1] - q[0] ? rows : q[1] - q[0];
    }
    if (rows == 0) return ISX_OK;
    hipLaunchKernelGGL(clear_rects_kernel, dim3(rows < 512 ? rows : 512, n), dim3(256), 0, (hipStream_t)stream, dst_dev, dpitch, R);
    return isx_launch_status("clear_rects_kernel");
}

extern "C" int isx_copy2d(const uint8_t *src_dev, int64_t spitch, uint8_t *dst_dev, int64_t dpitch, int h,
                          int64_t row_bytes, void *stream) {
    ISX_REQUIRE(h >= 0 && row_bytes >= 0, "negative size");
    if (h == 0 || row_bytes == 0) return ISX_OK;
    ISX_REQUIRE(src_dev && dst_dev && spitch >= row_bytes && dpitch >= row_bytes, "bad copy2d arguments");
    ISX_HIP_TRY(hipMemcpy2DAsync(dst_dev, (size_t)dpitch, src_dev, (size_t)spitch, (size_t)row_bytes, (size_t)h,
                                 hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return ISX_OK;
}

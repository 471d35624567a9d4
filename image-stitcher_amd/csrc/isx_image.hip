// Image-space kernels for gfx950: bbox of non-black pixels, warpAffine/warpPerspective (INTER_LINEAR,
// constant-0 border), the reference's two blenders.  All images are uint8, interleaved, PITCHED
// (row stride in bytes), c in {1,3}; a sub-rectangle of a larger device image is just a pointer to its
// first pixel plus the parent's pitch, so crops and ROIs cost nothing.
//
// These are HBM-bandwidth-bound byte kernels: one pass, coalesced row-major access, no LDS staging
// (there is no reuse to exploit except the 2x2 bilinear footprint, which L1/L2 serve).
#include "isx_common.h"
#include <limits.h>

// [OCV-K] cv2.cvtColor(BGR2GRAY) 8-bit; reference utils/proc.py:36, :51-52
__device__ __forceinline__ int gray3(int b, int g, int r) { return (b * 3735 + g * 19235 + r * 9798 + 16384) >> 15; }
template <int C>
__device__ __forceinline__ int gray_px(uint32_t px) {
    return C == 3 ? gray3(px & 255, (px >> 8) & 255, (px >> 16) & 255) : (int)(px & 255);
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int o = __shfl_xor(v, m);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int o = __shfl_xor(v, m);
        v = o > v ? o : v;
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// Row segments.  A wave moves 64 pixels of one row as a byte stream of 64*C bytes: lane k loads / stores
// the dword at byte 4k (any alignment: crops start at arbitrary x) and the pixels are cut out of / packed
// into that stream with two cross-lane reads.  One dword instruction per 64 pixels instead of C byte
// instructions, which is what lets these byte kernels approach the HBM rate.
// ------------------------------------------------------------------------------------------------
typedef uint32_t u32u __attribute__((aligned(1)));
typedef uint64_t u64u __attribute__((aligned(1)));

// dword `lane` of a segment of `nbytes` valid bytes (0 beyond the end)
__device__ __forceinline__ uint32_t seg_load(const uint8_t *__restrict__ base, int nbytes, int lane) {
    const int o = 4 * lane;
    uint32_t v = 0;
    if (o + 4 <= nbytes) {
        v = *(const u32u *)(base + o);
    } else {
        for (int j = 0; j < 4; j++)
            if (o + j < nbytes) v |= (uint32_t)base[o + j] << (8 * j);
    }
    return v;
}
__device__ __forceinline__ void seg_store(uint8_t *__restrict__ base, int nbytes, int lane, uint32_t v) {
    const int o = 4 * lane;
    if (o + 4 <= nbytes) {
        *(u32u *)(base + o) = v;
    } else {
        for (int j = 0; j < 4; j++)
            if (o + j < nbytes) base[o + j] = (uint8_t)(v >> (8 * j));
    }
}
// the C bytes of pixel `lane` (low bits) out of the dword stream
template <int C>
__device__ __forceinline__ uint32_t seg_pixel(uint32_t dw, int lane) {
    const int b = C * lane, d0 = b >> 2, sh = (b & 3) * 8;
    const uint32_t lo = __shfl(dw, d0), hi = __shfl(dw, d0 + 1);
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    return (uint32_t)(v >> sh) & (C == 3 ? 0xFFFFFFu : 0xFFu);
}
// inverse: every lane gives its pixel, lane k gets dword k of the stream (lanes >= 16*C get garbage)
template <int C>
__device__ __forceinline__ uint32_t seg_pack(uint32_t px, int lane) {
    if (C == 3) {
        const int p0 = (4 * lane) / 3, off = (4 * lane) % 3;
        const uint64_t a = __shfl(px, p0), b = __shfl(px, p0 + 1);
        return (uint32_t)((a | (b << 24)) >> (8 * off));
    }
    const uint32_t a = __shfl(px, 4 * lane), b = __shfl(px, 4 * lane + 1);
    const uint32_t c = __shfl(px, 4 * lane + 2), d = __shfl(px, 4 * lane + 3);
    return (a & 255u) | ((b & 255u) << 8) | ((c & 255u) << 16) | (d << 24);
}

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

// Workgroup-level bbox update (EVERY thread of a 256-thread block must call it): wave reduce, merge of
// the 4 waves through LDS, then one guarded atomic per bound.  The guard reads the current bound first
// (a stale value is only ever looser, so no needed update is skipped): without it tens of thousands of
// waves serialise on five addresses (measured: 1.9 ms for one 1080p warp).
__device__ __forceinline__ void bbox_block_accumulate(int minx, int miny, int maxx, int maxy, int32_t *bb) {
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
        if (x1 >= 0) {
            if (x0 < __hip_atomic_load(&bb[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&bb[0], x0);
            if (y0 < __hip_atomic_load(&bb[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&bb[1], y0);
            if (x1 > __hip_atomic_load(&bb[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&bb[2], x1);
            if (y1 > __hip_atomic_load(&bb[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&bb[3], y1);
            if (__hip_atomic_load(&bb[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&bb[4], 1);
        }
    }
}

// reference utils/proc.py:34-45 get_corners_from_image (cv2.findNonZero + cv2.boundingRect on gray).
// Tile = 64 pixels x 32 rows (wave w takes rows w, w+4, ...).  The grid is capped and every block walks
// tiles blockIdx, blockIdx + grid, ...: its local box is then close to the global one, so after the
// first few blocks have published, the guarded atomics of bbox_block_accumulate are skipped.
template <int C>
__global__ void __launch_bounds__(256) bbox_nonzero_kernel(const uint8_t *__restrict__ img, int h, int w,
                                                           int64_t pitch, int tiles_x, int n_tiles,
                                                           int32_t *__restrict__ bb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int minx = INT_MAX, miny = INT_MAX, maxx = -1, maxy = -1;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int xs = (t % tiles_x) * 64, x = xs + lane, y0 = (t / tiles_x) * 32 + wave;
        const int nvalid = min(64, w - xs);
        uint32_t dw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int y = y0 + 4 * i;
            dw[i] = y < h ? seg_load(img + y * pitch + (int64_t)xs * C, nvalid * C, lane) : 0u;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int y = y0 + 4 * i;
            const uint32_t px = seg_pixel<C>(dw[i], lane);
            if (lane < nvalid && gray_px<C>(px) != 0) { // rows beyond h loaded 0
                minx = minx < x ? minx : x;
                maxx = maxx > x + 1 ? maxx : x + 1;
                miny = miny < y ? miny : y;
                maxy = maxy > y + 1 ? maxy : y + 1;
            }
        }
    }
    bbox_block_accumulate(minx, miny, maxx, maxy, bb);
}

extern "C" int isx_bbox_nonzero(const uint8_t *img_dev, int h, int w, int c, int64_t pitch, int32_t *bbox_out_dev,
                                void *stream) {
    ISX_REQUIRE(c == 1 || c == 3, "channels must be 1 or 3 (got %d)", c);
    ISX_REQUIRE(h >= 0 && w >= 0 && pitch >= (int64_t)w * c, "bad image geometry h=%d w=%d pitch=%lld", h, w, (long long)pitch);
    ISX_REQUIRE(bbox_out_dev, "null bbox_out_dev");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, st, bbox_out_dev);
    if (h > 0 && w > 0) {
        ISX_REQUIRE(img_dev, "null img_dev");
        const int tiles_x = (w + 63) / 64;
        const int64_t n_tiles = (int64_t)tiles_x * ((h + 31) / 32);
        ISX_REQUIRE(n_tiles < ((int64_t)1 << 31), "image too large");
        const unsigned grid = (unsigned)(n_tiles < 1024 ? n_tiles : 1024);
        if (c == 3)
            hipLaunchKernelGGL((bbox_nonzero_kernel<3>), dim3(grid), dim3(256), 0, st, img_dev, h, w, pitch, tiles_x, (int)n_tiles, bbox_out_dev);
        else
            hipLaunchKernelGGL((bbox_nonzero_kernel<1>), dim3(grid), dim3(256), 0, st, img_dev, h, w, pitch, tiles_x, (int)n_tiles, bbox_out_dev);
    }
    return isx_launch_status("bbox_nonzero_kernel");
}

// ------------------------------------------------------------------------------------------------
// warp.  reference stitcher.py:143-149: cv2.warpAffine / cv2.warpPerspective(img, M, (W, H)) with the
// defaults INTER_LINEAR, BORDER_CONSTANT 0; M is the forward map.  [OCV-K] OpenCV 4.11 linear kernels:
// inverse map in double -> float32; coordinates, weights and the three lerps are float32 FMAs; result
// rounded to nearest even.  The source is the UN-padded frame placed at (src_x, src_y) of the virtual
// padded canvas (reference utils/proc.py:295-304), which is identical to sampling the zero-padded
// canvas; only the destination rectangle [dx0,dx0+dw) x [dy0,dy0+dh) of the canvas is produced.
// Interior pixels fetch the two taps of a row with ONE unaligned 8-byte (C=3) / 2-byte (C=1) load; the
// output row segment is packed across lanes and stored as dwords.
// ------------------------------------------------------------------------------------------------
struct WarpParams {
    float m[9];
    int src_x, src_y, sh, sw;
    int dx0, dy0, dw, dh;
    int64_t spitch, dpitch;
};

// one output pixel (packed B | G<<8 | R<<16) at canvas (x, y)
template <int C, bool PERSP>
__device__ __forceinline__ uint32_t warp_pixel(const uint8_t *__restrict__ src, const WarpParams &P, int x, int y) {
    const float fx = (float)x, fy = (float)y;
    float sx = fmaf(P.m[0], fx, fmaf(P.m[1], fy, P.m[2]));
    float sy = fmaf(P.m[3], fx, fmaf(P.m[4], fy, P.m[5]));
    if (PERSP) {
        const float w = fmaf(P.m[6], fx, fmaf(P.m[7], fy, P.m[8]));
        sx = sx / w;
        sy = sy / w;
    }
    // same guard as the oracle: coordinates are compared in canvas space
    const float cw = (float)(P.src_x + P.sw), ch = (float)(P.src_y + P.sh);
    if (!(sx > (float)P.src_x - 2.f && sx < cw + 1.f && sy > (float)P.src_y - 2.f && sy < ch + 1.f)) return 0u;
    const float flx = floorf(sx), fly = floorf(sy);
    const int ix = (int)flx - P.src_x, iy = (int)fly - P.src_y; // frame coordinates
    const float a = sx - flx, b = sy - fly;
    const bool x0ok = (unsigned)ix < (unsigned)P.sw, x1ok = (unsigned)(ix + 1) < (unsigned)P.sw;
    const bool y0ok = (unsigned)iy < (unsigned)P.sh, y1ok = (unsigned)(iy + 1) < (unsigned)P.sh;
    const uint8_t *r0 = src + (int64_t)iy * P.spitch + (int64_t)ix * C;
    const uint8_t *r1 = r0 + P.spitch;
    uint32_t p00[3], p01[3], p10[3], p11[3];
    // fast path: both taps of both rows inside the frame and the wide load stays inside the row
    const bool wide = x0ok && y0ok && y1ok && (C == 3 ? ix * 3 + 8 <= P.sw * 3 : x1ok);
    if (wide) {
        if (C == 3) {
            const uint64_t q0 = *(const u64u *)r0, q1 = *(const u64u *)r1;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                p00[c] = (uint32_t)(q0 >> (8 * c)) & 255u;
                p01[c] = (uint32_t)(q0 >> (8 * (c + 3))) & 255u;
                p10[c] = (uint32_t)(q1 >> (8 * c)) & 255u;
                p11[c] = (uint32_t)(q1 >> (8 * (c + 3))) & 255u;
            }
        } else {
            typedef uint16_t u16u __attribute__((aligned(1)));
            const uint32_t q0 = *(const u16u *)r0, q1 = *(const u16u *)r1;
            p00[0] = q0 & 255u;
            p01[0] = q0 >> 8;
            p10[0] = q1 & 255u;
            p11[0] = q1 >> 8;
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; c++) {
            p00[c] = (x0ok && y0ok) ? r0[c] : 0u;
            p01[c] = (x1ok && y0ok) ? r0[C + c] : 0u;
            p10[c] = (x0ok && y1ok) ? r1[c] : 0u;
            p11[c] = (x1ok && y1ok) ? r1[C + c] : 0u;
        }
    }
    uint32_t px = 0;
#pragma unroll
    for (int c = 0; c < C; c++) {
        const float f00 = (float)p00[c], f01 = (float)p01[c], f10 = (float)p10[c], f11 = (float)p11[c];
        const float t0 = fmaf(a, f01 - f00, f00);
        const float t1 = fmaf(a, f11 - f10, f10);
        const float v = fmaf(b, t1 - t0, t0);
        const float r = rintf(v);
        px |= (uint32_t)(r < 0.f ? 0 : (r > 255.f ? 255 : (int)r)) << (8 * c);
    }
    return px;
}

// Tile = 64 pixels x 16 rows (wave w takes rows w, w+4, w+8, w+12: four gathers in flight per lane); the
// grid is capped and strided like bbox_nonzero_kernel so that the bbox atomics stay rare.
template <int C, bool PERSP>
__global__ void __launch_bounds__(256) warp_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                   const WarpParams P, int tiles_x, int n_tiles,
                                                   int32_t *__restrict__ bb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int minx = INT_MAX, miny = INT_MAX, maxx = -1, maxy = -1;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int lx0 = (t % tiles_x) * 64, lx = lx0 + lane;
        const int ly0 = (t / tiles_x) * 16 + wave;
        const int nvalid = min(64, P.dw - lx0);
        const int x = P.dx0 + lx;
        uint32_t px[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int ly = ly0 + 4 * i;
            px[i] = (lx < P.dw && ly < P.dh) ? warp_pixel<C, PERSP>(src, P, x, P.dy0 + ly) : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int ly = ly0 + 4 * i;
            if (ly < P.dh) // wave-uniform
                seg_store(dst + (int64_t)ly * P.dpitch + (int64_t)lx0 * C, nvalid * C, lane, seg_pack<C>(px[i], lane));
            if (bb && gray_px<C>(px[i]) != 0) { // pixels outside the rectangle are 0
                const int y = P.dy0 + ly;
                minx = minx < x ? minx : x;
                maxx = maxx > x + 1 ? maxx : x + 1;
                miny = miny < y ? miny : y;
                maxy = maxy > y + 1 ? maxy : y + 1;
            }
        }
    }
    if (bb) bbox_block_accumulate(minx, miny, maxx, maxy, bb);
}

// host-side inverse exactly as cv::warpAffine / cv::warpPerspective compute it (double)
static void invert_affine_cv(const double *M, double *iM) {
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    iM[0] = A11;
    iM[1] = M[1] * (-D);
    iM[3] = M[3] * (-D);
    iM[4] = A22;
    iM[2] = -iM[0] * M[2] - iM[1] * M[5];
    iM[5] = -iM[3] * M[2] - iM[4] * M[5];
    iM[6] = 0;
    iM[7] = 0;
    iM[8] = 1;
}
static void invert_3x3_cv(const double *m, double *o) {
    double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
               m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (d != 0.) d = 1. / d;
    o[0] = (m[4] * m[8] - m[5] * m[7]) * d;
    o[1] = (m[2] * m[7] - m[1] * m[8]) * d;
    o[2] = (m[1] * m[5] - m[2] * m[4]) * d;
    o[3] = (m[5] * m[6] - m[3] * m[8]) * d;
    o[4] = (m[0] * m[8] - m[2] * m[6]) * d;
    o[5] = (m[2] * m[3] - m[0] * m[5]) * d;
    o[6] = (m[3] * m[7] - m[4] * m[6]) * d;
    o[7] = (m[1] * m[6] - m[0] * m[7]) * d;
    o[8] = (m[0] * m[4] - m[1] * m[3]) * d;
}

extern "C" int isx_warp(const uint8_t *src_dev, int sh, int sw, int c, int64_t spitch, int src_x, int src_y,
                        const double *M_host, int model, uint8_t *dst_dev, int64_t dpitch, int dx0, int dy0,
                        int dw, int dh, int32_t *bbox_out_dev, void *stream) {
    ISX_REQUIRE(c == 1 || c == 3, "channels must be 1 or 3 (got %d)", c);
    ISX_REQUIRE(model == ISX_MODEL_AFFINE_PARTIAL || model == ISX_MODEL_HOMOGRAPHY, "bad model %d", model);
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
    for (int i = 0; i < 9; i++) P.m[i] = (float)iM[i];
    P.src_x = src_x; P.src_y = src_y; P.sh = sh; P.sw = sw;
    P.dx0 = dx0; P.dy0 = dy0; P.dw = dw; P.dh = dh;
    P.spitch = spitch; P.dpitch = dpitch;
    hipStream_t st = (hipStream_t)stream;
    const int tiles_x = (dw + 63) / 64;
    const int64_t n_tiles64 = (int64_t)tiles_x * ((dh + 15) / 16);
    ISX_REQUIRE(n_tiles64 < ((int64_t)1 << 31), "destination rectangle too large");
    const int n_tiles = (int)n_tiles64;
    dim3 grid((unsigned)(n_tiles < 2048 ? n_tiles : 2048)), block(256);
    const bool persp = model == ISX_MODEL_HOMOGRAPHY;
    if (c == 3 && persp) hipLaunchKernelGGL((warp_kernel<3, true>), grid, block, 0, st, src_dev, dst_dev, P, tiles_x, n_tiles, bbox_out_dev);
    else if (c == 3) hipLaunchKernelGGL((warp_kernel<3, false>), grid, block, 0, st, src_dev, dst_dev, P, tiles_x, n_tiles, bbox_out_dev);
    else if (persp) hipLaunchKernelGGL((warp_kernel<1, true>), grid, block, 0, st, src_dev, dst_dev, P, tiles_x, n_tiles, bbox_out_dev);
    else hipLaunchKernelGGL((warp_kernel<1, false>), grid, block, 0, st, src_dev, dst_dev, P, tiles_x, n_tiles, bbox_out_dev);
    return isx_launch_status("warp_kernel");
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
    // tile = 64 pixels x 16 rows; wave wv takes rows wv, wv+4, wv+8, wv+12 (8 row segments in flight)
    const int lane = threadIdx.x & 63;
    const int xs = blockIdx.x * 64;
    const int y0 = blockIdx.y * 16 + (threadIdx.x >> 6);
    const int nb = min(64, w - xs) * C;
    uint32_t d1[4], d2[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int y = y0 + 4 * i;
        d1[i] = y < h ? seg_load(warped + y * wpitch + (int64_t)xs * C, nb, lane) : 0u;
        d2[i] = y < h ? seg_load(ref + y * rpitch + (int64_t)xs * C, nb, lane) : 0u;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int y = y0 + 4 * i;
        if (y >= h) break; // wave-uniform
        const uint32_t pa = seg_pixel<C>(d1[i], lane), pb = seg_pixel<C>(d2[i], lane);
        const bool overlap = gray_px<C>(pa) > 10 && gray_px<C>(pb) > 10;
        uint32_t po = 0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            int va = (pa >> (8 * c)) & 255, vb = (pb >> (8 * c)) & 255;
            if (overlap) {
                vb = (int)(contrib * (double)vb);
                va = (int)(one_minus * (double)va);
            }
            po |= (uint32_t)((va + vb) & 255) << (8 * c);
        }
        seg_store(out + y * opitch + (int64_t)xs * C, nb, lane, seg_pack<C>(po, lane));
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
    dim3 grid((w + 63) / 64, (h + 15) / 16), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (c == 3)
        hipLaunchKernelGGL((alpha_blend_kernel<3>), grid, block, 0, st, warped_dev, wpitch, ref_dev, rpitch, out_dev, opitch, h, w, ref_contrib, one_minus);
    else
        hipLaunchKernelGGL((alpha_blend_kernel<1>), grid, block, 0, st, warped_dev, wpitch, ref_dev, rpitch, out_dev, opitch, h, w, ref_contrib, one_minus);
    return isx_launch_status("alpha_blend_kernel");
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
    // one wave per (row, 1024-pixel span)
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (y >= h) return;
    int lo = INT_MAX, hi = -1;
    const uint8_t *row = warped + y * wpitch;
#pragma unroll 8
    for (int i = 0; i < 16; i++) {
        const int xs = blockIdx.x * 1024 + i * 64;
        if (xs < w) { // wave-uniform
            const int nv = min(64, w - xs);
            const uint32_t px = seg_pixel<C>(seg_load(row + (int64_t)xs * C, nv * C, lane), lane);
            if (lane < nv && (px & 255u) != 0u) {
                lo = lo < xs + lane ? lo : xs + lane;
                hi = xs + lane;
            }
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
    // wave = 64 adjacent columns x 32 rows (one coalesced row segment per step); lane owns a column
    const int lane = threadIdx.x & 63;
    const int xs = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (xs >= w) return; // wave-uniform
    const int nv = min(64, w - xs), x = xs + lane;
    const int ys = blockIdx.y * 32;
    int lo = INT_MAX, hi = -1;
#pragma unroll 8
    for (int i = 0; i < 32; i++) {
        const int y = ys + i;
        if (y < h) {
            const uint32_t px = seg_pixel<C>(seg_load(warped + y * wpitch + (int64_t)xs * C, nv * C, lane), lane);
            const bool on = lane < nv && (px & 255u) != 0u && x != ext[y] && x != ext[h + y];
            if (on) {
                lo = lo < y ? lo : y;
                hi = y;
            }
        }
    }
    if (hi >= 0) {
        atomicMin(&ext[2 * h + x], lo);
        atomicMax(&ext[2 * h + w + x], hi);
    }
}

template <int C>
__global__ void __launch_bounds__(256) gradient_blend_kernel(const uint8_t *__restrict__ warped, int64_t wpitch,
                                                             uint8_t *ref, int64_t rpitch, int h, int w, int feather,
                                                             const int32_t *__restrict__ ext) {
    // tile = 64 pixels x 16 rows; wave wv takes rows wv, wv+4, wv+8, wv+12 (8 row segments in flight)
    const int lane = threadIdx.x & 63;
    const int xs = blockIdx.x * 64, x = xs + lane;
    const int y0 = blockIdx.y * 16 + (threadIdx.x >> 6);
    const int nv = min(64, w - xs);
    uint32_t dw[4], dr[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int y = y0 + 4 * i;
        dw[i] = y < h ? seg_load(warped + y * wpitch + (int64_t)xs * C, nv * C, lane) : 0u;
        dr[i] = y < h ? seg_load(ref + y * rpitch + (int64_t)xs * C, nv * C, lane) : 0u;
    }
    const double f = (double)feather;
    const int xc = x < w ? x : w - 1;
    const int T = ext[2 * h + xc], B = ext[2 * h + w + xc];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int y = y0 + 4 * i;
        if (y >= h) break; // wave-uniform
        const uint32_t pw = seg_pixel<C>(dw[i], lane), pr = seg_pixel<C>(dr[i], lane);
        // the factor chain is the same for every channel of the pixel; start from 1 and apply to mask=1 channels
        double m = 1.0;
        const int L = ext[y], R = ext[h + y];
        if (R >= 0) {
            const int xe = L + feather < w ? L + feather : w;
            if (x >= L && x < xe) m *= (double)(x - L) / f;
            const int xb = R - feather > 0 ? R - feather : 0;
            if (x >= xb && x <= R) m *= (double)(R - x) / f;
        }
        if (B >= 0) {
            const int ye = T + feather < h ? T + feather : h;
            if (y >= T && y < ye) m *= (double)(y - T) / f;
            const int yb = B - feather > 0 ? B - feather : 0;
            if (y >= yb && y <= B) m *= (double)(B - y) / f;
        }
        uint32_t po = 0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int wv = (pw >> (8 * c)) & 255, rv = (pr >> (8 * c)) & 255;
            const double mk = wv > 0 ? m : 0.0;
            const double r = (double)wv * mk + (double)rv * (1 - mk);
            po |= (uint32_t)(uint8_t)(int)r << (8 * c);
        }
        seg_store(ref + y * rpitch + (int64_t)xs * C, nv * C, lane, seg_pack<C>(po, lane));
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
    dim3 gr((w + 1023) / 1024, (h + 3) / 4), gc((w + 255) / 256, (h + 31) / 32), gb((w + 63) / 64, (h + 15) / 16);
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
extern "C" int isx_copy2d(const uint8_t *src_dev, int64_t spitch, uint8_t *dst_dev, int64_t dpitch, int h,
                          int64_t row_bytes, void *stream) {
    ISX_REQUIRE(h >= 0 && row_bytes >= 0, "negative size");
    if (h == 0 || row_bytes == 0) return ISX_OK;
    ISX_REQUIRE(src_dev && dst_dev && spitch >= row_bytes && dpitch >= row_bytes, "bad copy2d arguments");
    ISX_HIP_TRY(hipMemcpy2DAsync(dst_dev, (size_t)dpitch, src_dev, (size_t)spitch, (size_t)row_bytes, (size_t)h,
                                 hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return ISX_OK;
}

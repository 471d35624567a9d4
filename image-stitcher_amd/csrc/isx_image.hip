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

__device__ __forceinline__ void bbox_accumulate(bool nz, int x, int y, int32_t *bb) {
    const unsigned long long any = __ballot(nz);
    if (any == 0ull) return; // wave-uniform
    const int x0 = wave_min(nz ? x : INT_MAX), y0 = wave_min(nz ? y : INT_MAX);
    const int x1 = wave_max(nz ? x + 1 : -1), y1 = wave_max(nz ? y + 1 : -1);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&bb[0], x0);
        atomicMin(&bb[1], y0);
        atomicMax(&bb[2], x1);
        atomicMax(&bb[3], y1);
        atomicOr(&bb[4], 1);
    }
}

// reference utils/proc.py:34-45 get_corners_from_image (cv2.findNonZero + cv2.boundingRect on gray)
template <int C>
__global__ void __launch_bounds__(256) bbox_nonzero_kernel(const uint8_t *__restrict__ img, int h, int w,
                                                           int64_t pitch, int32_t *__restrict__ bb) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y0 = blockIdx.y * 16;
    int minx = INT_MAX, miny = INT_MAX, maxx = -1, maxy = -1;
    if (x < w) {
        for (int dy = 0; dy < 16; dy++) {
            const int y = y0 + dy;
            if (y >= h) break;
            const uint8_t *p = img + y * pitch + (int64_t)x * C;
            const int g = C == 3 ? gray3(p[0], p[1], p[2]) : p[0];
            if (g) {
                minx = x;
                maxx = x + 1;
                miny = miny < y ? miny : y;
                maxy = y + 1;
            }
        }
    }
    const bool nz = maxx >= 0;
    const unsigned long long any = __ballot(nz);
    if (any == 0ull) return;
    const int a = wave_min(minx), b = wave_min(miny), c = wave_max(maxx), d = wave_max(maxy);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&bb[0], a);
        atomicMin(&bb[1], b);
        atomicMax(&bb[2], c);
        atomicMax(&bb[3], d);
        atomicOr(&bb[4], 1);
    }
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
        dim3 grid((w + 255) / 256, (h + 15) / 16);
        if (c == 3)
            hipLaunchKernelGGL((bbox_nonzero_kernel<3>), grid, dim3(256), 0, st, img_dev, h, w, pitch, bbox_out_dev);
        else
            hipLaunchKernelGGL((bbox_nonzero_kernel<1>), grid, dim3(256), 0, st, img_dev, h, w, pitch, bbox_out_dev);
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
// ------------------------------------------------------------------------------------------------
struct WarpParams {
    float m[9];
    int src_x, src_y, sh, sw;
    int dx0, dy0, dw, dh;
    int64_t spitch, dpitch;
};

template <int C, bool PERSP>
__global__ void __launch_bounds__(256) warp_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                   const WarpParams P, int32_t *__restrict__ bb) {
    const int lx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ly = blockIdx.y * 4 + (threadIdx.x >> 6);
    const bool inside = lx < P.dw && ly < P.dh;
    const int x = P.dx0 + lx, y = P.dy0 + ly; // canvas coordinates
    int out[3] = {0, 0, 0};
    if (inside) {
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
        if (sx > (float)P.src_x - 2.f && sx < cw + 1.f && sy > (float)P.src_y - 2.f && sy < ch + 1.f) {
            const float flx = floorf(sx), fly = floorf(sy);
            const int ix = (int)flx - P.src_x, iy = (int)fly - P.src_y; // frame coordinates
            const float a = sx - flx, b = sy - fly;
            const bool x0ok = (unsigned)ix < (unsigned)P.sw, x1ok = (unsigned)(ix + 1) < (unsigned)P.sw;
            const bool y0ok = (unsigned)iy < (unsigned)P.sh, y1ok = (unsigned)(iy + 1) < (unsigned)P.sh;
            const uint8_t *r0 = src + (int64_t)iy * P.spitch + (int64_t)ix * C;
            const uint8_t *r1 = r0 + P.spitch;
#pragma unroll
            for (int c = 0; c < C; c++) {
                const float p00 = (x0ok && y0ok) ? (float)r0[c] : 0.f;
                const float p01 = (x1ok && y0ok) ? (float)r0[C + c] : 0.f;
                const float p10 = (x0ok && y1ok) ? (float)r1[c] : 0.f;
                const float p11 = (x1ok && y1ok) ? (float)r1[C + c] : 0.f;
                const float t0 = fmaf(a, p01 - p00, p00);
                const float t1 = fmaf(a, p11 - p10, p10);
                const float v = fmaf(b, t1 - t0, t0);
                const float r = rintf(v);
                out[c] = r < 0.f ? 0 : (r > 255.f ? 255 : (int)r);
            }
        }
        uint8_t *o = dst + (int64_t)ly * P.dpitch + (int64_t)lx * C;
#pragma unroll
        for (int c = 0; c < C; c++) o[c] = (uint8_t)out[c];
    }
    if (bb) {
        const int g = C == 3 ? gray3(out[0], out[1], out[2]) : out[0];
        bbox_accumulate(inside && g != 0, x, y, bb);
    }
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
    dim3 grid((dw + 63) / 64, (dh + 3) / 4), block(256);
    const bool persp = model == ISX_MODEL_HOMOGRAPHY;
    if (c == 3 && persp) hipLaunchKernelGGL((warp_kernel<3, true>), grid, block, 0, st, src_dev, dst_dev, P, bbox_out_dev);
    else if (c == 3) hipLaunchKernelGGL((warp_kernel<3, false>), grid, block, 0, st, src_dev, dst_dev, P, bbox_out_dev);
    else if (persp) hipLaunchKernelGGL((warp_kernel<1, true>), grid, block, 0, st, src_dev, dst_dev, P, bbox_out_dev);
    else hipLaunchKernelGGL((warp_kernel<1, false>), grid, block, 0, st, src_dev, dst_dev, P, bbox_out_dev);
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
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const uint8_t *p1 = warped + y * wpitch + (int64_t)x * C;
    const uint8_t *p2 = ref + y * rpitch + (int64_t)x * C;
    int a[3], b[3];
#pragma unroll
    for (int c = 0; c < C; c++) {
        a[c] = p1[c];
        b[c] = p2[c];
    }
    const int g1 = C == 3 ? gray3(a[0], a[1], a[2]) : a[0];
    const int g2 = C == 3 ? gray3(b[0], b[1], b[2]) : b[0];
    const bool overlap = g1 > 10 && g2 > 10;
    uint8_t *o = out + y * opitch + (int64_t)x * C;
#pragma unroll
    for (int c = 0; c < C; c++) {
        int va = a[c], vb = b[c];
        if (overlap) {
            vb = (int)(contrib * (double)vb);
            va = (int)(one_minus * (double)va);
        }
        o[c] = (uint8_t)((va + vb) & 255);
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
    dim3 grid((w + 63) / 64, (h + 3) / 4), block(256);
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
    // one wave per (row, 1024-pixel segment)
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (y >= h) return;
    const int xs = blockIdx.x * 1024;
    int lo = INT_MAX, hi = -1;
    const uint8_t *row = warped + y * wpitch;
    for (int i = 0; i < 16; i++) {
        const int x = xs + i * 64 + lane;
        if (x < w && row[(int64_t)x * C] > 0) {
            lo = lo < x ? lo : x;
            hi = x;
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
    // one thread per (column, 32-row segment); threads of a wave walk adjacent columns
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    const int ys = blockIdx.y * 32;
    int lo = INT_MAX, hi = -1;
    for (int i = 0; i < 32; i++) {
        const int y = ys + i;
        if (y >= h) break;
        const bool on = warped[y * wpitch + (int64_t)x * C] > 0 && x != ext[y] && x != ext[h + y];
        if (on) {
            lo = lo < y ? lo : y;
            hi = y;
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
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const uint8_t *p1 = warped + y * wpitch + (int64_t)x * C;
    uint8_t *p2 = ref + y * rpitch + (int64_t)x * C;
    const double f = (double)feather;
    // the factor chain is the same for every channel of the pixel; start from 1 and apply to mask=1 channels
    double m = 1.0;
    const int L = ext[y], R = ext[h + y];
    if (R >= 0) {
        const int xe = L + feather < w ? L + feather : w;
        if (x >= L && x < xe) m *= (double)(x - L) / f;
        const int xs = R - feather > 0 ? R - feather : 0;
        if (x >= xs && x <= R) m *= (double)(R - x) / f;
    }
    const int T = ext[2 * h + x], B = ext[2 * h + w + x];
    if (B >= 0) {
        const int ye = T + feather < h ? T + feather : h;
        if (y >= T && y < ye) m *= (double)(y - T) / f;
        const int ys = B - feather > 0 ? B - feather : 0;
        if (y >= ys && y <= B) m *= (double)(B - y) / f;
    }
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int wv = p1[c];
        const double mk = wv > 0 ? m : 0.0;
        const double r = (double)wv * mk + (double)p2[c] * (1 - mk);
        p2[c] = (uint8_t)(int)r;
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
    dim3 gr((w + 1023) / 1024, (h + 3) / 4), gc((w + 255) / 256, (h + 31) / 32), gb((w + 63) / 64, (h + 3) / 4);
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

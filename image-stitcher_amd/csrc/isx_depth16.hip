// Opt-in depth-preserving mode (BASELINE configs[3]): single-channel uint16 frames
// keep their 16 bits through warp + alpha merge instead of wrapping modulo 256.
//
// The reference allocates every canvas as uint8 (utils/proc.py:283, :285), so a uint16 depth frame is reduced modulo 256
// when it is pasted in (:290-291); that is the default here too (bit parity).  This file is what the same code computes
// when the canvases take the FRAME's dtype: cv2.warpAffine / cv2.warpPerspective on CV_16UC1 (INTER_LINEAR,
// BORDER_CONSTANT 0: the float32 linear kernels of OpenCV 4.11 -- float32 coordinates and lerps, round half to even,
// saturate to 65535 [OCV-K, parity unpinned]) and seamless_merge (utils/proc.py:48-82) on two 2-D uint16 arrays:
// mask = v > 10 (cv2.threshold), overlap = mask1 & mask2, there ref' = (u16)trunc(c * ref), warped' =
// (u16)trunc((1 - c) * warped) in float64, out = warped' + ref' as uint16 (wraps modulo 65536).
// HBM-bound byte work (2 + 2 + 2 bytes per pixel of the warped frame's rectangle): one thread per 2 output pixels,
// coalesced 4-byte stores, gather reads; no LDS.  The gradient blender is not offered in this mode: the reference's
// hard-codes astype(np.uint8) on its result (utils/proc.py:253), which would undo the point of it.
#include "isx_common.h"
#include "isx_image_dev.h"

struct Warp16Params {
    float m[9];
    int src_x, src_y, sh, sw;
    int dx0, dy0, dw, dh;
    int64_t spitch, dpitch; // in ELEMENTS
};

template <bool PERSP>
__device__ __forceinline__ uint32_t warp16_pixel(const uint16_t *__restrict__ src, const Warp16Params &P, int x, int y) {
    const float fx = (float)x, fy = (float)y;
    float sx = fmaf(P.m[0], fx, fmaf(P.m[1], fy, P.m[2]));
    float sy = fmaf(P.m[3], fx, fmaf(P.m[4], fy, P.m[5]));
    if (PERSP) {
        const float w = fmaf(P.m[6], fx, fmaf(P.m[7], fy, P.m[8]));
        sx = sx / w;
        sy = sy / w;
    }
    const float cw = (float)(P.src_x + P.sw), ch = (float)(P.src_y + P.sh); // (the oracle's guard, in canvas space)
    if (!(sx > (float)P.src_x - 2.f && sx < cw + 1.f && sy > (float)P.src_y - 2.f && sy < ch + 1.f)) return 0u;
    const float flx = floorf(sx), fly = floorf(sy);
    const int ix = (int)flx - P.src_x, iy = (int)fly - P.src_y;
    const float a = sx - flx, b = sy - fly;
    const bool x0ok = (unsigned)ix < (unsigned)P.sw, x1ok = (unsigned)(ix + 1) < (unsigned)P.sw;
    const bool y0ok = (unsigned)iy < (unsigned)P.sh, y1ok = (unsigned)(iy + 1) < (unsigned)P.sh;
    const uint16_t *r0 = src + (int64_t)iy * P.spitch + ix;
    const uint16_t *r1 = r0 + P.spitch;
    const float p00 = (x0ok && y0ok) ? (float)r0[0] : 0.f, p01 = (x1ok && y0ok) ? (float)r0[1] : 0.f;
    const float p10 = (x0ok && y1ok) ? (float)r1[0] : 0.f, p11 = (x1ok && y1ok) ? (float)r1[1] : 0.f;
    const float t0 = fmaf(a, p01 - p00, p00);
    const float t1 = fmaf(a, p11 - p10, p10);
    const float v = fmaf(b, t1 - t0, t0);
    const float r = rintf(v);
    return r < 0.f ? 0u : (r > 65535.f ? 65535u : (uint32_t)r);
}

// dst: the rectangle's own buffer (dw x dh, pitch dpitch elements), canvas pixel (dx0 + x, dy0 + y) -> dst[y][x]
template <bool PERSP>
__global__ void __launch_bounds__(256) warp16_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst,
                                                     const Warp16Params P) {
    const int x = 2 * (blockIdx.x * 64 + (threadIdx.x & 63));
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= P.dw || y >= P.dh) return;
    const uint32_t a = warp16_pixel<PERSP>(src, P, P.dx0 + x, P.dy0 + y);
    uint16_t *o = dst + (int64_t)y * P.dpitch + x;
    if (x + 1 < P.dw) {
        const uint32_t b = warp16_pixel<PERSP>(src, P, P.dx0 + x + 1, P.dy0 + y);
        if (((uintptr_t)o & 3) == 0)
            *(uint32_t *)o = a | (b << 16);
        else
            o[0] = (uint16_t)a, o[1] = (uint16_t)b;
    } else {
        o[0] = (uint16_t)a;
    }
}

extern "C" int isx_warp16(const uint16_t *src_dev, int sh, int sw, int64_t spitch_bytes, int src_x, int src_y,
                          const double *M_host, int model, uint16_t *dst_dev, int64_t dpitch_bytes, int dx0, int dy0, int dw,
                          int dh, void *stream) {
    ISX_REQUIRE(model == ISX_MODEL_AFFINE_PARTIAL || model == ISX_MODEL_HOMOGRAPHY, "bad model %d", model);
    ISX_REQUIRE(M_host, "null M_host");
    ISX_REQUIRE(sh > 0 && sw > 0 && spitch_bytes >= 2 * (int64_t)sw && (spitch_bytes & 1) == 0, "bad source geometry");
    ISX_REQUIRE(dw >= 0 && dh >= 0 && dpitch_bytes >= 2 * (int64_t)dw && (dpitch_bytes & 1) == 0, "bad destination geometry");
    if (dw == 0 || dh == 0) return ISX_OK;
    ISX_REQUIRE(src_dev && dst_dev, "null device pointer");
    ISX_REQUIRE((((uintptr_t)src_dev | (uintptr_t)dst_dev) & 1) == 0, "uint16 images must be 2-byte aligned");
    Warp16Params P;
    double iM[9];
    if (model == ISX_MODEL_HOMOGRAPHY)
        invert_3x3_cv(M_host, iM);
    else
        invert_affine_cv(M_host, iM);
    for (int i = 0; i < 9; i++) P.m[i] = (float)iM[i];
    P.src_x = src_x; P.src_y = src_y; P.sh = sh; P.sw = sw;
    P.dx0 = dx0; P.dy0 = dy0; P.dw = dw; P.dh = dh;
    P.spitch = spitch_bytes / 2; P.dpitch = dpitch_bytes / 2;
    dim3 grid((unsigned)((dw + 127) / 128), (unsigned)((dh + 3) / 4)), block(256);
    ISX_REQUIRE(grid.y < 65536u * 16u, "destination rectangle too tall");
    if (model == ISX_MODEL_HOMOGRAPHY)
        hipLaunchKernelGGL((warp16_kernel<true>), grid, block, 0, (hipStream_t)stream, src_dev, dst_dev, P);
    else
        hipLaunchKernelGGL((warp16_kernel<false>), grid, block, 0, (hipStream_t)stream, src_dev, dst_dev, P);
    return isx_launch_status("warp16_kernel");
}

__global__ void __launch_bounds__(256) alpha_blend16_kernel(const uint16_t *__restrict__ warped, int64_t wpitch,
                                                            const uint16_t *ref, int64_t rpitch, uint16_t *out, int64_t opitch,
                                                            int h, int w, double contrib, double one_minus) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    uint32_t a = warped[(int64_t)y * wpitch + x], b = ref[(int64_t)y * rpitch + x];
    if (a > 10u && b > 10u) {
        b = (uint32_t)(contrib * (double)b);
        a = (uint32_t)(one_minus * (double)a);
    }
    out[(int64_t)y * opitch + x] = (uint16_t)(a + b);
}

extern "C" int isx_alpha_blend16(const uint16_t *warped_dev, int64_t wpitch_bytes, const uint16_t *ref_dev, int64_t rpitch_bytes,
                                 uint16_t *out_dev, int64_t opitch_bytes, int h, int w, double ref_contrib, void *stream) {
    ISX_REQUIRE(h >= 0 && w >= 0, "negative size");
    if (h == 0 || w == 0) return ISX_OK;
    ISX_REQUIRE(warped_dev && ref_dev && out_dev, "null device pointer");
    ISX_REQUIRE(wpitch_bytes >= 2 * (int64_t)w && rpitch_bytes >= 2 * (int64_t)w && opitch_bytes >= 2 * (int64_t)w &&
                    ((wpitch_bytes | rpitch_bytes | opitch_bytes) & 1) == 0, "bad pitch");
    dim3 grid((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4)), block(256);
    hipLaunchKernelGGL(alpha_blend16_kernel, grid, block, 0, (hipStream_t)stream, warped_dev, wpitch_bytes / 2, ref_dev,
                       rpitch_bytes / 2, out_dev, opitch_bytes / 2, h, w, ref_contrib, 1.0 - ref_contrib);
    return isx_launch_status("alpha_blend16_kernel");
}

// Device-side building blocks shared by the image kernels (isx_image.hip) and the per-frame pipeline
// (isx_frame.hip): BGR->gray, unaligned pixel-quad loads/stores, the two INTER_LINEAR warp arithmetics, the
// alpha-blend pixel, and cv's host-side matrix inversions.  gfx950 only.
#pragma once
#include "isx_common.h"
#include <limits.h>

// [OCV-K] cv2.cvtColor(BGR2GRAY) 8-bit; reference utils/proc.py:36, :51-52
__device__ __forceinline__ int gray3(int b, int g, int r) { return (b * 3735 + g * 19235 + r * 9798 + 16384) >> 15; }
template <int C>
__device__ __forceinline__ int gray_px(uint32_t px) {
    return C == 3 ? gray3(px & 255, (px >> 8) & 255, (px >> 16) & 255) : (int)(px & 255);
}

// Wave-wide min / max, every lane gets the result.  Four DPP steps give every lane its 16-lane row's value (quad
// permutes, then half-row and row mirrors: after two steps all lanes of a quad agree, so any lane of the partner group
// will do), the four rows are combined from v_readlane: no LDS-crossbar shuffles (six dependent ones per reduction
// before).  Every lane of the wave must be active (all call sites are in uniform control flow of full waves); `old` = v,
// so a lane never picks up anything but a lane's own value.
__device__ __forceinline__ int wave_min(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// ------------------------------------------------------------------------------------------------
// Pixel quads.  A lane owns 4 horizontally adjacent pixels = 4*C contiguous bytes and moves them with one
// unaligned 12-byte (C=3) / 4-byte (C=1) access (crops start at arbitrary x, so no alignment can be
// assumed; gfx950 global accesses are byte-addressable in the default unaligned mode).  A wave therefore
// covers 256 pixels of a row (768 B per instruction for BGR) with no cross-lane traffic.
// ------------------------------------------------------------------------------------------------
typedef uint16_t u16u __attribute__((aligned(1)));
typedef uint32_t u32u __attribute__((aligned(1)));
typedef uint64_t u64u __attribute__((aligned(1)));
struct __attribute__((packed, aligned(1))) u96u {
    uint32_t a, b, c;
};

// pixels x .. x+3 of a row (packed B | G<<8 | R<<16, or the gray byte); pixels at or beyond w read as 0
template <int C>
__device__ __forceinline__ void quad_load(const uint8_t *__restrict__ row, int x, int w, uint32_t (&px)[4]) {
    const uint8_t *p = row + (int64_t)x * C;
    if (x + 4 <= w) {
        if (C == 3) {
            const u96u v = *(const u96u *)p;
            px[0] = v.a & 0xFFFFFFu;
            px[1] = (v.a >> 24) | ((v.b & 0xFFFFu) << 8);
            px[2] = (v.b >> 16) | ((v.c & 0xFFu) << 16);
            px[3] = v.c >> 8;
        } else {
            const uint32_t v = *(const u32u *)p;
            px[0] = v & 255u, px[1] = (v >> 8) & 255u, px[2] = (v >> 16) & 255u, px[3] = v >> 24;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            px[i] = 0u;
            if (x + i < w)
                for (int c = 0; c < C; c++) px[i] |= (uint32_t)p[i * C + c] << (8 * c);
        }
    }
}
template <int C>
__device__ __forceinline__ void quad_store(uint8_t *__restrict__ row, int x, int w, const uint32_t (&px)[4]) {
    uint8_t *p = row + (int64_t)x * C;
    if (x + 4 <= w) {
        if (C == 3) {
            u96u v;
            v.a = px[0] | (px[1] << 24);
            v.b = (px[1] >> 8) | (px[2] << 16);
            v.c = (px[2] >> 16) | (px[3] << 8);
            *(u96u *)p = v;
        } else {
            *(u32u *)p = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (x + i < w)
                for (int c = 0; c < C; c++) p[i * C + c] = (uint8_t)(px[i] >> (8 * c));
    }
}

// The same quad kept PACKED, as the 3 (C=3) or 1 (C=1) dwords that hold its 4*C bytes: byte-parallel kernels work on
// these directly and only unpack to pixels where per-pixel arithmetic is needed.
template <int C>
struct QuadRaw {
    static constexpr int N = C == 3 ? 3 : 1;
};
template <int C>
__device__ __forceinline__ void raw_load(const uint8_t *__restrict__ row, int x, int w, uint32_t (&d)[QuadRaw<C>::N]) {
    const uint8_t *p = row + (int64_t)x * C;
    if (x + 4 <= w) {
        if (C == 3) {
            const u96u v = *(const u96u *)p;
            d[0] = v.a, d[1] = v.b, d[2] = v.c;
        } else {
            d[0] = *(const u32u *)p;
        }
    } else {
#pragma unroll
        for (int j = 0; j < QuadRaw<C>::N; j++) d[j] = 0u;
#pragma unroll
        for (int b = 0; b < 4 * C; b++)
            if (x + b / C < w) d[b >> 2] |= (uint32_t)p[b] << (8 * (b & 3));
    }
}
template <int C>
__device__ __forceinline__ void raw_store(uint8_t *__restrict__ row, int x, int w, const uint32_t (&d)[QuadRaw<C>::N]) {
    uint8_t *p = row + (int64_t)x * C;
    if (x + 4 <= w) {
        if (C == 3) {
            u96u v;
            v.a = d[0], v.b = d[1], v.c = d[2];
            *(u96u *)p = v;
        } else {
            *(u32u *)p = d[0];
        }
    } else {
#pragma unroll
        for (int b = 0; b < 4 * C; b++)
            if (x + b / C < w) p[b] = (uint8_t)(d[b >> 2] >> (8 * (b & 3)));
    }
}
template <int C>
__device__ __forceinline__ void raw_unpack(const uint32_t (&d)[QuadRaw<C>::N], uint32_t (&px)[4]) {
    if (C == 3) {
        px[0] = d[0] & 0xFFFFFFu;
        px[1] = (d[0] >> 24) | ((d[1] & 0xFFFFu) << 8);
        px[2] = (d[1] >> 16) | ((d[2] & 0xFFu) << 16);
        px[3] = d[2] >> 8;
    } else {
        px[0] = d[0] & 255u, px[1] = (d[0] >> 8) & 255u, px[2] = (d[0] >> 16) & 255u, px[3] = d[0] >> 24;
    }
}
template <int C>
__device__ __forceinline__ void raw_pack(const uint32_t (&px)[4], uint32_t (&d)[QuadRaw<C>::N]) {
    if (C == 3) {
        d[0] = px[0] | (px[1] << 24);
        d[1] = (px[1] >> 8) | (px[2] << 16);
        d[2] = (px[2] >> 16) | (px[3] << 8);
    } else {
        d[0] = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
    }
}
// 0xFF in every byte of w that is != 0
__device__ __forceinline__ uint32_t nz_bytes(uint32_t w) {
    const uint32_t a = (((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w) & 0x80808080u;
    return a | (a - (a >> 7));
}
// sufficient (not necessary) test that all four pixels of a packed quad have gray != 0: the G byte of a BGR pixel
// alone carries 19235 >= 16384 (gray3), for C = 1 the byte is the gray value
template <int C>
__device__ __forceinline__ bool raw_all_gray_nz(const uint32_t (&d)[QuadRaw<C>::N]) {
    if (C == 3) return (d[0] & 0xFF00u) != 0u && (d[1] & 0xFFu) != 0u && d[1] > 0xFFFFFFu && (d[2] & 0xFF0000u) != 0u;
    return nz_bytes(d[0]) == 0xFFFFFFFFu;
}

template <int C>
__device__ __forceinline__ void bbox_quad(const uint32_t (&px)[4], int x, int y, int &minx, int &miny, int &maxx,
                                          int &maxy) {
    // pixels outside the image were loaded / produced as 0 and never count
    int first = -1, last = -1;
#pragma unroll
    for (int i = 3; i >= 0; i--)
        if (gray_px<C>(px[i]) != 0) first = i;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (gray_px<C>(px[i]) != 0) last = i;
    if (last >= 0) {
        minx = min(minx, x + first);
        maxx = max(maxx, x + last + 1);
        miny = min(miny, y);
        maxy = max(maxy, y + 1);
    }
}

// ------------------------------------------------------------------------------------------------
// warp.  reference stitcher.py:143-149: cv2.warpAffine / cv2.warpPerspective(img, M, (W, H)) with the
// defaults INTER_LINEAR, BORDER_CONSTANT 0; M is the forward map.
// ------------------------------------------------------------------------------------------------
struct WarpParams {
    float m[9];
    double im[9]; // the inverse map in double (classic arithmetic)
    int bw0;      // classic perspective: width of the invoker's column blocks on the full canvas
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

// The classic arithmetic of OpenCV <= 4.10 (imgwarp.cpp WarpAffineInvoker / WarpPerspectiveInvoker +
// remapBilinear with FixedPtCast<int, uchar, 15>), restated in oracle/orc_image.c warp_classic(): source
// coordinates in 1/32 px fixed point from the double inverse map, int16 bilinear weights that sum to 32768
// (entry (0,0) = {32767, 0, 0, 1}), value = (sum + 16384) >> 15.  Selected with ISX_WARP_CLASSIC.
template <int C, bool PERSP>
__device__ __forceinline__ uint32_t warp_pixel_classic(const uint8_t *__restrict__ src, const WarpParams &P, int x, int y) {
    int X, Y;
    if (!PERSP) {
        const int X0 = __double2int_rn((P.im[1] * (double)y + P.im[2]) * 1024.0) + 16;
        const int Y0 = __double2int_rn((P.im[4] * (double)y + P.im[5]) * 1024.0) + 16;
        X = (X0 + __double2int_rn(P.im[0] * (double)x * 1024.0)) >> 5;
        Y = (Y0 + __double2int_rn(P.im[3] * (double)x * 1024.0)) >> 5;
    } else {
        const int xb = x - x % P.bw0, x1 = x - xb;
        const double X0 = P.im[0] * (double)xb + P.im[1] * (double)y + P.im[2];
        const double Y0 = P.im[3] * (double)xb + P.im[4] * (double)y + P.im[5];
        const double W0 = P.im[6] * (double)xb + P.im[7] * (double)y + P.im[8];
        double W = W0 + P.im[6] * (double)x1;
        W = W != 0.0 ? 32.0 / W : 0.0;
        double fX = (X0 + P.im[0] * (double)x1) * W, fY = (Y0 + P.im[3] * (double)x1) * W;
        // std::max(INT_MIN, std::min(INT_MAX, v)): a NaN ends at INT_MAX
        fX = !(fX < 2147483647.0) ? 2147483647.0 : fX;
        fX = fX < -2147483648.0 ? -2147483648.0 : fX;
        fY = !(fY < 2147483647.0) ? 2147483647.0 : fY;
        fY = fY < -2147483648.0 ? -2147483648.0 : fY;
        X = __double2int_rn(fX);
        Y = __double2int_rn(fY);
    }
    int ix = X >> 5, iy = Y >> 5;
    ix = ix < -32768 ? -32768 : (ix > 32767 ? 32767 : ix);
    iy = iy < -32768 ? -32768 : (iy > 32767 ? 32767 : iy);
    const int fx = X & 31, fy = Y & 31;
    int w00 = (32 - fy) * (32 - fx) * 32, w11 = fy * fx * 32;
    const int w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32;
    if ((fx | fy) == 0) {
        w00 = 32767;
        w11 = 1;
    }
    ix -= P.src_x; // frame coordinates: taps outside the frame are the zero padding / border
    iy -= P.src_y;
    const bool x0ok = (unsigned)ix < (unsigned)P.sw, x1ok = (unsigned)(ix + 1) < (unsigned)P.sw;
    const bool y0ok = (unsigned)iy < (unsigned)P.sh, y1ok = (unsigned)(iy + 1) < (unsigned)P.sh;
    if (!((x0ok || x1ok) && (y0ok || y1ok))) return 0u;
    const uint8_t *r0 = src + (int64_t)iy * P.spitch + (int64_t)ix * C;
    const uint8_t *r1 = r0 + P.spitch;
    uint32_t px = 0;
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int p00 = (x0ok && y0ok) ? r0[c] : 0, p01 = (x1ok && y0ok) ? r0[C + c] : 0;
        const int p10 = (x0ok && y1ok) ? r1[c] : 0, p11 = (x1ok && y1ok) ? r1[C + c] : 0;
        const int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + 16384) >> 15;
        px |= (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)) << (8 * c);
    }
    return px;
}

// host-side inverse exactly as cv::warpAffine / cv::warpPerspective compute it (double)
static inline void invert_affine_cv(const double *M, double *iM) {
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
static inline void invert_3x3_cv(const double *m, double *o) {
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

// one pixel of seamless_merge: warped pixel a, reference pixel b (packed B | G<<8 | R<<16 or the gray byte)
template <int C>
__device__ __forceinline__ uint32_t alpha_px(uint32_t a, uint32_t b, double contrib, double one_minus) {
    const bool overlap = gray_px<C>(a) > 10 && gray_px<C>(b) > 10;
    uint32_t o = 0;
#pragma unroll
    for (int c = 0; c < C; c++) {
        int va = (a >> (8 * c)) & 255, vb = (b >> (8 * c)) & 255;
        if (overlap) {
            vb = (int)(contrib * (double)vb);
            va = (int)(one_minus * (double)va);
        }
        o |= (uint32_t)((va + vb) & 255) << (8 * c);
    }
    return o;
}


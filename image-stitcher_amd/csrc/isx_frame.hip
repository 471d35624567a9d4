// Per-frame pipeline on a resident mosaic: Stitcher.warp_target + Stitcher.merge + the two bounding boxes of
// Stitcher.stitch (reference stitcher.py:216-225 -> utils/proc.py:48-82, :192-254, :34-45, :307-312) as ONE
// entry point that runs 3 (gradient blender) or 1 (alpha blender) launches and returns every per-frame result
// in one small read-back.
//
// The reference builds two padded canvases per call and scans the whole canvas three times.  Here the mosaic
// lives in a caller-owned arena that is zero outside the mosaic; the frame is warped only over Q, a
// conservative canvas rectangle around the warped quad, and the merged canvas is never materialised: its
// bounding box is  bbox(old mosaic) U bbox(pixels written)  as long as no pixel went from gray != 0 to
// gray == 0 (counted on the device; the caller rescans in that rare case).
//
//   gradient blender (utils/proc.py:192-254)
//     K1 frame_warp_kernel      one workgroup per canvas row of Q: warped pixels -> staging image, the row's
//                               gray != 0 extent (-> bbox(warped), stitcher.py:220 / proc.py:194) and channel-0
//                               extent (-> left/right feather edges, proc.py:213-218), a 1-bit/pixel mask of
//                               channel 0 != 0 with the row's two edge bits cleared (the feather factor there
//                               is 0); plain stores only, nothing to initialise
//     K2 frame_extent_kernel    one workgroup per 256-column chunk: folds the row table into bbox(warped) and
//                               the ROI (bbox +- 50 px clipped to the canvas, proc.py:199-201); column extents of
//                               the horizontally feathered mask (proc.py:230-236) through a three-level OR
//                               hierarchy of the bit mask in LDS -- one thread per column and direction, so no
//                               atomics and nothing to initialise
//     K3 frame_gradient_kernel  feathered blend in place on the mosaic inside the ROI (proc.py:211-253): a
//                               byte-parallel select outside the feather bands, the float64 factor chain inside;
//                               per-workgroup partials of the written pixels' bbox and the zeroed / dirty flags
//   Row workgroups (K1, alpha) and K3's tiles are dealt to the 8 XCDs in BANDS (frame_row_of_block): each L2 then
//   fetches its part of the frame / staging image once.
//   alpha blender (utils/proc.py:48-82)
//     K1 frame_warp_alpha_kernel  warp + blend in place, per-row table of both bounding boxes and the flags
//   The small tables (K3's partials / the alpha row table) come back with the result block in ONE device-to-
//   host copy and are folded on the host: a fold kernel would cost a dependent launch (4-8 us measured) for a
//   few microseconds of PCIe.  No same-address atomics anywhere: device-scope atomics from different XCDs are
//   resolved at the memory side, ~30 ns each, serialised per address (the first version spent 16 + 50 us there).
#include "isx_common.h"
#include "isx_image_dev.h"
#include "isx_ctx.h"
#include <string.h>

// result block (int32)
enum {
    R_WBBOX = 0,    // 5: bbox of gray != 0 warped pixels, canvas coordinates {x0,y0,x1,y1,found}
    R_ROI = 5,      // 4: gradient ROI {gx0,gy0,gx1,gy1} (canvas coordinates)
    R_MBBOX = 9,    // 5: bbox of gray != 0 pixels among the pixels written into the mosaic
    R_ZEROED = 14,  // != 0: some pixel's gray was != 0 before the blend and is 0 after it
    R_DIRTY = 15,   // != 0: some pixel was written with gray == 0 but a byte != 0
    R_ROWS_OK = 16, // gradient: every channel-0 != 0 pixel of the ROI's rows lies inside the ROI's columns
    R_DONE = 17,    // the blend ran (0: nothing warped, or rows_ok == 0 -> the caller takes the slow path)
    R_INTS = ISX_FRAME_RESULT_INTS
};
#define FRAME_MAX_WAVES 16
#define FE_BATCH 1536 // rows of bit mask staged in LDS at a time (48 KiB)

extern "C" int isx_ctx_create(isx_ctx **out) {
    ISX_REQUIRE(out, "null out");
    isx_ctx *c = new isx_ctx();
    memset(c, 0, sizeof(*c));
    if (hipGetDevice(&c->device) != hipSuccess) {
        const hipError_t e = hipGetLastError();
        delete c;
        return isx_fail(ISX_E_HIP, "isx_ctx_create: %s", hipGetErrorString(e));
    }
    const int rc = ctx_ensure_out(c, 4096);
    if (rc != ISX_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return ISX_OK;
}

extern "C" int isx_ctx_destroy(isx_ctx *c) {
    if (!c) return ISX_OK;
    if (c->wbuf) (void)hipFree(c->wbuf);
    if (c->rows) (void)hipFree(c->rows);
    if (c->cols) (void)hipFree(c->cols);
    if (c->mask) (void)hipFree(c->mask);
    if (c->out_dev) (void)hipFree(c->out_dev);
    if (c->out_host) (void)hipHostFree(c->out_host);
    if (c->hs_dev) (void)hipFree(c->hs_dev);
    if (c->hs_host) (void)hipHostFree(c->hs_host);
    delete c;
    return ISX_OK;
}

// ------------------------------------------------------------------------------------------------
// Four horizontally adjacent warped pixels (float32 arithmetic, affine or perspective).  When EVERY pixel of the
// wave takes warp_pixel's wide-load path (both taps of both rows inside the frame) the per-tap validity logic,
// the guard and the slow path are skipped wave-uniformly: the same loads and the same FMAs, about a third of
// the vector instructions -- this kernel is VALU-bound (150 instructions per pixel in the general form).
// ------------------------------------------------------------------------------------------------
// one tap of the general path: pixel (cx, cy) of the frame, coordinates already clamped into the frame; the
// 4-byte load is pulled back by one byte at the end of a row so that it never leaves the row (needs sw*C >= 4)
template <int C>
__device__ __forceinline__ uint32_t tap_clamped(const uint8_t *__restrict__ src, const WarpParams &P, int cx, int cy) {
    if (C == 3) {
        const int want = cx * 3, off = min(want, P.sw * 3 - 4);
        const uint32_t v = *(const u32u *)(src + ((uint32_t)cy * (uint32_t)P.spitch + (uint32_t)off));
        return (v >> (8 * (want - off))) & 0xFFFFFFu;
    }
    return src[(uint32_t)cy * (uint32_t)P.spitch + (uint32_t)cx];
}

template <int C, bool PERSP>
__device__ __forceinline__ void warp_quad_f32(const uint8_t *__restrict__ src, const WarpParams &P, int X0, int Y, int n_valid,
                                              uint32_t (&px)[4]) {
    const float fy = (float)Y;
    const float tx = fmaf(P.m[1], fy, P.m[2]), ty = fmaf(P.m[4], fy, P.m[5]), tw = fmaf(P.m[7], fy, P.m[8]);
    const float x_lo = (float)P.src_x, x_hi = (float)(P.src_x + P.sw), y_lo = (float)P.src_y, y_hi = (float)(P.src_y + P.sh);
    // "inside": both taps of both rows in the frame and the wide load inside the row (warp_pixel's `wide`):
    //   0 <= floor(sx) - src_x <= sw - 3 (C = 3; sw - 2 for C = 1)  and  0 <= floor(sy) - src_y <= sh - 2
    const float x_in = x_hi - (C == 3 ? 2.f : 1.f), y_in = y_hi - 1.f;
    float sx[4], sy[4];
    bool inside, outside;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float fx = (float)(X0 + i);
        sx[i] = fmaf(P.m[0], fx, tx);
        sy[i] = fmaf(P.m[3], fx, ty);
        if (PERSP) {
            const float w = fmaf(P.m[6], fx, tw);
            sx[i] = sx[i] / w;
            sy[i] = sy[i] / w;
        }
    }
    if (!PERSP) {
        // affine: fmaf(m, fx, t) is monotone in fx, so the four pixels of the lane lie between its end pixels and
        // two of them decide for all four
        inside = n_valid == 4 && sx[0] >= x_lo && sx[0] < x_in && sx[3] >= x_lo && sx[3] < x_in && sy[0] >= y_lo && sy[0] < y_in &&
                 sy[3] >= y_lo && sy[3] < y_in;
        outside = (sx[0] <= x_lo - 2.f && sx[3] <= x_lo - 2.f) || (sx[0] >= x_hi + 1.f && sx[3] >= x_hi + 1.f) ||
                  (sy[0] <= y_lo - 2.f && sy[3] <= y_lo - 2.f) || (sy[0] >= y_hi + 1.f && sy[3] >= y_hi + 1.f) || n_valid == 0;
    } else {
        inside = n_valid == 4;
        outside = true;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            inside = inside && sx[i] >= x_lo && sx[i] < x_in && sy[i] >= y_lo && sy[i] < y_in;
            outside = outside && !(i < n_valid && sx[i] > x_lo - 2.f && sx[i] < x_hi + 1.f && sy[i] > y_lo - 2.f && sy[i] < y_hi + 1.f);
        }
    }
    if (__all(outside)) { // margins of Q: nothing to fetch
#pragma unroll
        for (int i = 0; i < 4; i++) px[i] = 0u;
        return;
    }
#ifdef ISX_DIAG_K1_NOGENERAL // A/B only: border waves produce zeros (WRONG results) to see what the general path costs
    if (!__all(inside)) {
#pragma unroll
        for (int i = 0; i < 4; i++) px[i] = 0u;
        return;
    }
#endif
    if (__all(inside)) {
        const uint32_t sp = (uint32_t)P.spitch; // the frame is smaller than 2 GiB (checked by the host): 32-bit offsets
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float flx = floorf(sx[i]), fly = floorf(sy[i]);
            const float a = sx[i] - flx, b = sy[i] - fly;
            const uint32_t off = (uint32_t)((int)fly - P.src_y) * sp + (uint32_t)((int)flx - P.src_x) * C;
            const uint8_t *r0 = src + off;
            uint32_t o = 0;
            if (C == 3) {
#ifdef ISX_DIAG_K1_NOLOAD
                const uint64_t q0 = off * 0x0101010101ull, q1 = q0 + sp;
#else
                const uint64_t q0 = *(const u64u *)r0, q1 = *(const u64u *)(r0 + sp);
#endif
                const uint32_t q0l = (uint32_t)q0, q0h = (uint32_t)(q0 >> 32), q1l = (uint32_t)q1, q1h = (uint32_t)(q1 >> 32);
                // taps: bytes 0..2 = left pixel, bytes 3..5 = right pixel
                const float f00[3] = {(float)(q0l & 255u), (float)((q0l >> 8) & 255u), (float)((q0l >> 16) & 255u)};
                const float f01[3] = {(float)(q0l >> 24), (float)(q0h & 255u), (float)((q0h >> 8) & 255u)};
                const float f10[3] = {(float)(q1l & 255u), (float)((q1l >> 8) & 255u), (float)((q1l >> 16) & 255u)};
                const float f11[3] = {(float)(q1l >> 24), (float)(q1h & 255u), (float)((q1h >> 8) & 255u)};
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float t0 = fmaf(a, f01[c] - f00[c], f00[c]), t1 = fmaf(a, f11[c] - f10[c], f10[c]);
                    // v_cvt_pk_u8_f32 itself rounds to nearest, ties to even, and saturates (measured: 0.5 -> 0, 1.5 -> 2,
                    // 2.5 -> 2, 254.5 -> 254, 300 -> 255, -1 -> 0; tools/valu_rate.hip) -- rintf + the cast of warp_pixel
                    // in one 4-cycle instruction (the kernel runs in the default round-to-nearest-even mode)
                    o = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(b, t1 - t0, t0), c, o);
                }
            } else {
                const uint32_t q0 = *(const u16u *)r0, q1 = *(const u16u *)(r0 + sp);
                const float f00 = (float)(q0 & 255u), f01 = (float)(q0 >> 8), f10 = (float)(q1 & 255u), f11 = (float)(q1 >> 8);
                const float t0 = fmaf(a, f01 - f00, f00), t1 = fmaf(a, f11 - f10, f10);
                o = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(b, t1 - t0, t0), 0, 0u);
            }
            px[i] = o;
        }
        return;
    }
    bool guard[4];
    float flx[4], fly[4];
    int ix[4], iy[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // warp_pixel's guard (a NaN fails it); (int)floorf() is only formed from values inside it
        guard[i] = i < n_valid && sx[i] > x_lo - 2.f && sx[i] < x_hi + 1.f && sy[i] > y_lo - 2.f && sy[i] < y_hi + 1.f;
        flx[i] = floorf(guard[i] ? sx[i] : x_lo);
        fly[i] = floorf(guard[i] ? sy[i] : y_lo);
        ix[i] = (int)flx[i] - P.src_x;
        iy[i] = (int)fly[i] - P.src_y;
    }
    // the frame's border: taps outside the frame are the constant-0 border.  BRANCH-FREE: every tap is fetched from
    // clamped coordinates and masked afterwards, so the 16 loads of a lane are in flight together (the per-tap
    // branches of warp_pixel serialise up to 12 memory round trips per pixel, and one such wave sets the kernel time)
    uint32_t t00[4], t01[4], t10[4], t11[4];
    if (C == 3 && P.sw * 3 >= 8) {
        // BGR: both taps of a row come out of ONE 8-byte load whose start is clamped into the row; a byte permute picks
        // the three bytes of each tap (or zeros for a tap outside the frame): 8 loads per lane instead of 16 and no
        // per-tap masking afterwards
        const uint32_t sp = (uint32_t)P.spitch;
        uint64_t q0[4], q1[4];
        int offx[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int cy0 = min(max(iy[i], 0), P.sh - 1), cy1 = min(max(iy[i] + 1, 0), P.sh - 1);
            offx[i] = min(max(ix[i] * 3, 0), P.sw * 3 - 8);
            q0[i] = *(const u64u *)(src + ((uint32_t)cy0 * sp + (uint32_t)offx[i]));
            q1[i] = *(const u64u *)(src + ((uint32_t)cy1 * sp + (uint32_t)offx[i]));
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool x0ok = (unsigned)ix[i] < (unsigned)P.sw, x1ok = (unsigned)(ix[i] + 1) < (unsigned)P.sw;
            const bool y0ok = guard[i] && (unsigned)iy[i] < (unsigned)P.sh, y1ok = guard[i] && (unsigned)(iy[i] + 1) < (unsigned)P.sh;
            // selector bytes {k, k+1, k+2, 0x0C = constant 0} with k = the tap's byte offset inside the 8 loaded bytes
            // (0..5 whenever the tap is inside the frame)
            const uint32_t kl = (uint32_t)(ix[i] * 3 - offx[i]), kr = kl + 3u;
            const uint32_t sl = x0ok ? 0x0C020100u + kl * 0x00010101u : 0x0C0C0C0Cu;
            const uint32_t sr = x1ok ? 0x0C020100u + kr * 0x00010101u : 0x0C0C0C0Cu;
            const uint32_t l0 = (uint32_t)q0[i], h0 = (uint32_t)(q0[i] >> 32), l1 = (uint32_t)q1[i], h1 = (uint32_t)(q1[i] >> 32);
            t00[i] = y0ok ? __builtin_amdgcn_perm(h0, l0, sl) : 0u;
            t01[i] = y0ok ? __builtin_amdgcn_perm(h0, l0, sr) : 0u;
            t10[i] = y1ok ? __builtin_amdgcn_perm(h1, l1, sl) : 0u;
            t11[i] = y1ok ? __builtin_amdgcn_perm(h1, l1, sr) : 0u;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int cx0 = min(max(ix[i], 0), P.sw - 1), cx1 = min(max(ix[i] + 1, 0), P.sw - 1);
            const int cy0 = min(max(iy[i], 0), P.sh - 1), cy1 = min(max(iy[i] + 1, 0), P.sh - 1);
            const bool x0ok = (unsigned)ix[i] < (unsigned)P.sw, x1ok = (unsigned)(ix[i] + 1) < (unsigned)P.sw;
            const bool y0ok = (unsigned)iy[i] < (unsigned)P.sh, y1ok = (unsigned)(iy[i] + 1) < (unsigned)P.sh;
            const uint32_t a00 = tap_clamped<C>(src, P, cx0, cy0), a01 = tap_clamped<C>(src, P, cx1, cy0);
            const uint32_t a10 = tap_clamped<C>(src, P, cx0, cy1), a11 = tap_clamped<C>(src, P, cx1, cy1);
            t00[i] = (guard[i] && x0ok && y0ok) ? a00 : 0u, t01[i] = (guard[i] && x1ok && y0ok) ? a01 : 0u;
            t10[i] = (guard[i] && x0ok && y1ok) ? a10 : 0u, t11[i] = (guard[i] && x1ok && y1ok) ? a11 : 0u;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t p00 = t00[i], p01 = t01[i], p10 = t10[i], p11 = t11[i];
        const float a = sx[i] - flx[i], b = sy[i] - fly[i];
        uint32_t o = 0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float f00 = (float)((p00 >> (8 * c)) & 255u), f01 = (float)((p01 >> (8 * c)) & 255u);
            const float f10 = (float)((p10 >> (8 * c)) & 255u), f11 = (float)((p11 >> (8 * c)) & 255u);
            const float t0 = fmaf(a, f01 - f00, f00), t1 = fmaf(a, f11 - f10, f10);
            o = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(b, t1 - t0, t0), c, o);
        }
        px[i] = guard[i] ? o : 0u;
    }
}

// Row of Q a row workgroup takes.  Workgroup ids go round the 8 XCDs (id % 8); with row = id every XCD's L2 would fetch
// every source row (a canvas row needs source rows y and y + 1): 2.8 x the frame in memory-side reads, measured with
// FETCH_SIZE.  Instead XCD x takes the band of rows [x * rpx, (x + 1) * rpx), rpx = ceil(qh / 8): each L2 fetches its
// band of the frame once.  The grid is 8 * rpx workgroups; the few beyond qh exit.
__device__ __forceinline__ int frame_row_of_block(int qh) {
    const int rpx = (qh + 7) >> 3;
    return (int)(blockIdx.x & 7) * rpx + (int)(blockIdx.x >> 3);
}

// Ballot of "gray != 0" over the wave.  BGR: gray != 0 whenever G != 0 (19235 >= 16384, gray3); the exact test only runs
// if some pixel of the wave is non-zero with G == 0.
template <int C>
__device__ __forceinline__ uint64_t ballot_gray_nz(uint32_t p) {
    if (C == 3) {
        const uint64_t g = __ballot((p & 0xFF00u) != 0u), nz = __ballot(p != 0u);
        return (nz & ~g) ? __ballot(gray_px<C>(p) != 0) : g;
    }
    return __ballot((p & 255u) != 0u);
}

template <int C, bool PERSP, bool CLASSIC>
__device__ __forceinline__ void warp_quad(const uint8_t *__restrict__ src, const WarpParams &P, int X0, int Y, int n_valid,
                                          uint32_t (&px)[4]) {
    if (CLASSIC) {
#pragma unroll
        for (int i = 0; i < 4; i++) px[i] = i < n_valid ? warp_pixel_classic<C, PERSP>(src, P, X0 + i, Y) : 0u;
    } else {
        warp_quad_f32<C, PERSP>(src, P, X0, Y, n_valid, px);
    }
}

// ------------------------------------------------------------------------------------------------
// K1 (gradient path).  grid = qh workgroups of nw = min(nchunk, 4) waves; wave w takes the 256-pixel chunks
// w, w+nw, ... of its row (rotated by the row, see below).
// ------------------------------------------------------------------------------------------------
#ifndef ISX_K1_WAVES
#define ISX_K1_WAVES 6
#endif
template <int C, bool PERSP, bool CLASSIC>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(ISX_K1_WAVES))) frame_warp_kernel(const uint8_t *__restrict__ src, const WarpParams P,
                                                          uint8_t *__restrict__ wbuf, int nchunk,
                                                          int32_t *__restrict__ rows, uint64_t *__restrict__ mask) {
    // (readfirstlane: the wave index is uniform, which the compiler cannot see -- the chunk origin and the extent
    // bookkeeping below then stay on the scalar unit)
    const int ly = frame_row_of_block(P.dh);
    if (ly >= P.dh) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    int gl = INT_MAX, gr = -1, l0 = INT_MAX, r0 = -1;
    __shared__ uint64_t mw[(ISX_FRAME_MAX_Q / 256) * 4];
    for (int cw = wave; cw < nchunk; cw += nw) {
        // the chunk a wave takes rotates with the row: the frame's left and right borders (the expensive general path of
        // warp_quad) sit in the same two chunks of every row, and a fixed wave -> chunk map would give all of them to the
        // same two SIMDs of every CU (wave w runs on SIMD w % 4)
        const int chunk = (cw + ly) % nchunk;
        const int lx = chunk * 256 + 4 * lane;
        const int nv = min(4, max(0, P.dw - lx));
        uint32_t px[4];
        warp_quad<C, PERSP, CLASSIC>(src, P, P.dx0 + lx, P.dy0 + ly, nv, px);
#if defined(ISX_K3_REWARP)
        // (A/B build, profiles/r03_image_kernels.md: K3 recomputes the warp of its own tile, nothing is staged)
#elif !defined(ISX_DIAG_K1_NOSTORE)
        if (nv > 0) quad_store<C>(wbuf + (int64_t)ly * P.dpitch, lx, P.dw, px);
#else
        if (px[0] + px[1] + px[2] + px[3] == 0x7FFFFFFFu) wbuf[0] = 1;
#endif
#ifdef ISX_DIAG_K1_NOPOST
        if (px[0] == 0x7FFFFFFFu) gl = 0;
        continue;
#endif
        // extents from ballots: bit l of word i is pixel 4*l + i of the chunk; the scalar unit finds the first / last
        // set bit, no per-lane min/max and no cross-lane reduction
        uint64_t bc[4], bg[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            bc[i] = __ballot((px[i] & 255u) != 0u);
            bg[i] = C == 3 ? ballot_gray_nz<C>(px[i]) : bc[i];
        }
        const int cx = P.dx0 + chunk * 256;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (bg[i]) {
                gl = min(gl, cx + 4 * (int)__builtin_ctzll(bg[i]) + i);
                gr = max(gr, cx + 4 * (63 - (int)__builtin_clzll(bg[i])) + i);
            }
            if (bc[i]) {
                l0 = min(l0, cx + 4 * (int)__builtin_ctzll(bc[i]) + i);
                r0 = max(r0, cx + 4 * (63 - (int)__builtin_clzll(bc[i])) + i);
            }
        }
        if (lane < 4) mw[chunk * 4 + lane] = lane == 0 ? bc[0] : lane == 1 ? bc[1] : lane == 2 ? bc[2] : bc[3];
    }
    __shared__ __align__(16) int red[FRAME_MAX_WAVES][4];
    if (lane == 0) red[wave][0] = gl, red[wave][1] = gr, red[wave][2] = l0, red[wave][3] = r0; // (wave-uniform values)
    __syncthreads();
    // The bit mask goes to memory as the HORIZONTALLY FEATHERED mask's support: the feather factor is 0 at both edges of
    // the row (alpha = 0 / f, proc.py:222, :226), so the row's first and last channel-0 pixel are cleared here, where
    // the row's extent is known -- K2 then only has to OR rows.  Layout [chunk][row][4]: K2 reads a chunk's rows
    // contiguously.
    if ((int)threadIdx.x < nchunk * 4) { // (nchunk >= 1: thread 0 is always here)
        int4 f = *(const int4 *)red[0];
        for (int w = 1; w < nw; w++) {
            const int4 e = *(const int4 *)red[w];
            f.x = min(f.x, e.x), f.y = max(f.y, e.y), f.z = min(f.z, e.z), f.w = max(f.w, e.w);
        }
        if (threadIdx.x == 0) ((int4 *)rows)[ly] = f;
        for (int w = threadIdx.x; w < nchunk * 4; w += blockDim.x) {
            const int chunk = w >> 2, kk = w & 3, cx = P.dx0 + chunk * 256;
            const int eL = f.z - cx, eR = f.w - cx; // (an empty row has l0 = INT_MAX, r0 = -1: both out of range)
            uint64_t word = mw[w];
            if ((unsigned)eL < 256u && (eL & 3) == kk) word &= ~(1ull << (eL >> 2));
            if ((unsigned)eR < 256u && (eR & 3) == kk) word &= ~(1ull << (eR >> 2));
            mask[((int64_t)chunk * P.dh + ly) * 4 + kk] = word;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2 (gradient path).  grid = nchunk workgroups of 1024.  Every workgroup folds the row table (a few KB from L2),
// so no grid-wide synchronisation is needed; workgroup 0 publishes the result block.  The chunk's bit words (the
// horizontally feathered mask's support: K1 cleared each row's two edge bits) are copied into LDS, OR-ed per 8 and per
// 64 rows, and eight waves walk the three levels: wave k < 4 finds the top row of the 64 columns 256c + 4i + k (bit i of
// ballot word k, the layout K1 stores), wave 4 + k the bottom row.  All global loads (row table + this chunk's bit mask)
// are issued up front, in one round trip: this kernel costs its launch (2.8 us) and the latency of those loads, not its
// bytes and not its barriers (a variant with half the barriers measured the same, profiles/r02_image_kernels.md).
// ------------------------------------------------------------------------------------------------
#define FE_PAIRS_PER_THREAD ((FE_BATCH * 2 + 1023) / 1024)
__global__ void __launch_bounds__(1024) frame_extent_kernel(const int32_t *__restrict__ rows, int qx0, int qy0, int qw,
                                                            int qh, int canvas_w, int canvas_h, int margin,
                                                            const uint64_t *__restrict__ mask, int nchunk,
                                                            int32_t *__restrict__ cols, int cstride,
                                                            int32_t *__restrict__ res) {
    __shared__ __align__(16) uint64_t wm[FE_BATCH][4];
    __shared__ uint64_t or8[FE_BATCH / 8][4];
    __shared__ uint64_t gor[FE_BATCH / 64][4];
    const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
    const int chunk = blockIdx.x;
    const int4 *rt = (const int4 *)rows;
    // ---- fold the row table: bbox of gray != 0 pixels; any channel-0 extent outside [lo, hi) is checked later ----
    int x0 = INT_MAX, y0 = INT_MAX, x1 = -1, y1 = -1, l0min = INT_MAX, r0max = -1;
    // thread t of pass u moves the two words [r][kk], [r][kk + 1] with r = (t + 1024 u) / 2, kk = 2 (t & 1)
    uint4 wv[FE_PAIRS_PER_THREAD];
    const uint4 *mp = (const uint4 *)(mask + (int64_t)chunk * qh * 4);
    const int n0 = min(FE_BATCH, qh); // first batch: rows [0, n0) of Q
#pragma unroll
    for (int u = 0; u < FE_PAIRS_PER_THREAD; u++) {
        const int t = threadIdx.x + 1024 * u, r = min(t >> 1, n0 - 1);
        wv[u] = mp[r * 2 + (t & 1)];
    }
    for (int ly = threadIdx.x; ly < qh; ly += 1024) {
        const int4 e = rt[ly];
        if (e.y >= 0) {
            x0 = min(x0, e.x);
            x1 = max(x1, e.y + 1);
            y0 = min(y0, qy0 + ly);
            y1 = max(y1, qy0 + ly + 1);
        }
        if (e.w >= 0) { // channel-0 extent over ALL rows of Q (a superset of the ROI's rows: conservative)
            l0min = min(l0min, e.z);
            r0max = max(r0max, e.w);
        }
    }
    {   // one block-wide fold of the six values: per wave, then wave 0 folds the 16 wave results (one lane each) and
        // publishes them -- every thread reading all 16 x 6 partials cost ~100 LDS instructions per wave
        __shared__ int red6[FRAME_MAX_WAVES][6];
        __shared__ int fin6[6];
        x0 = wave_min(x0), y0 = wave_min(y0), l0min = wave_min(l0min);
        x1 = wave_max(x1), y1 = wave_max(y1), r0max = wave_max(r0max);
        if (lane == 0) red6[k][0] = x0, red6[k][1] = y0, red6[k][2] = l0min, red6[k][3] = x1, red6[k][4] = y1, red6[k][5] = r0max;
        __syncthreads();
        if (k == 0) {
            const int w = lane & (FRAME_MAX_WAVES - 1); // (lanes 16..63 repeat entries: harmless for min / max)
            x0 = wave_min(red6[w][0]), y0 = wave_min(red6[w][1]), l0min = wave_min(red6[w][2]);
            x1 = wave_max(red6[w][3]), y1 = wave_max(red6[w][4]), r0max = wave_max(red6[w][5]);
            if (lane == 0) fin6[0] = x0, fin6[1] = y0, fin6[2] = l0min, fin6[3] = x1, fin6[4] = y1, fin6[5] = r0max;
        }
        __syncthreads();
        x0 = fin6[0], y0 = fin6[1], l0min = fin6[2], x1 = fin6[3], y1 = fin6[4], r0max = fin6[5];
    }
    const bool found = x1 >= 0;
    const bool first = blockIdx.x == 0 && threadIdx.x == 0;
    if (!found) {
        if (first) {
            res[R_WBBOX + 0] = res[R_WBBOX + 1] = INT_MAX;
            res[R_WBBOX + 2] = res[R_WBBOX + 3] = -1;
            res[R_WBBOX + 4] = 0;
            res[R_ROI + 0] = res[R_ROI + 1] = res[R_ROI + 2] = res[R_ROI + 3] = 0;
            res[R_ROWS_OK] = 1;
        }
        return;
    }
    const int gx0 = max(0, x0 - margin), gy0 = max(0, y0 - margin);
    const int gx1 = min(canvas_w, x1 + margin), gy1 = min(canvas_h, y1 + margin);
    const int ly_lo = max(0, gy0 - qy0), ly_hi = min(qh, gy1 - qy0);
    // rows_ok: no channel-0 pixel outside the ROI's columns (then the row extents K1 took over all of Q are the
    // ROI's own, proc.py:213-218).  Tested over all rows of Q, which can only send more frames to the exact slow path.
    const bool bad = r0max >= 0 && (l0min < gx0 || r0max >= gx1);
    if (first) {
        res[R_WBBOX + 0] = x0, res[R_WBBOX + 1] = y0, res[R_WBBOX + 2] = x1, res[R_WBBOX + 3] = y1, res[R_WBBOX + 4] = 1;
        res[R_ROI + 0] = gx0, res[R_ROI + 1] = gy0, res[R_ROI + 2] = gx1, res[R_ROI + 3] = gy1;
        res[R_ROWS_OK] = !bad;
    }
    if (bad) return;
#ifdef ISX_DIAG_K2_NOCOLS
    return;
#endif
    // ---- column extents.  Per batch of rows: stage the (edge-cleared, ROI-clipped) bit words in LDS, OR them per
    // 8 rows and per 64 rows, then ONE THREAD PER COLUMN walks group words -> 8-row words -> rows: three short
    // levels whose LDS reads are issued together (registers), no cross-lane traffic, no atomics, one writer per column.
    const int n_batch = (qh + FE_BATCH - 1) / FE_BATCH;
    int mytop = INT_MAX, mybot = -1;
    const int ck = (threadIdx.x >> 6) & 3, ci = lane; // column thread (waves 0..3): word ck, bit ci -> column 4*ci + ck
    for (int bi = 0; bi < n_batch; bi++) {
        const int base = bi * FE_BATCH, n = min(FE_BATCH, qh - base);
        __syncthreads();
        if (bi > 0) { // (the first batch was fetched at the top of the kernel)
#pragma unroll
            for (int u = 0; u < FE_PAIRS_PER_THREAD; u++) {
                const int t = threadIdx.x + 1024 * u, r = min(t >> 1, n - 1);
                wv[u] = mp[(base + r) * 2 + (t & 1)];
            }
        }
#pragma unroll
        for (int u = 0; u < FE_PAIRS_PER_THREAD; u++) { // all FE_BATCH rows are written (zeros beyond n)
            const int t = threadIdx.x + 1024 * u, r = t >> 1;
            // (K1 cleared the two edge bits of every row already); rows outside the ROI do not take part
            const bool use = r < n && base + r >= ly_lo && base + r < ly_hi;
            if (r < FE_BATCH) *(uint4 *)&wm[r][2 * (t & 1)] = use ? wv[u] : make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
        if (threadIdx.x < (FE_BATCH / 8) * 4) { // OR of every 8 rows
            const int s8 = threadIdx.x >> 2, kk = threadIdx.x & 3;
            uint64_t o = 0;
#pragma unroll
            for (int r = 0; r < 8; r++) o |= wm[s8 * 8 + r][kk];
            or8[s8][kk] = o;
        }
        __syncthreads();
        if (threadIdx.x < (FE_BATCH / 64) * 4) { // OR of every 64 rows
            const int g = threadIdx.x >> 2, kk = threadIdx.x & 3;
            uint64_t o = 0;
#pragma unroll
            for (int r = 0; r < 8; r++) o |= or8[g * 8 + r][kk];
            gor[g][kk] = o;
        }
        __syncthreads();
#ifdef ISX_DIAG_K2_NOWALK
        if (false) {
#else
        if (k < 8) {
#endif
            // waves 0..3 look for the top row of their 64 columns, waves 4..7 for the bottom row: half the serial chain
            // each.  Only the 32-bit half of a word that holds the lane's bit is read; the bits of a level are gathered
            // into one mask (bfe + shift-or per word) and the row is its lowest / highest set bit
            const bool top = k < 4;
            const int half = ci >> 5, sh = ci & 31;
            const uint32_t *g32 = (const uint32_t *)&gor[0][0] + ck * 2 + half;
            const uint32_t *o32 = (const uint32_t *)&or8[0][0] + ck * 2 + half;
            const uint32_t *w32 = (const uint32_t *)&wm[0][0] + ck * 2 + half;
            uint32_t gm = 0;
#pragma unroll
            for (int g = 0; g < FE_BATCH / 64; g++) gm |= ((g32[g * 8] >> sh) & 1u) << g;
            if (gm) {
                const int g = top ? __builtin_ctz(gm) : 31 - __builtin_clz(gm);
                uint32_t m8 = 0, m1 = 0;
#pragma unroll
                for (int r = 0; r < 8; r++) m8 |= ((o32[(g * 8 + r) * 8] >> sh) & 1u) << r;
                const int s8 = top ? __builtin_ctz(m8) : 31 - __builtin_clz(m8);
                const int row0 = (g * 8 + s8) * 8;
#pragma unroll
                for (int r = 0; r < 8; r++) m1 |= ((w32[(row0 + r) * 8] >> sh) & 1u) << r;
                const int row = qy0 + base + row0 + (top ? __builtin_ctz(m1) : 31 - __builtin_clz(m1));
                if (top) {
                    if (mytop == INT_MAX) mytop = row;
                } else {
                    mybot = row;
                }
            }
        }
    }
    const int lx = chunk * 256 + 4 * ci + ck;
    if (k < 8 && lx < cstride) { // (the padding columns qw .. cstride-1 hold no pixel: INT_MAX / -1)
        if (k < 4)
            cols[lx] = mytop;
        else
            cols[cstride + lx] = mybot;
    }
}

// ------------------------------------------------------------------------------------------------
// K3 (gradient path).  A wave owns a tile of 64 pixels x 4 rows of Q (lane = 4 pixels of one row), a workgroup four
// such tiles side by side, G3_IT row groups deep.  Narrow tiles keep the waves that meet a feather band (10 px wide,
// along the frame's outline) few; everything else is byte-parallel on the packed quads: outside every feather band
// the mask is 1 (warped channel > 0) or 0, for which  warped*mask + ref*(1-mask)  is exactly the warped or the
// reference byte, i.e. a select on "warped byte != 0" -- 8 vector ops per dword instead of ~5 per byte, and this
// kernel is bound by its instruction count, not by its bytes.  mosaic_q = the arena pixel under canvas (qx0, qy0);
// cols = {top[cstride], bottom[cstride]}.
// ------------------------------------------------------------------------------------------------
#define G3_IT 1
#ifdef ISX_K3_REWARP
#define ISX_K3_EXTRA , src_dev, P, (int)persp
#else
#define ISX_K3_EXTRA
#endif
template <int C>
__global__ void __launch_bounds__(256) frame_gradient_kernel(const uint8_t *__restrict__ wbuf, int64_t wpitch,
                                                             uint8_t *mosaic_q, int64_t mpitch, int qx0, int qy0,
                                                             int qw, int qh, const int32_t *__restrict__ rows,
                                                             const int32_t *__restrict__ cols, int cstride, int nchunk,
                                                             int feather, int clip_y0, int clip_y1,
                                                             const int32_t *__restrict__ res, int32_t *__restrict__ part
#ifdef ISX_K3_REWARP
                                                             , const uint8_t *__restrict__ src, const WarpParams P, int persp
#endif
                                                             ) {
    constexpr int NW = QuadRaw<C>::N;
    // per-workgroup partial {minx, miny, maxx, maxy, zeroed, dirty} (plain stores, folded on the host)
    // Tile of this workgroup.  Workgroup ids go round the 8 XCDs (id % 8): XCD x takes the x-th eighth of the tiles in
    // row-major order, so horizontally adjacent tiles share an L2 (they share 128-byte lines at their edges) and an XCD
    // blends the band of rows whose warped pixels K1 wrote from the same XCD (frame_row_of_block).  The grid is padded to
    // a multiple of 8; the workgroups beyond the last tile exit.
    const int n_tiles = nchunk * ((qh + 4 * G3_IT - 1) / (4 * G3_IT)), per_xcd = (n_tiles + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || tile >= n_tiles) return;
    int32_t *mypart = part + 6 * tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, lr = lane >> 4;
    const int bx = tile % nchunk, by = tile / nchunk;
    const int lxw = bx * 256 + wave * 64, lx = lxw + 4 * l16, X = qx0 + lx, Xw = qx0 + lxw;
    const int lyb = by * (4 * G3_IT); // first row of the workgroup
    // every load of the lane is issued before anything depends on the result block (one memory round trip)
    uint32_t dw[G3_IT][NW], dr[G3_IT][NW];
    int L[G3_IT], R[G3_IT];
    int4 T = make_int4(INT_MAX, INT_MAX, INT_MAX, INT_MAX), B = make_int4(-1, -1, -1, -1);
    const int4 *rt = (const int4 *)rows;
#pragma unroll
    for (int it = 0; it < G3_IT; it++) {
        const int ly = lyb + 4 * it + lr, lyc = min(ly, qh - 1), Y = qy0 + ly;
        // (rows beyond Q and rows of another strip -- not addressable in this GPU's arena -- read as 0)
        const bool rd = lx < qw && ly < qh;
#ifdef ISX_K3_REWARP
        {
            uint32_t px4[4];
            const int nv = rd ? min(4, qw - lx) : 0;
            if (persp)
                warp_quad_f32<C, true>(src, P, qx0 + lx, Y, nv, px4);
            else
                warp_quad_f32<C, false>(src, P, qx0 + lx, Y, nv, px4);
            raw_pack<C>(px4, dw[it]);
        }
#else
        raw_load<C>(wbuf + (int64_t)lyc * wpitch, lx, rd ? qw : 0, dw[it]);
#endif
        raw_load<C>(mosaic_q + (int64_t)lyc * mpitch, lx, (rd && Y >= clip_y0 && Y < clip_y1) ? qw : 0, dr[it]);
#ifdef ISX_DIAG_K3_NOFEATHER
        L[it] = INT_MAX, R[it] = -1;
#else
        const int4 e = rt[lyc];
        L[it] = e.z, R[it] = e.w;
#endif
    }
#ifndef ISX_DIAG_K3_NOFEATHER
    if (lx < qw) { // (cstride is a multiple of 4 and >= qw: the 16-byte loads stay inside the two arrays)
        T = *(const int4 *)(cols + lx);
        B = *(const int4 *)(cols + cstride + lx);
    }
#endif
    const int found = res[R_WBBOX + 4], rows_ok = res[R_ROWS_OK];
    // ROI rows [ry0, ry1) clip the feather bands; rows outside [clip_y0, clip_y1) belong to another strip of the mosaic
    // (another GPU) and are never read or written here: [gy0, gy1) is what this call blends
    const int gx0 = res[R_ROI + 0], ry0 = res[R_ROI + 1], gx1 = res[R_ROI + 2], ry1 = res[R_ROI + 3];
    const int gy0 = max(ry0, clip_y0), gy1 = min(ry1, clip_y1);
    // alpha = k / feather for k = 0 .. feather (proc.py:220, :225, :239, :244): the same IEEE quotient as the reference's
    // per-pixel division, formed ONCE PER WAVE THAT MEETS A FEATHER BAND into that wave's own table (no workgroup barrier:
    // the nine waves in ten that never read it pay neither the float64 division nor the wait for it)
    __shared__ double ftab_w[4][64];
    __shared__ int red[4][6];
    double *ftab = ftab_w[wave];
    bool have_ftab = false; // wave-uniform
    if (threadIdx.x == 0) mypart[0] = mypart[1] = INT_MAX, mypart[2] = mypart[3] = -1, mypart[4] = mypart[5] = 0;
    if (!found || !rows_ok) return;
    // workgroup entirely outside the ROI (block-uniform)
    if (qx0 + bx * 256 >= gx1 || qx0 + bx * 256 + 256 <= gx0 || qy0 + lyb >= gy1 ||
        qy0 + lyb + 4 * G3_IT <= gy0)
        return;
    const int T4[4] = {T.x, T.y, T.z, T.w}, B4[4] = {B.x, B.y, B.z, B.w};
    const bool cols_all = Xw >= gx0 && Xw + 64 <= gx1 && lxw + 64 <= qw;     // wave-uniform
    const bool wave_out = Xw >= gx1 || Xw + 64 <= gx0 || lxw >= qw;         // wave-uniform
    int sminx = INT_MAX, sminy = INT_MAX, smaxx = -1, smaxy = -1;           // wave-uniform (fast path)
    int vminx = INT_MAX, vminy = INT_MAX, vmaxx = -1, vmaxy = -1;           // per lane (general path)
    bool zeroed = false, dirty = false, any_general = false;
#pragma unroll
    for (int it = 0; it < G3_IT; it++) {
        const int lyw = lyb + 4 * it, Yw = qy0 + lyw; // first row of the wave's tile (wave-uniform)
        if (wave_out || Yw >= gy1 || Yw + 4 <= gy0 || lyw >= qh) continue;
        const int ly = lyw + lr, Y = qy0 + ly;
        // does any pixel of this lane sit in a feather band (conservative: clipping only shrinks the bands).  An empty
        // row / column has L = T = INT_MAX, R = B = -1: the unsigned differences are then huge
        bool feathered = (unsigned)(X + 3 - L[it]) < (unsigned)(feather + 3) || (unsigned)(R[it] - X) <= (unsigned)(feather + 3);
#pragma unroll
        for (int k = 0; k < 4; k++)
            feathered = feathered || (unsigned)(Y - T4[k]) < (unsigned)feather || (unsigned)(B4[k] - Y) <= (unsigned)feather;
#ifdef ISX_DIAG_K3_NOSLOW
        const bool slow = false;
#else
        const bool slow = __any(feathered);
#endif
        const bool rows_all = Yw >= gy0 && Yw + 4 <= gy1 && lyw + 4 <= qh; // wave-uniform
        if (cols_all && rows_all && !slow) {
            // ---- byte-parallel path: every pixel of the tile is inside the ROI and outside every feather band ----
            uint32_t any_w = 0;
#pragma unroll
            for (int j = 0; j < NW; j++) any_w |= dw[it][j];
            // nothing warped onto this tile (the ROI's 50 px margin): the mosaic stays as it is.  Its gray != 0 pixels
            // lie inside the old mosaic rectangle already, so they need not be reported
            if (!__any(any_w != 0u)) continue;
            uint32_t o[NW];
#pragma unroll
            for (int j = 0; j < NW; j++) o[j] = dw[it][j] | (dr[it][j] & ~nz_bytes(dw[it][j]));
            raw_store<C>(mosaic_q + (int64_t)ly * mpitch, lx, qw, o);
#ifdef ISX_DIAG_K3_NOBOOK
            if (true) {
#else
            if (__all(raw_all_gray_nz<C>(o))) { // the whole tile is written with gray != 0
#endif
                sminx = min(sminx, Xw), smaxx = max(smaxx, Xw + 64), sminy = min(sminy, Yw), smaxy = max(smaxy, Yw + 4);
            } else {
                uint32_t po[4], pr[4];
                raw_unpack<C>(o, po);
                raw_unpack<C>(dr[it], pr);
                any_general = true;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (gray_px<C>(po[k]) != 0) {
                        vminx = min(vminx, X + k), vmaxx = max(vmaxx, X + k + 1), vminy = min(vminy, Y), vmaxy = max(vmaxy, Y + 1);
                    } else {
                        dirty = dirty || po[k] != 0u;
                        zeroed = zeroed || gray_px<C>(pr[k]) != 0;
                    }
                }
            }
            continue;
        }
        // ---- general path: per pixel, ROI borders and feather bands ----
#ifdef ISX_DIAG_K3_NOGENERAL
        continue;
#endif
        any_general = true;
        if (slow && !have_ftab) { // (feather < 64: one lane per entry; LDS accesses of a wave complete in order)
            if (lane <= feather) ftab[lane] = (double)lane / (double)feather;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            have_ftab = true;
        }
        const bool row_in = ly < qh && Y >= gy0 && Y < gy1;
        uint32_t pw[4], pr[4], po[4];
        raw_unpack<C>(dw[it], pw);
        raw_unpack<C>(dr[it], pr);
        bool any_in = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int xx = X + k;
            const bool in = row_in && xx >= gx0 && xx < gx1 && lx + k < qw;
            uint32_t o = 0;
            if (!slow) {
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const uint32_t wv = (pw[k] >> (8 * c)) & 255u, rv = (pr[k] >> (8 * c)) & 255u;
                    o |= (wv ? wv : rv) << (8 * c);
                }
            } else {
                // factor chain of proc.py:211-246 in ROI-relative form: [L, min(L+f, W)) and [max(0, R-f), R].  Branch
                // free: a factor that does not apply is read as ftab[feather] = 1.0 (an exact no-op), so the four
                // LDS reads of a pixel are in flight together
                const int Lr = L[it], Rr = R[it];
                const int xe = Lr + feather < gx1 ? Lr + feather : gx1, xb = Rr - feather > gx0 ? Rr - feather : gx0;
                const int ye = T4[k] + feather < ry1 ? T4[k] + feather : ry1, yb = B4[k] - feather > ry0 ? B4[k] - feather : ry0;
                const int i0 = (Rr >= 0 && xx >= Lr && xx < xe) ? xx - Lr : feather;
                const int i1 = (Rr >= 0 && xx >= xb && xx <= Rr) ? Rr - xx : feather;
                const int i2 = (B4[k] >= 0 && Y >= T4[k] && Y < ye) ? Y - T4[k] : feather;
                const int i3 = (B4[k] >= 0 && Y >= yb && Y <= B4[k]) ? B4[k] - Y : feather;
                const double m = ((ftab[i0] * ftab[i1]) * ftab[i2]) * ftab[i3];
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const int wv = (pw[k] >> (8 * c)) & 255, rv = (pr[k] >> (8 * c)) & 255;
                    const double mk = wv > 0 ? m : 0.0;
                    const double r = (double)wv * mk + (double)rv * (1 - mk);
                    o |= (uint32_t)(uint8_t)(int)r << (8 * c);
                }
            }
            po[k] = in ? o : pr[k];
            any_in = any_in || in;
            if (in) {
                if (gray_px<C>(o) != 0) {
                    vminx = min(vminx, xx), vmaxx = max(vmaxx, xx + 1), vminy = min(vminy, Y), vmaxy = max(vmaxy, Y + 1);
                } else {
                    dirty = dirty || o != 0u;
                    zeroed = zeroed || gray_px<C>(pr[k]) != 0;
                }
            }
        }
        if (any_in) { // (pixels of the quad outside the ROI are written back unchanged)
            uint32_t o[NW];
            raw_pack<C>(po, o);
            raw_store<C>(mosaic_q + (int64_t)ly * mpitch, lx, qw, o);
        }
    }
    if (__any(any_general)) {
        sminx = min(sminx, wave_min(vminx)), sminy = min(sminy, wave_min(vminy));
        smaxx = max(smaxx, wave_max(vmaxx)), smaxy = max(smaxy, wave_max(vmaxy));
    }
    const int zf = __any(zeroed), df = __any(dirty);
    if (lane == 0)
        red[wave][0] = sminx, red[wave][1] = sminy, red[wave][2] = smaxx, red[wave][3] = smaxy, red[wave][4] = zf, red[wave][5] = df;
    __syncthreads();
    if (threadIdx.x == 0) {
        int a0 = red[0][0], a1 = red[0][1], a2 = red[0][2], a3 = red[0][3], a4 = red[0][4], a5 = red[0][5];
        for (int w = 1; w < 4; w++) {
            a0 = min(a0, red[w][0]);
            a1 = min(a1, red[w][1]);
            a2 = max(a2, red[w][2]);
            a3 = max(a3, red[w][3]);
            a4 |= red[w][4];
            a5 |= red[w][5];
        }
        mypart[0] = a0, mypart[1] = a1, mypart[2] = a2, mypart[3] = a3, mypart[4] = a4, mypart[5] = a5;
    }
}

// ------------------------------------------------------------------------------------------------
// alpha path: warp + seamless_merge in place over all of Q (outside Q the warped image is 0 and the merged
// canvas equals the mosaic).  Per-row table (6 ints): warped gray extent, merged gray extent, zeroed, dirty.
// ------------------------------------------------------------------------------------------------
// (5 waves per SIMD = 96 VGPRs: the 1190 four-wave row workgroups of a 1080p frame are then all resident at once; at the
// 104 VGPRs the compiler takes by itself a fifth of them waits for a slot and runs as a second generation: 15.2 -> 13.7 us)
#ifndef ISX_KA_WAVES
#define ISX_KA_WAVES 5
#endif
template <int C, bool PERSP, bool CLASSIC>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(ISX_KA_WAVES))) frame_warp_alpha_kernel(const uint8_t *__restrict__ src, const WarpParams P,
                                                               uint8_t *mosaic_q, int64_t mpitch, int nchunk,
                                                               double contrib, double one_minus, int clip_y0, int clip_y1,
                                                               int32_t *__restrict__ rows) {
    const int ly = frame_row_of_block(P.dh);
    if (ly >= P.dh) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    // trunc(c * ref) and trunc((1 - c) * warped) of proc.py:70-71 for every byte value, the float64 products formed once per
    // workgroup: in the overlap a channel costs two LDS byte reads (256-byte tables: one dword per bank, no conflicts)
    // instead of two int -> double conversions, two float64 products and two conversions back
    __shared__ uint8_t tab_ref[256], tab_warped[256];
    for (int t = threadIdx.x; t < 256; t += blockDim.x) {
        tab_ref[t] = (uint8_t)(int)(contrib * (double)t);
        tab_warped[t] = (uint8_t)(int)(one_minus * (double)t);
    }
    __syncthreads();
    int gl = INT_MAX, gr = -1, ml = INT_MAX, mr = -1, zeroed = 0, dirty = 0; // (wave-uniform: found from ballots)
    uint8_t *mrow = mosaic_q + (int64_t)ly * mpitch;
    // rows of another strip of the mosaic (another GPU): the warped row still counts for bbox(warped), the mosaic is
    // neither read nor written
    const bool mine = P.dy0 + ly >= clip_y0 && P.dy0 + ly < clip_y1;
    for (int chunk = wave; chunk < nchunk; chunk += nw) {
        const int lx = chunk * 256 + 4 * lane;
        const int nv = min(4, max(0, P.dw - lx));
        // (pixels at or beyond the end of the row: the warp yields 0 for them, the mosaic reads as 0, so they blend to 0
        // and take no part in any extent or flag below)
        uint32_t px[4], pb[4] = {0u, 0u, 0u, 0u}, po[4];
        if (nv > 0 && mine) quad_load<C>(mrow, lx, P.dw, pb);
        warp_quad<C, PERSP, CLASSIC>(src, P, P.dx0 + lx, P.dy0 + ly, nv, px);
        if (!__any((pb[0] | pb[1] | pb[2] | pb[3]) != 0u)) {
            // nothing of the mosaic under this wave: warped + 0
#pragma unroll
            for (int i = 0; i < 4; i++) po[i] = px[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // outside the overlap: uint8 + uint8 with wrap-around (proc.py:79), all channels at once (carries between
                // the bytes suppressed)
                po[i] = ((px[i] & 0x7F7F7Fu) + (pb[i] & 0x7F7F7Fu)) ^ ((px[i] ^ pb[i]) & 0x808080u);
                if (gray_px<C>(px[i]) > 10 && gray_px<C>(pb[i]) > 10) { // overlap = mask1 & mask2 (proc.py:58-66)
                    uint32_t o = 0;
#pragma unroll
                    for (int c = 0; c < C; c++)
                        o |= (((uint32_t)tab_warped[(px[i] >> (8 * c)) & 255u] + (uint32_t)tab_ref[(pb[i] >> (8 * c)) & 255u]) & 255u) << (8 * c);
                    po[i] = o;
                }
            }
        }
        if (nv > 0 && mine) quad_store<C>(mrow, lx, P.dw, po);
        // extents and flags from ballots (bit l of word i = pixel 4*l + i of the chunk), on the scalar unit
        const int cx = P.dx0 + chunk * 256;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t gw = ballot_gray_nz<C>(px[i]);
            if (gw) {
                gl = min(gl, cx + 4 * (int)__builtin_ctzll(gw) + i);
                gr = max(gr, cx + 4 * (63 - (int)__builtin_clzll(gw)) + i);
            }
            if (!mine) continue;
            const uint64_t go = ballot_gray_nz<C>(po[i]);
            if (go) {
                ml = min(ml, cx + 4 * (int)__builtin_ctzll(go) + i);
                mr = max(mr, cx + 4 * (63 - (int)__builtin_clzll(go)) + i);
            }
            if (~go) { // some pixel is written with gray == 0: was it gray != 0 before (zeroed), is a byte of it != 0 (dirty)
                if (C == 3 && (__ballot(po[i] != 0u) & ~go)) dirty = 1; // (C = 1: gray == 0 means the byte is 0)
                if (ballot_gray_nz<C>(pb[i]) & ~go) zeroed = 1;
            }
        }
    }
    __shared__ int red[FRAME_MAX_WAVES][6];
    if (lane == 0)
        red[wave][0] = gl, red[wave][1] = gr, red[wave][2] = ml, red[wave][3] = mr, red[wave][4] = zeroed,
        red[wave][5] = dirty;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; w++) {
            gl = min(gl, red[w][0]);
            gr = max(gr, red[w][1]);
            ml = min(ml, red[w][2]);
            mr = max(mr, red[w][3]);
            zeroed |= red[w][4];
            dirty |= red[w][5];
        }
        int32_t *o = rows + 6 * ly;
        o[0] = gl, o[1] = gr, o[2] = ml, o[3] = mr, o[4] = zeroed, o[5] = dirty;
    }
}

static void fill_warp_params(WarpParams &P, const double *M_host, int model, int arith, int canvas_w, int canvas_h,
                             int sh, int sw, int64_t spitch, int src_x, int src_y, int qx0, int qy0, int qw, int qh,
                             int64_t dpitch) {
    double iM[9];
    if (model == ISX_MODEL_HOMOGRAPHY)
        invert_3x3_cv(M_host, iM);
    else
        invert_affine_cv(M_host, iM);
    for (int i = 0; i < 9; i++) P.m[i] = (float)iM[i], P.im[i] = iM[i];
    P.bw0 = 64;
    if (arith == ISX_WARP_CLASSIC && model == ISX_MODEL_HOMOGRAPHY) {
        const int bh0 = canvas_h < 16 ? canvas_h : 16;
        P.bw0 = 1024 / bh0 < canvas_w ? 1024 / bh0 : canvas_w;
    }
    P.src_x = src_x, P.src_y = src_y, P.sh = sh, P.sw = sw;
    P.dx0 = qx0, P.dy0 = qy0, P.dw = qw, P.dh = qh;
    P.spitch = spitch, P.dpitch = dpitch;
}

static void empty_result(int32_t *r) {
    memset(r, 0, R_INTS * 4);
    r[R_WBBOX + 0] = r[R_WBBOX + 1] = r[R_MBBOX + 0] = r[R_MBBOX + 1] = INT_MAX;
    r[R_WBBOX + 2] = r[R_WBBOX + 3] = r[R_MBBOX + 2] = r[R_MBBOX + 3] = -1;
    r[R_ROWS_OK] = 1;
}

extern "C" int isx_frame_blend(isx_ctx *ctx, const uint8_t *src_dev, int sh, int sw, int c, int64_t spitch, int src_x,
                               int src_y, const double *M_host, int model, int arith, int canvas_w, int canvas_h,
                               uint8_t *mosaic_q_dev, int64_t mpitch, int q_x0, int q_y0, int q_w, int q_h, int clip_y0,
                               int clip_y1, int blender, double ref_contrib, int feather, int32_t *result_host,
                               void *stream) {
    ISX_REQUIRE(ctx, "null ctx");
    ISX_REQUIRE(c == 1 || c == 3, "channels must be 1 or 3 (got %d)", c);
    ISX_REQUIRE(model == ISX_MODEL_AFFINE_PARTIAL || model == ISX_MODEL_HOMOGRAPHY, "bad model %d", model);
    ISX_REQUIRE(arith == ISX_WARP_LINEAR_F32 || arith == ISX_WARP_CLASSIC, "bad warp arithmetic %d", arith);
    ISX_REQUIRE(blender == ISX_BLEND_ALPHA || blender == ISX_BLEND_GRADIENT, "bad blender %d", blender);
    ISX_REQUIRE(M_host, "null M_host");
    ISX_REQUIRE(sh > 0 && sw > 1 && spitch >= (int64_t)sw * c, "bad source geometry (the frame must be at least 2 pixels wide)");
    ISX_REQUIRE((int64_t)sh * spitch < ((int64_t)1 << 31), "the frame must be smaller than 2 GiB (the mosaic may be larger)");
    ISX_REQUIRE(canvas_w > 0 && canvas_h > 0 && q_x0 >= 0 && q_y0 >= 0 && q_w >= 0 && q_h >= 0 &&
                    q_x0 + q_w <= canvas_w && q_y0 + q_h <= canvas_h,
                "Q = (%d,%d %dx%d) outside the %dx%d canvas", q_x0, q_y0, q_w, q_h, canvas_w, canvas_h);
    ISX_REQUIRE(q_w <= ISX_FRAME_MAX_Q && q_h <= ISX_FRAME_MAX_Q, "Q larger than %d pixels on a side", ISX_FRAME_MAX_Q);
    ISX_REQUIRE(blender == ISX_BLEND_ALPHA || (feather > 0 && feather < 64), "feather must be in 1..63");
    int cur = -1;
    ISX_HIP_TRY(hipGetDevice(&cur));
    ISX_REQUIRE(cur == ctx->device, "ctx belongs to device %d but device %d is current", ctx->device, cur);
    hipStream_t st = (hipStream_t)stream;
    ctx->pending = 0;
    if (q_w == 0 || q_h == 0) { // nothing of the frame lands on the canvas
        empty_result(ctx->last);
        if (result_host) memcpy(result_host, ctx->last, R_INTS * 4);
        return ISX_OK;
    }
    ISX_REQUIRE(src_dev && mosaic_q_dev && mpitch >= (int64_t)q_w * c, "null device pointer / bad mosaic pitch");
    const int nchunk = (q_w + 255) / 256;
    // waves per row workgroup: FOUR, each taking every fourth 256-pixel chunk of the row (two chunks at 1080p).  All row
    // workgroups of a 1080p frame are then resident at once (1190 x 4 waves on 1024 SIMDs at 5-6 waves each) instead of
    // running as two generations that each wait for their first loads: the fused alpha kernel measured 1 us faster than
    // with a wave per chunk, the warp kernel 0.5 us (11.8 -> 11.3; two waves per row: 13.1)
#ifndef ISX_K1_MAX_WAVES
#define ISX_K1_MAX_WAVES 4
#endif
    const int max_waves = blender == ISX_BLEND_GRADIENT ? ISX_K1_MAX_WAVES : 4;
    const int row_threads = 64 * (nchunk < max_waves ? nchunk : max_waves);
    const int row_blocks = 8 * ((q_h + 7) / 8); // frame_row_of_block()
    const bool persp = model == ISX_MODEL_HOMOGRAPHY, classic = arith == ISX_WARP_CLASSIC;
    int rc;
    WarpParams P;
    if (blender == ISX_BLEND_GRADIENT) {
        const int64_t wpitch = (int64_t)q_w * c;
        const int n_part = nchunk * ((q_h + 4 * G3_IT - 1) / (4 * G3_IT)); // K3's tiles (256 pixels x 4 rows)
        dim3 gb(8 * ((n_part + 7) / 8));
        const int cstride = (q_w + 3) & ~3;
        if ((rc = ctx_ensure(&ctx->wbuf, &ctx->wbuf_cap, (size_t)wpitch * q_h + 16)) != ISX_OK) return rc;
        if ((rc = ctx_ensure(&ctx->rows, &ctx->rows_cap, (size_t)4 * q_h)) != ISX_OK) return rc;
        if ((rc = ctx_ensure(&ctx->cols, &ctx->cols_cap, (size_t)2 * cstride)) != ISX_OK) return rc;
        if ((rc = ctx_ensure(&ctx->mask, &ctx->mask_cap, (size_t)q_h * nchunk * 4)) != ISX_OK) return rc;
        if ((rc = ctx_ensure_out(ctx, (size_t)R_INTS + (size_t)6 * n_part)) != ISX_OK) return rc;
        fill_warp_params(P, M_host, model, arith, canvas_w, canvas_h, sh, sw, spitch, src_x, src_y, q_x0, q_y0, q_w, q_h, wpitch);
#define ISX_FW_LAUNCH(CC, PP, AA)                                                                                \
    hipLaunchKernelGGL((frame_warp_kernel<CC, PP, AA>), dim3(row_blocks), dim3(row_threads), 0, st, src_dev, P, ctx->wbuf, \
                       nchunk, ctx->rows, ctx->mask)
        if (c == 3) {
            if (persp && classic) ISX_FW_LAUNCH(3, true, true);
            else if (persp) ISX_FW_LAUNCH(3, true, false);
            else if (classic) ISX_FW_LAUNCH(3, false, true);
            else ISX_FW_LAUNCH(3, false, false);
        } else {
            if (persp && classic) ISX_FW_LAUNCH(1, true, true);
            else if (persp) ISX_FW_LAUNCH(1, true, false);
            else if (classic) ISX_FW_LAUNCH(1, false, true);
            else ISX_FW_LAUNCH(1, false, false);
        }
#undef ISX_FW_LAUNCH
        hipLaunchKernelGGL(frame_extent_kernel, dim3(nchunk), dim3(1024), 0, st, ctx->rows, q_x0, q_y0, q_w, q_h, canvas_w,
                           canvas_h, 50, ctx->mask, nchunk, ctx->cols, cstride, ctx->out_dev);
        if (c == 3)
            hipLaunchKernelGGL((frame_gradient_kernel<3>), gb, dim3(256), 0, st, ctx->wbuf, wpitch, mosaic_q_dev, mpitch, q_x0,
                               q_y0, q_w, q_h, ctx->rows, ctx->cols, cstride, nchunk, feather, clip_y0, clip_y1, ctx->out_dev, ctx->out_dev + R_INTS ISX_K3_EXTRA);
        else
            hipLaunchKernelGGL((frame_gradient_kernel<1>), gb, dim3(256), 0, st, ctx->wbuf, wpitch, mosaic_q_dev, mpitch, q_x0,
                               q_y0, q_w, q_h, ctx->rows, ctx->cols, cstride, nchunk, feather, clip_y0, clip_y1, ctx->out_dev, ctx->out_dev + R_INTS ISX_K3_EXTRA);
        ctx->fold_n = n_part;
        ctx->pending = 1;
    } else {
        if ((rc = ctx_ensure_out(ctx, (size_t)R_INTS + (size_t)6 * q_h)) != ISX_OK) return rc;
        fill_warp_params(P, M_host, model, arith, canvas_w, canvas_h, sh, sw, spitch, src_x, src_y, q_x0, q_y0, q_w, q_h, mpitch);
        const double one_minus = 1 - ref_contrib; // Python: (1-ref_image_contrib)
#define ISX_FA_LAUNCH(CC, PP, AA)                                                                                     \
    hipLaunchKernelGGL((frame_warp_alpha_kernel<CC, PP, AA>), dim3(row_blocks), dim3(row_threads), 0, st, src_dev, P, mosaic_q_dev, \
                       mpitch, nchunk, ref_contrib, one_minus, clip_y0, clip_y1, ctx->out_dev + R_INTS)
        if (c == 3) {
            if (persp && classic) ISX_FA_LAUNCH(3, true, true);
            else if (persp) ISX_FA_LAUNCH(3, true, false);
            else if (classic) ISX_FA_LAUNCH(3, false, true);
            else ISX_FA_LAUNCH(3, false, false);
        } else {
            if (persp && classic) ISX_FA_LAUNCH(1, true, true);
            else if (persp) ISX_FA_LAUNCH(1, true, false);
            else if (classic) ISX_FA_LAUNCH(1, false, true);
            else ISX_FA_LAUNCH(1, false, false);
        }
#undef ISX_FA_LAUNCH
        ctx->fold_n = q_h;
        ctx->fold_qy0 = q_y0;
        ctx->pending = 2;
    }
    if ((rc = isx_launch_status("frame kernels")) != ISX_OK) return rc;
    if (!result_host) return ISX_OK; // queued only: the caller fetches the result block with isx_frame_result
    return isx_frame_result(ctx, result_host, stream);
}

extern "C" int isx_frame_result(isx_ctx *ctx, int32_t *result_host, void *stream) {
    ISX_REQUIRE(ctx && result_host, "null ctx / result_host");
    if (ctx->pending) {
        hipStream_t st = (hipStream_t)stream;
        const size_t ints = (size_t)R_INTS + (size_t)6 * ctx->fold_n;
        ISX_HIP_TRY(hipMemcpyAsync(ctx->out_host, ctx->out_dev, ints * 4, hipMemcpyDeviceToHost, st));
        ISX_HIP_TRY(hipStreamSynchronize(st));
        int32_t *r = ctx->last;
        const int32_t *t = ctx->out_host + R_INTS;
        empty_result(r);
        if (ctx->pending == 1) { // gradient: K2's block + the fold of K3's partials
            memcpy(r, ctx->out_host, (R_ROI + 4) * 4);
            r[R_ROWS_OK] = ctx->out_host[R_ROWS_OK];
            for (int i = 0; i < ctx->fold_n; i++, t += 6) {
                if (t[2] >= 0) {
                    if (t[0] < r[R_MBBOX + 0]) r[R_MBBOX + 0] = t[0];
                    if (t[1] < r[R_MBBOX + 1]) r[R_MBBOX + 1] = t[1];
                    if (t[2] > r[R_MBBOX + 2]) r[R_MBBOX + 2] = t[2];
                    if (t[3] > r[R_MBBOX + 3]) r[R_MBBOX + 3] = t[3];
                }
                r[R_ZEROED] |= t[4];
                r[R_DIRTY] |= t[5];
            }
            r[R_MBBOX + 4] = r[R_MBBOX + 2] >= 0;
            r[R_DONE] = r[R_WBBOX + 4] && r[R_ROWS_OK];
        } else { // alpha: the fold of the row table
            for (int i = 0; i < ctx->fold_n; i++, t += 6) {
                const int y = ctx->fold_qy0 + i;
                if (t[1] >= 0) {
                    if (t[0] < r[R_WBBOX + 0]) r[R_WBBOX + 0] = t[0];
                    if (t[1] + 1 > r[R_WBBOX + 2]) r[R_WBBOX + 2] = t[1] + 1;
                    if (y < r[R_WBBOX + 1]) r[R_WBBOX + 1] = y;
                    r[R_WBBOX + 3] = y + 1;
                }
                if (t[3] >= 0) {
                    if (t[2] < r[R_MBBOX + 0]) r[R_MBBOX + 0] = t[2];
                    if (t[3] + 1 > r[R_MBBOX + 2]) r[R_MBBOX + 2] = t[3] + 1;
                    if (y < r[R_MBBOX + 1]) r[R_MBBOX + 1] = y;
                    r[R_MBBOX + 3] = y + 1;
                }
                r[R_ZEROED] |= t[4];
                r[R_DIRTY] |= t[5];
            }
            r[R_WBBOX + 4] = r[R_WBBOX + 2] >= 0;
            r[R_MBBOX + 4] = r[R_MBBOX + 2] >= 0;
            r[R_DONE] = 1;
        }
        ctx->pending = 0;
    }
    memcpy(result_host, ctx->last, R_INTS * 4);
    return ISX_OK;
}

// The staging image of the last gradient-path call (the warped frame over Q, pitch q_w * c): what the caller's
// slow path blends from when the result says rows_ok == 0.
extern "C" int isx_frame_warped(isx_ctx *ctx, const uint8_t **wbuf_dev_out) {
    ISX_REQUIRE(ctx && wbuf_dev_out, "null argument");
    *wbuf_dev_out = ctx->wbuf;
    return ISX_OK;
}

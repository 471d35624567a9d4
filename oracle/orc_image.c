/*
 * oracle/orc_image.c -- TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * Image-space half of the hot path, restated from the reference's utils/proc.py (line by line for
 * what the reference computes itself) and from OpenCV 4.x for what it delegates ([OCV-K], parity
 * unpinned: cv2 absent).  All images are uint8, row-major, interleaved channels, c in {1,3}.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* [OCV-K] cv2.cvtColor(BGR2GRAY), 8-bit: (B*3735 + G*19235 + R*9798 + 2^14) >> 15 */
static inline uint8_t gray_of(const uint8_t *p) {
    return (uint8_t)((p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + 16384) >> 15);
}

void orc_bgr2gray(const uint8_t *img, int h, int w, uint8_t *gray_out) {
    for (long i = 0; i < (long)h * w; i++) gray_out[i] = gray_of(img + 3 * i);
}

/* proc.py:34-45 get_corners_from_image: gray (if 3-D) -> cv2.findNonZero -> cv2.boundingRect.
 * bbox_out = x, y, x+w, y+h.  Returns 0 when the image has no non-zero gray pixel (reference: False). */
int orc_bbox_nonzero(const uint8_t *img, int h, int w, int c, int32_t *bbox_out) {
    int x0 = w, y0 = h, x1 = -1, y1 = -1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t *p = img + ((long)y * w + x) * c;
            int g = c == 3 ? gray_of(p) : p[0];
            if (g) {
                if (x < x0) x0 = x;
                if (x > x1) x1 = x;
                if (y < y0) y0 = y;
                if (y > y1) y1 = y;
            }
        }
    if (x1 < 0) return 0;
    bbox_out[0] = x0; bbox_out[1] = y0; bbox_out[2] = x1 + 1; bbox_out[3] = y1 + 1;
    return 1;
}

/* proc.py:280-292 pad_image: zero canvas, image pasted at ((new_w-w)//2, (new_h-h)//2) */
void orc_pad_image(const uint8_t *img, int h, int w, int c, int new_h, int new_w, uint8_t *out) {
    memset(out, 0, (size_t)new_h * new_w * c);
    int xc = (new_w - w) / 2, yc = (new_h - h) / 2;
    for (int y = 0; y < h; y++)
        memcpy(out + ((size_t)(y + yc) * new_w + xc) * c, img + (size_t)y * w * c, (size_t)w * c);
}

/* ---- warp ------------------------------------------------------------------------------------ */
/* [OCV-K] cv::warpAffine / cv::warpPerspective with the defaults the reference uses (stitcher.py:146,148):
 * INTER_LINEAR, BORDER_CONSTANT(0), M is the forward map and is inverted in double first.
 *
 * arith 0 ("B", the OpenCV 4.11 linear kernels, modules/imgproc/src/warp_kernels.simd.hpp): the inverse
 * matrix is cast to float32; source coordinates, the two fractional weights and the three lerps are
 * float32 fused multiply-adds; the result is rounded to nearest-even and saturated.
 *
 * arith 1 ("A", the classic path of OpenCV <= 4.10, modules/imgproc/src/imgwarp.cpp WarpAffineInvoker /
 * WarpPerspectiveInvoker + remapBilinear<FixedPtCast<int,uchar,15>>): source coordinates in 1/32 px
 * fixed point -- affine: X = (cvRound((M1*y + M2)*1024) + 16 + cvRound(M0*x*1024)) >> 5 on the double
 * inverse; perspective: per pixel in double, W = 32/w, X = cvRound(clamp((X0 + M0*x1)*W)) with X0 taken
 * at the start of the 64-column block the pixel lies in (bw0/bh0 blocking of the invoker) -- integer part
 * X>>5 (saturated to short), fraction X&31; bilinear weights from the 32x32 int16 table whose rows sum to
 * 32768 (entry (0,0) is {32767,0,0,1}: 1.0 saturates to 32767 and the table's sum correction adds the
 * missing 1 to the last tap); value = (sum(tap*w) + 16384) >> 15; taps outside the image are the border 0.
 */
static void invert_affine(const double *M, double *iM) {
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = M[4] * D, A22 = M[0] * D;
    iM[0] = A11; iM[1] = M[1] * (-D); iM[3] = M[3] * (-D); iM[4] = A22;
    iM[2] = -iM[0] * M[2] - iM[1] * M[5];
    iM[5] = -iM[3] * M[2] - iM[4] * M[5];
}

/* [OCV-K] cv::invert(3x3, DECOMP_LU) closed form (adjugate / determinant, double) */
static void invert_3x3(const double *m, double *o) {
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

static inline float px(const uint8_t *src, int sh, int sw, int c, int x, int y, int ch) {
    if ((unsigned)x >= (unsigned)sw || (unsigned)y >= (unsigned)sh) return 0.f;
    return (float)src[((long)y * sw + x) * c + ch];
}

static inline int tap(const uint8_t *src, int sh, int sw, int c, int x, int y, int ch) {
    if ((unsigned)x >= (unsigned)sw || (unsigned)y >= (unsigned)sh) return 0;
    return src[((long)y * sw + x) * c + ch];
}

static inline int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

/* cv::saturate_cast<int>(double) = cvRound: round half to even in the default rounding mode */
static inline int cv_round(double v) { return (int)lrint(v); }

static void warp_classic(const uint8_t *src, int sh, int sw, int c, const double *iM, int perspective,
                         uint8_t *dst, int dh, int dw) {
    /* block geometry of WarpPerspectiveInvoker (BLOCK_SZ = 32) */
    int bh0 = 16 < dh ? 16 : dh;
    int bw0 = 1024 / bh0 < dw ? 1024 / bh0 : dw;
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int X, Y;
            if (!perspective) {
                const int X0 = cv_round((iM[1] * y + iM[2]) * 1024) + 16;
                const int Y0 = cv_round((iM[4] * y + iM[5]) * 1024) + 16;
                X = (X0 + cv_round(iM[0] * x * 1024)) >> 5;
                Y = (Y0 + cv_round(iM[3] * x * 1024)) >> 5;
            } else {
                const int xb = x - x % bw0, x1 = x - xb;
                const double X0 = iM[0] * xb + iM[1] * y + iM[2];
                const double Y0 = iM[3] * xb + iM[4] * y + iM[5];
                const double W0 = iM[6] * xb + iM[7] * y + iM[8];
                double W = W0 + iM[6] * x1;
                W = W ? 32. / W : 0;
                double fX = (X0 + iM[0] * x1) * W, fY = (Y0 + iM[3] * x1) * W;
                fX = fmax((double)INT32_MIN, fmin((double)INT32_MAX, fX));
                fY = fmax((double)INT32_MIN, fmin((double)INT32_MAX, fY));
                X = cv_round(fX);
                Y = cv_round(fY);
            }
            const int ix = sat_short(X >> 5), iy = sat_short(Y >> 5), fx = X & 31, fy = Y & 31;
            int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
            if (fx == 0 && fy == 0) { w00 = 32767; w11 = 1; }
            uint8_t *o = dst + ((long)y * dw + x) * c;
            for (int ch = 0; ch < c; ch++) {
                const int v = tap(src, sh, sw, c, ix, iy, ch) * w00 + tap(src, sh, sw, c, ix + 1, iy, ch) * w01 +
                              tap(src, sh, sw, c, ix, iy + 1, ch) * w10 + tap(src, sh, sw, c, ix + 1, iy + 1, ch) * w11;
                const int r = (v + 16384) >> 15;
                o[ch] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
            }
        }
}

void orc_warp(const uint8_t *src, int sh, int sw, int c, const double *M, int perspective,
              uint8_t *dst, int dh, int dw, int arith) {
    double iMd[9];
    float m[9];
    if (perspective) invert_3x3(M, iMd);
    else { invert_affine(M, iMd); iMd[6] = 0; iMd[7] = 0; iMd[8] = 1; }
    if (arith == 1) {
        warp_classic(src, sh, sw, c, iMd, perspective, dst, dh, dw);
        return;
    }
    for (int i = 0; i < 9; i++) m[i] = (float)iMd[i];
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            float fx = (float)x, fy = (float)y;
            float sx = fmaf(m[0], fx, fmaf(m[1], fy, m[2]));
            float sy = fmaf(m[3], fx, fmaf(m[4], fy, m[5]));
            if (perspective) {
                float w = fmaf(m[6], fx, fmaf(m[7], fy, m[8]));
                sx = sx / w;
                sy = sy / w;
            }
            uint8_t *o = dst + ((long)y * dw + x) * c;
            /* NaN / huge coordinates sample the constant border */
            if (!(sx > -2.f && sx < (float)sw + 1.f && sy > -2.f && sy < (float)sh + 1.f)) {
                for (int ch = 0; ch < c; ch++) o[ch] = 0;
                continue;
            }
            float flx = floorf(sx), fly = floorf(sy);
            int ix = (int)flx, iy = (int)fly;
            float a = sx - flx, b = sy - fly;
            for (int ch = 0; ch < c; ch++) {
                float p00 = px(src, sh, sw, c, ix, iy, ch), p01 = px(src, sh, sw, c, ix + 1, iy, ch);
                float p10 = px(src, sh, sw, c, ix, iy + 1, ch), p11 = px(src, sh, sw, c, ix + 1, iy + 1, ch);
                float t0 = fmaf(a, p01 - p00, p00);
                float t1 = fmaf(a, p11 - p10, p10);
                float v = fmaf(b, t1 - t0, t0);
                float r = rintf(v); /* round half to even (default rounding mode), then saturate */
                o[ch] = (uint8_t)(r < 0.f ? 0 : (r > 255.f ? 255 : (int)r));
            }
        }
}

/* The same warp on a single-channel uint16 image (the opt-in depth-preserving mode: what stitcher.py:143-149 computes when
 * the padded canvases take the frame's dtype instead of utils/proc.py:283's uint8).  [OCV-K] OpenCV 4.11's linear kernels serve
 * CV_16U with the same float32 arithmetic; round half to even, saturate to 65535.  arith 0 only. */
void orc_warp16(const uint16_t *src, int sh, int sw, const double *M, int perspective, uint16_t *dst, int dh, int dw) {
    double iMd[9];
    float m[9];
    if (perspective) invert_3x3(M, iMd);
    else { invert_affine(M, iMd); iMd[6] = 0; iMd[7] = 0; iMd[8] = 1; }
    for (int i = 0; i < 9; i++) m[i] = (float)iMd[i];
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            float fx = (float)x, fy = (float)y;
            float sx = fmaf(m[0], fx, fmaf(m[1], fy, m[2]));
            float sy = fmaf(m[3], fx, fmaf(m[4], fy, m[5]));
            if (perspective) {
                float w = fmaf(m[6], fx, fmaf(m[7], fy, m[8]));
                sx = sx / w;
                sy = sy / w;
            }
            uint16_t *o = dst + (long)y * dw + x;
            if (!(sx > -2.f && sx < (float)sw + 1.f && sy > -2.f && sy < (float)sh + 1.f)) { *o = 0; continue; }
            float flx = floorf(sx), fly = floorf(sy);
            int ix = (int)flx, iy = (int)fly;
            float a = sx - flx, b = sy - fly;
#define PX16(X, Y) (((unsigned)(X) >= (unsigned)sw || (unsigned)(Y) >= (unsigned)sh) ? 0.f : (float)src[(long)(Y) * sw + (X)])
            float p00 = PX16(ix, iy), p01 = PX16(ix + 1, iy), p10 = PX16(ix, iy + 1), p11 = PX16(ix + 1, iy + 1);
#undef PX16
            float t0 = fmaf(a, p01 - p00, p00);
            float t1 = fmaf(a, p11 - p10, p10);
            float r = rintf(fmaf(b, t1 - t0, t0));
            *o = (uint16_t)(r < 0.f ? 0 : (r > 65535.f ? 65535 : (int)r));
        }
}

/* proc.py:48-82 seamless_merge on two 2-D uint16 arrays (same mode): :58-59 mask = v > 10 (cv2.threshold on the array itself),
 * :70-71 float64 products stored into uint16 (truncation), :79 uint16 + uint16 wraps modulo 65536. */
void orc_alpha_blend16(const uint16_t *warped, const uint16_t *ref, int h, int w, double contrib, uint16_t *out) {
    const double one_minus = 1 - contrib;
    for (long i = 0; i < (long)h * w; i++) {
        uint16_t a = warped[i], b = ref[i];
        if (a > 10 && b > 10) {
            b = (uint16_t)(contrib * (double)b);
            a = (uint16_t)(one_minus * (double)a);
        }
        out[i] = (uint16_t)(a + b);
    }
}

/* ---- blenders ---------------------------------------------------------------------------------- */
/* proc.py:48-82 seamless_merge(image1=warped, image2=ref, ref_image_contrib).
 *   :50-55 gray (3-D only)     :58-59 mask = gray > 10     :66 overlap = mask1 & mask2
 *   :70 image2[overlap] = contrib * image2[overlap]        (float64 product stored into uint8 = truncation)
 *   :71 image1[overlap] = (1-contrib) * image1[overlap]
 *   :79 merged = image1 + image2                           (uint8 + uint8 wraps modulo 256)          */
void orc_alpha_blend(const uint8_t *warped, const uint8_t *ref, int h, int w, int c, double contrib,
                     uint8_t *out) {
    const double one_minus = 1 - contrib;
    for (long i = 0; i < (long)h * w; i++) {
        const uint8_t *p1 = warped + i * c, *p2 = ref + i * c;
        int g1 = c == 3 ? gray_of(p1) : p1[0];
        int g2 = c == 3 ? gray_of(p2) : p2[0];
        int overlap = (g1 > 10) && (g2 > 10);
        for (int ch = 0; ch < c; ch++) {
            uint8_t a = p1[ch], b = p2[ch];
            if (overlap) {
                b = (uint8_t)(contrib * (double)b);
                a = (uint8_t)(one_minus * (double)a);
            }
            out[i * c + ch] = (uint8_t)(a + b);
        }
    }
}

/* proc.py:192-254 seamless_gradient_merge(warped_img1, img2, feather_pixels=10); img2 (ref) is modified
 * in place inside bbox(warped) +- 50 px and returned. */
void orc_gradient_blend(const uint8_t *warped, uint8_t *ref, int h, int w, int c, int feather) {
    int32_t bb[4];
    if (!orc_bbox_nonzero(warped, h, w, c, bb)) return; /* :195, :250-251 */
    /* :199-205 ROI = corners +- 50 clipped to the canvas */
    int rx0 = bb[0] - 50 > 0 ? bb[0] - 50 : 0, ry0 = bb[1] - 50 > 0 ? bb[1] - 50 : 0;
    int rx1 = bb[2] + 50 < w ? bb[2] + 50 : w, ry1 = bb[3] + 50 < h ? bb[3] + 50 : h;
    int rw = rx1 - rx0, rh = ry1 - ry0;
    double *mask = (double *)malloc(sizeof(double) * (size_t)rw * rh * c);
#define WP(y, x, ch) warped[((long)((y) + ry0) * w + (x) + rx0) * c + (ch)]
#define RP(y, x, ch) ref[((long)((y) + ry0) * w + (x) + rx0) * c + (ch)]
#define MK(y, x, ch) mask[((long)(y) * rw + (x)) * c + (ch)]
    /* :207-208 mask = float64(roi1 > 0), element-wise per channel */
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++)
            for (int ch = 0; ch < c; ch++) MK(y, x, ch) = WP(y, x, ch) > 0 ? 1.0 : 0.0;
    /* :211-227 horizontal pass: edges found on channel 0 */
    for (int y = 0; y < rh; y++) {
        int left = -1, right = -1;
        for (int x = 0; x < rw; x++)
            if (MK(y, x, 0) > 0) { if (left < 0) left = x; right = x; }
        if (left < 0) continue;
        int xe = left + feather < rw ? left + feather : rw;
        for (int x = left; x < xe; x++) {
            double alpha = (double)(x - left) / (double)feather;
            for (int ch = 0; ch < c; ch++) MK(y, x, ch) *= alpha;
        }
        int xs = right - feather > 0 ? right - feather : 0;
        for (int x = xs; x <= right; x++) {
            double alpha = (double)(right - x) / (double)feather;
            for (int ch = 0; ch < c; ch++) MK(y, x, ch) *= alpha;
        }
    }
    /* :230-246 vertical pass on the already-modified mask */
    for (int x = 0; x < rw; x++) {
        int top = -1, bottom = -1;
        for (int y = 0; y < rh; y++)
            if (MK(y, x, 0) > 0) { if (top < 0) top = y; bottom = y; }
        if (top < 0) continue;
        int ye = top + feather < rh ? top + feather : rh;
        for (int y = top; y < ye; y++) {
            double alpha = (double)(y - top) / (double)feather;
            for (int ch = 0; ch < c; ch++) MK(y, x, ch) *= alpha;
        }
        int ys = bottom - feather > 0 ? bottom - feather : 0;
        for (int y = ys; y <= bottom; y++) {
            double alpha = (double)(bottom - y) / (double)feather;
            for (int ch = 0; ch < c; ch++) MK(y, x, ch) *= alpha;
        }
    }
    /* :249, :253 result = roi1*mask + roi2*(1-mask) in float64, astype(uint8) = truncation */
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++)
            for (int ch = 0; ch < c; ch++) {
                double m = MK(y, x, ch);
                double r = (double)WP(y, x, ch) * m + (double)RP(y, x, ch) * (1 - m);
                RP(y, x, ch) = (uint8_t)r;
            }
#undef WP
#undef RP
#undef MK
    free(mask);
}

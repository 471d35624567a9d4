// Batched RANSAC for gfx950: replaces Stitcher.get_transform (reference stitcher.py:135-141:
// cv2.estimateAffinePartial2D(src, dst) for type 0, cv2.findHomography(src, dst, cv2.RANSAC, 5.0)
// for type 1).
//
// The sequential OpenCV loop (sample -> minimal solve -> count inliers -> adaptive stop) is split so
// that everything data-parallel runs on the GPU while the SELECTION replays the loop in order (same
// subsets, same "strictly better wins", same adaptive stop).  The arithmetic follows cv's to tolerance, not bit
// for bit: the 4-point hypotheses are solved by Gauss-Jordan here (cv: eigenvector of LtL), so a point sitting on
// the float32 threshold can change an inlier count.  tests/test_gpu_ransac.py gates the models to 1e-6 relative and
// the inlier counts on the oracle's hypotheses; against cv2 itself this stage is "parity unpinned" (DESIGN.md 2):
//   host   isx_ransac_sample      cv::RNG-faithful subset list (depends on the points only, never on
//                                 the models, so all <=2000 subsets can be drawn up front)
//   device isx_ransac_hypotheses  one lane per subset: 4-point DLT (normalised, null vector by
//                                 Gauss-Jordan with full pivoting) or the 2-point similarity closed form
//   device isx_ransac_score       one wavefront per hypothesis: float32 reprojection error exactly as
//                                 cv's computeError, inlier count per hypothesis
//   host   isx_ransac_select      scans the counts in order, replaying cv's "strictly better wins" and
//                                 RANSACUpdateNumIters adaptive stop
//   device isx_ransac_refine      one wavefront: inlier mask of the winner, all-inlier DLT refit
//                                 (homography) and <=10 Levenberg-Marquardt iterations in float64
#include "isx_common.h"
#include <float.h>
#include <math.h>

#define MODEL_H ISX_MODEL_HOMOGRAPHY

// ------------------------------------------------------------------------------------------------
// host: sampler + selection
// ------------------------------------------------------------------------------------------------
namespace {
struct CvRng { // cv::RNG: multiply-with-carry, state = (uint32)state*4164903690 + (state>>32)
    uint64_t s;
    unsigned next() {
        s = (uint64_t)(unsigned)s * 4164903690U + (unsigned)(s >> 32);
        return (unsigned)s;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// cv haveCollinearPoints: only the last point is tested against lines through pairs of earlier ones
bool last_point_collinear(const float *p, int count) {
    const int i = count - 1;
    for (int j = 0; j < i; j++) {
        const double dx1 = p[2 * j] - p[2 * i], dy1 = p[2 * j + 1] - p[2 * i + 1];
        for (int k = 0; k < j; k++) {
            const double dx2 = p[2 * k] - p[2 * i], dy2 = p[2 * k + 1] - p[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return true;
        }
    }
    return false;
}

double tri_det(const float *p, int a, int b, int c) { // det [[xa ya 1][xb yb 1][xc yc 1]]
    const double xa = p[2 * a], ya = p[2 * a + 1], xb = p[2 * b], yb = p[2 * b + 1], xc = p[2 * c], yc = p[2 * c + 1];
    return xa * (yb * 1. - 1. * yc) - ya * (xb * 1. - 1. * xc) + 1. * (xb * yc - yb * xc);
}

bool subset_ok(int model, const float *a, const float *b, int count) {
    if (last_point_collinear(a, count) || last_point_collinear(b, count)) return false;
    if (model == MODEL_H && count == 4) {
        static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
        int negative = 0;
        for (int i = 0; i < 4; i++)
            negative += tri_det(a, tt[i][0], tt[i][1], tt[i][2]) * tri_det(b, tt[i][0], tt[i][1], tt[i][2]) < 0;
        if (negative != 0 && negative != 4) return false;
    }
    return true;
}
} // namespace

extern "C" int isx_ransac_sample(int model, const float *src_host, const float *dst_host, int n, int max_iters,
                                 int32_t *subsets_out_host) {
    ISX_REQUIRE(model == ISX_MODEL_AFFINE_PARTIAL || model == MODEL_H, "bad model %d", model);
    ISX_REQUIRE(n >= 0 && max_iters >= 0, "negative size");
    const int mp = model == MODEL_H ? 4 : 2;
    if (n < mp || max_iters == 0) return 0;
    ISX_REQUIRE(src_host && dst_host && subsets_out_host, "null host pointer");
    if (n == mp) {
        for (int i = 0; i < mp; i++) subsets_out_host[i] = i;
        return 1;
    }
    CvRng rng{~(uint64_t)0};
    float a[8], b[8];
    int idx[4];
    for (int it = 0; it < max_iters; it++) {
        int attempt = 0;
        for (; attempt < 10000; attempt++) {
            for (int i = 0; i < mp; i++) {
                int v;
                bool dup;
                do {
                    v = rng.uniform(0, n);
                    dup = false;
                    for (int k = 0; k < i; k++) dup |= idx[k] == v;
                } while (dup);
                idx[i] = v;
                a[2 * i] = src_host[2 * v];
                a[2 * i + 1] = src_host[2 * v + 1];
                b[2 * i] = dst_host[2 * v];
                b[2 * i + 1] = dst_host[2 * v + 1];
            }
            if (subset_ok(model, a, b, mp)) break;
        }
        if (attempt == 10000) return it; // cv: `break` (or failure when it == 0)
        for (int i = 0; i < mp; i++) subsets_out_host[it * mp + i] = idx[i];
    }
    return max_iters;
}

static int update_num_iters(double p, double ep, int mp, int max_iters) { // cv RANSACUpdateNumIters
    p = fmax(p, 0.);
    p = fmin(p, 1.);
    ep = fmax(ep, 0.);
    ep = fmin(ep, 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, mp);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

extern "C" int isx_ransac_select(int model, const int32_t *counts_host, const int32_t *valid_host, int n_hyp,
                                 int n_points, double confidence, int max_iters, int32_t *best_out,
                                 int32_t *iters_out) {
    ISX_REQUIRE(counts_host && best_out && iters_out, "null host pointer");
    ISX_REQUIRE(confidence > 0 && confidence < 1, "confidence must be in (0,1)");
    const int mp = model == MODEL_H ? 4 : 2;
    int niters = max_iters > 1 ? max_iters : 1, best = -1, max_good = 0, it = 0;
    for (; it < niters && it < n_hyp; it++) {
        if (valid_host && !valid_host[it]) continue;
        const int good = counts_host[it];
        if (good > (max_good > mp - 1 ? max_good : mp - 1)) {
            max_good = good;
            best = it;
            niters = update_num_iters(confidence, (double)(n_points - good) / n_points, mp, niters);
        }
    }
    *best_out = best;
    *iters_out = it;
    return ISX_OK;
}

// ------------------------------------------------------------------------------------------------
// device: minimal solves
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void affine_partial_2pt(const float *from, const float *to, double *M) {
    const double x1 = from[0], y1 = from[1], x2 = from[2], y2 = from[3];
    const double X1 = to[0], Y1 = to[1], X2 = to[2], Y2 = to[3];
    const double d = 1. / ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
    const double S0 = d * ((X1 - X2) * (x1 - x2) + (Y1 - Y2) * (y1 - y2));
    const double S1 = d * ((Y1 - Y2) * (x1 - x2) - (X1 - X2) * (y1 - y2));
    const double S2 = d * ((Y1 - Y2) * (x1 * y2 - x2 * y1) - (X1 * y2 - X2 * y1) * (y1 - y2) - (X1 * x2 - X2 * x1) * (x1 - x2));
    const double S3 = d * (-(X1 - X2) * (x1 * y2 - x2 * y1) - (Y1 * x2 - Y2 * x1) * (x1 - x2) - (Y1 * y2 - Y2 * y1) * (y1 - y2));
    M[0] = S0; M[1] = -S1; M[2] = S2; M[3] = S1; M[4] = S0; M[5] = S3;
}

struct Norm2 { // cv's DLT normalisation: centroid + mean absolute deviation per axis
    double cMx, cMy, cmx, cmy, sMx, sMy, smx, smy;
};

__device__ __forceinline__ void denormalise(const double *h, const Norm2 &N, double *H) {
    // H = invHnorm * H0 * Hnorm2, then scaled by 1/H[8]
    const double inv[9] = {1. / N.smx, 0, N.cmx, 0, 1. / N.smy, N.cmy, 0, 0, 1};
    const double nr2[9] = {N.sMx, 0, -N.cMx * N.sMx, 0, N.sMy, -N.cMy * N.sMy, 0, 0, 1};
    double t[9], f[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) t[r * 3 + c] = inv[r * 3] * h[c] + inv[r * 3 + 1] * h[3 + c] + inv[r * 3 + 2] * h[6 + c];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) f[r * 3 + c] = t[r * 3] * nr2[c] + t[r * 3 + 1] * nr2[3 + c] + t[r * 3 + 2] * nr2[6 + c];
    const double sc = 1. / f[8];
#pragma unroll
    for (int i = 0; i < 9; i++) H[i] = f[i] * sc;
}

// one lane per hypothesis; the 8x9 DLT matrix of each lane lives in LDS as A[(r*9+c)*64 + lane]
__global__ void __launch_bounds__(64) ransac_hyp_kernel(int model, const float *__restrict__ src,
                                                        const float *__restrict__ dst,
                                                        const int32_t *__restrict__ subsets, int n_hyp,
                                                        double *__restrict__ models, int32_t *__restrict__ valid) {
    __shared__ double A[72 * 64];
    const int lane = threadIdx.x;
    const int t = blockIdx.x * 64 + lane;
    if (t >= n_hyp) return;
    if (model != MODEL_H) {
        float a[4], b[4];
        for (int i = 0; i < 2; i++) {
            const int v = subsets[t * 2 + i];
            a[2 * i] = src[2 * v]; a[2 * i + 1] = src[2 * v + 1];
            b[2 * i] = dst[2 * v]; b[2 * i + 1] = dst[2 * v + 1];
        }
        double M[6];
        affine_partial_2pt(a, b, M);
        for (int i = 0; i < 6; i++) models[(int64_t)t * 6 + i] = M[i];
        valid[t] = 1;
        return;
    }
    double Mx[4], My[4], mx[4], my[4];
    Norm2 N = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = subsets[t * 4 + i];
        Mx[i] = src[2 * v]; My[i] = src[2 * v + 1];
        mx[i] = dst[2 * v]; my[i] = dst[2 * v + 1];
        N.cMx += Mx[i]; N.cMy += My[i]; N.cmx += mx[i]; N.cmy += my[i];
    }
    N.cMx /= 4; N.cMy /= 4; N.cmx /= 4; N.cmy /= 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        N.sMx += fabs(Mx[i] - N.cMx); N.sMy += fabs(My[i] - N.cMy);
        N.smx += fabs(mx[i] - N.cmx); N.smy += fabs(my[i] - N.cmy);
    }
    if (fabs(N.smx) < DBL_EPSILON || fabs(N.smy) < DBL_EPSILON || fabs(N.sMx) < DBL_EPSILON || fabs(N.sMy) < DBL_EPSILON) {
        valid[t] = 0;
        for (int i = 0; i < 9; i++) models[(int64_t)t * 9 + i] = 0;
        return;
    }
    N.smx = 4 / N.smx; N.smy = 4 / N.smy; N.sMx = 4 / N.sMx; N.sMy = 4 / N.sMy;
#define AT(r, c) A[((r) * 9 + (c)) * 64 + lane]
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double x = (mx[i] - N.cmx) * N.smx, y = (my[i] - N.cmy) * N.smy;
        const double X = (Mx[i] - N.cMx) * N.sMx, Y = (My[i] - N.cMy) * N.sMy;
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
#pragma unroll
        for (int c = 0; c < 9; c++) {
            AT(2 * i, c) = Lx[c];
            AT(2 * i + 1, c) = Ly[c];
        }
    }
    // Gauss-Jordan with full pivoting: 8 pivots, the one column left over is the free variable
    unsigned rows_used = 0, cols_used = 0;
    int prow[8], pcol[8];
    bool ok = true;
    for (int s = 0; s < 8; s++) {
        double best = -1;
        int br = 0, bc = 0;
        for (int r = 0; r < 8; r++) {
            if (rows_used >> r & 1) continue;
            for (int c = 0; c < 9; c++) {
                if (cols_used >> c & 1) continue;
                const double v = fabs(AT(r, c));
                if (v > best) { best = v; br = r; bc = c; }
            }
        }
        if (!(best > 1e-280)) { ok = false; break; }
        rows_used |= 1u << br;
        cols_used |= 1u << bc;
        prow[s] = br;
        pcol[s] = bc;
        const double ip = 1. / AT(br, bc);
        for (int c = 0; c < 9; c++) AT(br, c) *= ip;
        for (int r = 0; r < 8; r++) {
            if (r == br) continue;
            const double f = AT(r, bc);
            if (f != 0.)
                for (int c = 0; c < 9; c++) AT(r, c) -= f * AT(br, c);
        }
    }
    double h[9];
    if (ok) {
        int fc = 0;
        for (int c = 0; c < 9; c++)
            if (!(cols_used >> c & 1)) fc = c;
        for (int c = 0; c < 9; c++) h[c] = 0;
        h[fc] = 1;
        for (int s = 0; s < 8; s++) {
            const double v = -AT(prow[s], fc);
            // static-index store (runtime-indexed register arrays would spill)
#pragma unroll
            for (int c = 0; c < 9; c++)
                if (c == pcol[s]) h[c] = v;
        }
    }
#undef AT
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (ok) denormalise(h, N, H);
    for (int i = 0; i < 9; i++) models[(int64_t)t * 9 + i] = H[i];
    valid[t] = ok ? 1 : 0;
}

extern "C" int isx_ransac_hypotheses(int model, const float *src_dev, const float *dst_dev, int n,
                                     const int32_t *subsets_dev, int n_hyp, double *models_out_dev,
                                     int32_t *valid_out_dev, void *stream) {
    ISX_REQUIRE(model == ISX_MODEL_AFFINE_PARTIAL || model == MODEL_H, "bad model %d", model);
    ISX_REQUIRE(n_hyp >= 0 && n >= 0, "negative size");
    if (n_hyp == 0) return ISX_OK;
    ISX_REQUIRE(src_dev && dst_dev && subsets_dev && models_out_dev && valid_out_dev, "null device pointer");
    hipLaunchKernelGGL(ransac_hyp_kernel, dim3((n_hyp + 63) / 64), dim3(64), 0, (hipStream_t)stream, model, src_dev,
                       dst_dev, subsets_dev, n_hyp, models_out_dev, valid_out_dev);
    return isx_launch_status("ransac_hyp_kernel");
}

// ------------------------------------------------------------------------------------------------
// device: scoring.  float32 arithmetic in cv's order, no FMA contraction (-ffp-contract=off).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float reproj_err_f32(bool homog, const float *F, float x, float y, float u, float v) {
    if (homog) {
        const float ww = 1.f / (F[6] * x + F[7] * y + 1.f);
        const float dx = (F[0] * x + F[1] * y + F[2]) * ww - u;
        const float dy = (F[3] * x + F[4] * y + F[5]) * ww - v;
        return dx * dx + dy * dy;
    }
    const float a = F[0] * x + F[1] * y + F[2] - u;
    const float b = F[3] * x + F[4] * y + F[5] - v;
    return a * a + b * b;
}

__global__ void __launch_bounds__(256) ransac_score_kernel(int model, const double *__restrict__ models, int n_hyp,
                                                           const int32_t *__restrict__ valid,
                                                           const float2 *__restrict__ src,
                                                           const float2 *__restrict__ dst, int n, float thr2,
                                                           int32_t *__restrict__ counts) {
    const int hyp = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (hyp >= n_hyp) return;
    const bool homog = model == MODEL_H;
    const int msz = homog ? 9 : 6;
    float F[8];
#pragma unroll
    for (int i = 0; i < 8; i++) F[i] = (i < msz) ? (float)models[(int64_t)hyp * msz + i] : 0.f;
    int cnt = 0;
    if (!valid || valid[hyp]) {
        for (int i = lane; i < n; i += 64) {
            const float2 s = src[i], d = dst[i];
            cnt += reproj_err_f32(homog, F, s.x, s.y, d.x, d.y) <= thr2;
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) cnt += __shfl_xor(cnt, m);
    if (lane == 0) counts[hyp] = cnt;
}

extern "C" int isx_ransac_score(int model, const double *models_dev, int n_hyp, const int32_t *valid_dev,
                                const float *src_dev, const float *dst_dev, int n, double thr,
                                int32_t *counts_out_dev, void *stream) {
    ISX_REQUIRE(model == ISX_MODEL_AFFINE_PARTIAL || model == MODEL_H, "bad model %d", model);
    ISX_REQUIRE(n_hyp >= 0 && n >= 0, "negative size");
    if (n_hyp == 0) return ISX_OK;
    ISX_REQUIRE(models_dev && counts_out_dev && (n == 0 || (src_dev && dst_dev)), "null device pointer");
    hipLaunchKernelGGL(ransac_score_kernel, dim3((n_hyp + 3) / 4), dim3(256), 0, (hipStream_t)stream, model,
                       models_dev, n_hyp, valid_dev, (const float2 *)src_dev, (const float2 *)dst_dev, n,
                       (float)(thr * thr), counts_out_dev);
    return isx_launch_status("ransac_score_kernel");
}

// ------------------------------------------------------------------------------------------------
// device: winner's inlier mask + refit + Levenberg-Marquardt, one wavefront, float64
// ------------------------------------------------------------------------------------------------
// Wave-wide sum / max of a double, every lane gets the result.  The value tree is that of the xor butterfly
// (v += shfl_xor(v, m), m = 1 .. 32) -- ((v0 + v1) + (v2 + v3)) + ... -- so results are bit-identical to it, but the
// four in-row steps are DPP moves (quad permutes, then half-row and row mirrors: after the first two steps all lanes of
// a quad hold the same value, so any lane of the partner group will do) and the two cross-row steps read the four row
// sums with v_readlane: no LDS crossbar round trip (~100 cycles each, twelve per reduction, and the LM step of
// ransac_refine_kernel does 15 to 45 reductions per iteration).  Needs all 64 lanes active.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wsum(double v) {
    v += dpp_f64<0xB1>(v);  // quad_perm [1,0,3,2]  (lane ^ 1)
    v += dpp_f64<0x4E>(v);  // quad_perm [2,3,0,1]  (lane ^ 2)
    v += dpp_f64<0x141>(v); // row_half_mirror      (a lane of the other quad of the 8-group)
    v += dpp_f64<0x140>(v); // row_mirror           (a lane of the other 8-group of the row)
    const double s0 = readlane_f64(v, 0), s1 = readlane_f64(v, 16), s2 = readlane_f64(v, 32), s3 = readlane_f64(v, 48);
    return (s0 + s1) + (s2 + s3);
}
__device__ __forceinline__ double wmax(double v) {
    v = fmax(v, dpp_f64<0xB1>(v));
    v = fmax(v, dpp_f64<0x4E>(v));
    v = fmax(v, dpp_f64<0x141>(v));
    v = fmax(v, dpp_f64<0x140>(v));
    const double s0 = readlane_f64(v, 0), s1 = readlane_f64(v, 16), s2 = readlane_f64(v, 32), s3 = readlane_f64(v, 48);
    return fmax(fmax(s0, s1), fmax(s2, s3));
}

// symmetric n x n eigen-decomposition in LDS by cyclic Jacobi; lanes 0..n-1 each own one index.
// On return W[i] = eigenvalues, V[i*n + :] = eigenvectors (rows), unsorted.
__device__ __forceinline__ void jacobi_lds(double *A, double *V, double *W, int n, int lane) {
    if (lane < n)
        for (int j = 0; j < n; j++) V[lane * n + j] = lane == j ? 1. : 0.;
    __syncthreads();
    for (int sweep = 0; sweep < 40; sweep++) {
        double off = 0, dg = 0;
        if (lane < n)
            for (int j = 0; j < n; j++) {
                const double v = A[lane * n + j];
                if (j == lane) dg += v * v; else off += v * v;
            }
        off = wsum(off);
        dg = wsum(dg);
        if (!(off > 1e-30 * dg)) break; // wave-uniform
        for (int k = 0; k < n - 1; k++)
            for (int l = k + 1; l < n; l++) {
                const double p = A[k * n + l];
                if (fabs(p) > 1e-300) {
                    const double y = (A[l * n + l] - A[k * n + k]) * 0.5;
                    double t = fabs(y) + hypot(p, y);
                    double s = hypot(p, t);
                    const double c = t / s;
                    s = p / s;
                    if (y < 0) s = -s;
                    __syncthreads();
                    // A <- R^T A R with the rotation in the (k,l) plane; V rows k,l rotate alike
                    if (lane < n) {
                        const double aik = A[lane * n + k], ail = A[lane * n + l];
                        A[lane * n + k] = aik * c - ail * s;
                        A[lane * n + l] = aik * s + ail * c;
                    }
                    __syncthreads();
                    if (lane < n) {
                        const double aki = A[k * n + lane], ali = A[l * n + lane];
                        A[k * n + lane] = aki * c - ali * s;
                        A[l * n + lane] = aki * s + ali * c;
                        const double vk = V[k * n + lane], vl = V[l * n + lane];
                        V[k * n + lane] = vk * c - vl * s;
                        V[l * n + lane] = vk * s + vl * c;
                    }
                    __syncthreads();
                }
            }
    }
    if (lane < n) W[lane] = A[lane * n + lane];
    __syncthreads();
}

// Eigenvector of the SMALLEST eigenvalue of the symmetric positive semi-definite 9 x 9 matrix given by its upper triangle
// (acc[u], rows j <= k in order), by inverse iteration on an LDL^T factorisation held in registers: every index is a compile-time
// constant, every lane computes the same thing (no LDS, no barrier).  The DLT's LtL has one eigenvalue near zero and the next
// one orders of magnitude above it, so the iteration settles in two or three steps; it returns false when 24 steps do not
// (two comparably small eigenvalues: a degenerate point set) and the caller falls back to the Jacobi sweep, which the refit
// used for every frame before (cyclic Jacobi in LDS: ~70 us of three barriers and two hypot per rotation).
__device__ __forceinline__ bool smallest_eigvec9(const double (&acc)[45], double (&x)[9]) {
    double A[9][9];
    {
        int u = 0;
#pragma unroll
        for (int j = 0; j < 9; j++)
#pragma unroll
            for (int k = j; k < 9; k++) A[k][j] = acc[u++]; // lower triangle
    }
    double dmax = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) dmax = fmax(dmax, A[j][j]);
    const double floor_d = dmax * 1e-18 + 1e-300;
    double L[9][9], LD[9][9], invD[9]; // L unit lower triangular, LD[i][k] = L[i][k] * D[k]
#pragma unroll
    for (int j = 0; j < 9; j++) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= L[j][k] * LD[j][k];
        if (!(d > floor_d)) d = floor_d; // (the near-zero eigenvalue shows up as the last pivot; rounding may push it below 0)
        invD[j] = 1. / d;
#pragma unroll
        for (int i = j + 1; i < 9; i++) {
            double v = A[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) v -= L[i][k] * LD[j][k];
            LD[i][j] = v;
            L[i][j] = v * invD[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 9; i++) x[i] = 1. / 3. + 0.01 * i;
    bool settled = false;
    for (int it = 0; it < 24; it++) {
        double y[9];
#pragma unroll
        for (int i = 0; i < 9; i++) { // L y = x
            double v = x[i];
#pragma unroll
            for (int k = 0; k < i; k++) v -= L[i][k] * y[k];
            y[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 9; i++) y[i] *= invD[i];
#pragma unroll
        for (int i = 8; i >= 0; i--) { // L^T w = y (in place)
            double v = y[i];
#pragma unroll
            for (int k = i + 1; k < 9; k++) v -= L[k][i] * y[k];
            y[i] = v;
        }
        double nn = 0, dot = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) nn += y[i] * y[i], dot += y[i] * x[i];
        const double sc = (dot < 0 ? -1. : 1.) / sqrt(nn);
        double diff = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const double v = y[i] * sc;
            diff = fmax(diff, fabs(v - x[i]));
            x[i] = v;
        }
        if (!(nn > 0) || nn != nn) return false;
        if (settled) return true; // one step past the first that moved by less than 1e-13
        settled = diff < 1e-13;
    }
    return false;
}

// Gaussian elimination with partial pivoting on lane 0 (n <= 8), Ab is n x (n+1) in LDS; x out
__device__ __forceinline__ void solve_ge(double *Ab, double *x, int n) {
    const int w = n + 1;
    for (int s = 0; s < n; s++) {
        int piv = s;
        double best = fabs(Ab[s * w + s]);
        for (int r = s + 1; r < n; r++)
            if (fabs(Ab[r * w + s]) > best) { best = fabs(Ab[r * w + s]); piv = r; }
        if (piv != s)
            for (int c = 0; c < w; c++) { const double t = Ab[s * w + c]; Ab[s * w + c] = Ab[piv * w + c]; Ab[piv * w + c] = t; }
        const double d = Ab[s * w + s];
        if (fabs(d) < 1e-300) continue;
        for (int r = s + 1; r < n; r++) {
            const double f = Ab[r * w + s] / d;
            for (int c = s; c < w; c++) Ab[r * w + c] -= f * Ab[s * w + c];
        }
    }
    for (int s = n - 1; s >= 0; s--) {
        double v = Ab[s * w + n];
        for (int c = s + 1; c < n; c++) v -= Ab[s * w + c] * x[c];
        const double d = Ab[s * w + s];
        x[s] = fabs(d) < 1e-300 ? 0. : v / d;
    }
}

template <int NP>
__device__ __forceinline__ void lm_residual(const double *h, double Mx, double My, double u, double v, double &r0,
                                            double &r1, double *J0, double *J1) {
    if (NP == 8) {
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        r0 = xi - u;
        r1 = yi - v;
        if (J0) {
            J0[0] = Mx * ww; J0[1] = My * ww; J0[2] = ww; J0[3] = 0; J0[4] = 0; J0[5] = 0;
            J0[6] = -Mx * ww * xi; J0[7] = -My * ww * xi;
            J1[0] = 0; J1[1] = 0; J1[2] = 0; J1[3] = Mx * ww; J1[4] = My * ww; J1[5] = ww;
            J1[6] = -Mx * ww * yi; J1[7] = -My * ww * yi;
        }
    } else {
        r0 = h[0] * Mx - h[1] * My + h[2] - u;
        r1 = h[1] * Mx + h[0] * My + h[3] - v;
        if (J0) {
            J0[0] = Mx; J0[1] = -My; J0[2] = 1; J0[3] = 0;
            J1[0] = My; J1[1] = Mx; J1[2] = 0; J1[3] = 1;
        }
    }
}

// solve_ge with the augmented matrix in REGISTERS: every index is a compile-time constant (loops fully unrolled, the
// pivot row swap done as predicated element swaps), the arithmetic and its order are those of solve_ge.  One lane walking
// an LDS-resident matrix pays ~100 cycles per dependent access, which made the LM step of ransac_refine_kernel
// latency-bound on this routine (4.5 us per 4 x 4 solve, 25 us per 8 x 8).
template <int N>
__device__ __forceinline__ void solve_ge_reg(double (&Ab)[N][N + 1], double (&x)[N]) {
#pragma unroll
    for (int s = 0; s < N; s++) {
        int piv = s;
        double best = fabs(Ab[s][s]);
#pragma unroll
        for (int r = s + 1; r < N; r++)
            if (fabs(Ab[r][s]) > best) { best = fabs(Ab[r][s]); piv = r; }
#pragma unroll
        for (int r = s + 1; r < N; r++) {
            const bool sw = piv == r;
#pragma unroll
            for (int c = 0; c < N + 1; c++) {
                const double a = Ab[s][c], b = Ab[r][c];
                Ab[s][c] = sw ? b : a;
                Ab[r][c] = sw ? a : b;
            }
        }
        const double d = Ab[s][s];
        if (fabs(d) < 1e-300) continue;
#pragma unroll
        for (int r = s + 1; r < N; r++) {
            const double f = Ab[r][s] / d;
#pragma unroll
            for (int c = s; c < N + 1; c++) Ab[r][c] -= f * Ab[s][c];
        }
    }
#pragma unroll
    for (int s = N - 1; s >= 0; s--) {
        double v = Ab[s][N];
#pragma unroll
        for (int c = s + 1; c < N; c++) v -= Ab[s][c] * x[c];
        const double d = Ab[s][s];
        x[s] = fabs(d) < 1e-300 ? 0. : v / d;
    }
}

struct RefineShared {
    double A[81], V[81], W[9];
    double JtJ[64], Ab[72], v[8], d[8], x[8], xd[8], D[8];
    double S, Sd, lambda, lc;
    int proceed;
    double red[45 * 65], sum[48]; // wave_sums' transposed partials (rows padded to 65: lanes then read different banks) and totals
};

// Totals over the wave of N per-lane sums at once (N <= 45), left in sh.sum[0..N): every lane writes its N partials to LDS,
// lane t adds up column t.  Doing them one by one costs a DPP + v_readlane butterfly each (~35 instructions): the 45 sums of a
// homography's normal equations were ~3.5 us of every LM iteration, the 15 of the similarity's ~1 us.  The order of the
// additions is lane 0, 1, ... 63 (the butterfly's was a tree): same sums to rounding.
template <int N>
__device__ __forceinline__ void wave_sums(RefineShared &sh, const double (&acc)[N], int lane) {
    static_assert(N <= 45, "sh.red holds 45 rows");
#pragma unroll
    for (int i = 0; i < N; i++) sh.red[i * 65 + lane] = acc[i];
    __syncthreads();
    if (lane < N) {
        double t = 0;
#pragma unroll 8
        for (int l = 0; l < 64; l++) t += sh.red[lane * 65 + l];
        sh.sum[lane] = t;
    }
    __syncthreads();
}

// accumulate J^T J, J^T r and |r|^2 over the inliers at parameters sh.x
template <int NP>
__device__ __forceinline__ void lm_normal_equations(RefineShared &sh, const float2 *src, const float2 *dst, const uint8_t *mask, int n,
                                    int lane, double &rinf) {
    constexpr int NU = NP * (NP + 1) / 2;
    double acc[NU + NP + 1];
#pragma unroll
    for (int i = 0; i < NU + NP + 1; i++) acc[i] = 0;
    double h[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) h[i] = sh.x[i];
    double ri = 0;
    for (int i = lane; i < n; i += 64) {
        if (!mask[i]) continue;
        const float2 s = src[i], d = dst[i];
        double r0, r1, J0[NP], J1[NP];
        lm_residual<NP>(h, s.x, s.y, d.x, d.y, r0, r1, J0, J1);
        int u = 0;
#pragma unroll
        for (int a = 0; a < NP; a++) {
#pragma unroll
            for (int b = a; b < NP; b++) acc[u++] += J0[a] * J0[b] + J1[a] * J1[b];
        }
#pragma unroll
        for (int a = 0; a < NP; a++) acc[NU + a] += J0[a] * r0 + J1[a] * r1;
        acc[NU + NP] += r0 * r0 + r1 * r1;
        ri = fmax(ri, fmax(fabs(r0), fabs(r1)));
    }
    wave_sums<NU + NP + 1>(sh, acc, lane);
    rinf = wmax(ri);
    if (lane == 0) {
        int u = 0;
#pragma unroll
        for (int a = 0; a < NP; a++) {
#pragma unroll
            for (int b = a; b < NP; b++) {
                const double t = sh.sum[u];
                sh.JtJ[a * NP + b] = t;
                sh.JtJ[b * NP + a] = t;
                u++;
            }
        }
#pragma unroll
        for (int a = 0; a < NP; a++) sh.v[a] = sh.sum[NU + a];
        sh.S = sh.sum[NU + NP];
    }
    __syncthreads();
}

// LDL^T of a symmetric positive definite N x N matrix in registers (no pivoting, every index a compile-time constant) and the
// solves on it.  The LM step's matrix JtJ + lambda diag(JtJ) is positive definite whenever the points determine the model;
// solve_ge_reg pays ~1000 predicated element swaps for a pivot search this matrix does not need, and the lambda-reset branch
// can reuse ONE factorisation for its NP unit vectors (homography refinement 123 -> 89 us at 500 points).  Returns false when a
// pivot is not safely positive: the caller then takes solve_ge_reg.
template <int N>
__device__ __forceinline__ bool ldlt_reg(const double (&A)[N][N], double (&L)[N][N], double (&invD)[N]) {
    double LD[N][N];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; j++) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= L[j][k] * LD[j][k];
        ok = ok && d > A[j][j] * 1e-10; // (scale-free: a pivot that lost ten digits to cancellation means a near-singular matrix)
        invD[j] = 1. / d;
#pragma unroll
        for (int i = j + 1; i < N; i++) {
            double v = A[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) v -= L[i][k] * LD[j][k];
            LD[i][j] = v;
            L[i][j] = v * invD[j];
        }
    }
    return ok;
}
template <int N>
__device__ __forceinline__ void ldlt_solve_reg(const double (&L)[N][N], const double (&invD)[N], const double (&b)[N], double (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) v -= L[i][k] * x[k];
        x[i] = v;
    }
#pragma unroll
    for (int i = 0; i < N; i++) x[i] *= invD[i];
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        double v = x[i];
#pragma unroll
        for (int k = i + 1; k < N; k++) v -= L[k][i] * x[k];
        x[i] = v;
    }
}

template <int NP>
__device__ __forceinline__ double lm_cost(const double *hx, const float2 *src, const float2 *dst, const uint8_t *mask, int n, int lane) {
    double h[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) h[i] = hx[i];
    double s = 0;
    for (int i = lane; i < n; i += 64) {
        if (!mask[i]) continue;
        const float2 a = src[i], b = dst[i];
        double r0, r1;
        lm_residual<NP>(h, a.x, a.y, b.x, b.y, r0, r1, nullptr, nullptr);
        s += r0 * r0 + r1 * r1;
    }
    return wsum(s);
}

// cv::LMSolver::run (levmarq.cpp) with the refine callbacks of fundam.cpp / ptsetreg.cpp
template <int NP>
__device__ __forceinline__ void lm_run(RefineShared &sh, const float2 *src, const float2 *dst, const uint8_t *mask, int n, int lane,
                       int max_iters) {
    const double eps = FLT_EPSILON, Rlo = 0.25, Rhi = 0.75;
    double rinf;
    lm_normal_equations<NP>(sh, src, dst, mask, n, lane, rinf);
    if (lane == 0) {
        for (int i = 0; i < NP; i++) sh.D[i] = sh.JtJ[i * NP + i];
        sh.lambda = 1;
        sh.lc = 0.75;
    }
    __syncthreads();
    for (int iter = 0;;) {
        if (lane == 0) {
            double Ab[NP][NP + 1], dd[NP];
            const double lam = sh.lambda;
#pragma unroll
            for (int a = 0; a < NP; a++) {
#pragma unroll
                for (int b = 0; b < NP; b++) Ab[a][b] = sh.JtJ[a * NP + b] + (a == b ? lam * sh.D[a] : 0.);
                Ab[a][NP] = sh.v[a];
            }
            {
                double As[NP][NP], Lf[NP][NP], iD[NP], rhs[NP];
#pragma unroll
                for (int a = 0; a < NP; a++) {
                    rhs[a] = Ab[a][NP];
#pragma unroll
                    for (int b = 0; b < NP; b++) As[a][b] = Ab[a][b];
                }
#if defined(ISX_LM_DIAG) && ISX_LM_DIAG == 1 /* timing diagnostics: 1 = always Gaussian elimination, 2 = never */
                solve_ge_reg<NP>(Ab, dd);
#elif defined(ISX_LM_DIAG) && ISX_LM_DIAG == 2
                ldlt_reg<NP>(As, Lf, iD);
                ldlt_solve_reg<NP>(Lf, iD, rhs, dd);
#else
                if (ldlt_reg<NP>(As, Lf, iD))
                    ldlt_solve_reg<NP>(Lf, iD, rhs, dd);
                else
                    solve_ge_reg<NP>(Ab, dd);
#endif
            }
#pragma unroll
            for (int i = 0; i < NP; i++) {
                sh.d[i] = dd[i];
                sh.xd[i] = sh.x[i] - dd[i];
            }
        }
        __syncthreads();
        const double Sd = lm_cost<NP>(sh.xd, src, dst, mask, n, lane);
        if (lane == 0) {
            const double S = sh.S;
            double dS = 0, t = 0, dinf = 0;
            for (int a = 0; a < NP; a++) {
                double s = 0;
                for (int b = 0; b < NP; b++) s += sh.JtJ[a * NP + b] * sh.d[b];
                dS += sh.d[a] * (-s + 2 * sh.v[a]);
                t += sh.d[a] * sh.v[a];
                dinf = fmax(dinf, fabs(sh.d[a]));
            }
            const double R = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
            if (R > Rhi) {
                sh.lambda *= 0.5;
                if (sh.lambda < sh.lc) sh.lambda = 0;
            } else if (R < Rlo) {
                double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
                nu = fmin(fmax(nu, 2.), 10.);
                if (sh.lambda == 0) {
                    double maxval = DBL_EPSILON;
                    // diagonal of inv(JtJ): NP solves, in registers like the step's own (solve_ge_reg: the same arithmetic in the
                    // same order as solve_ge on the LDS copy, which took 25 us per 8 x 8 solve -- 200 us of a homography
                    // refinement whenever lambda reached 0, i.e. whenever LM starts close to its optimum, as it does after the refit)
                    double Js[NP][NP], Lj[NP][NP], iDj[NP];
#pragma unroll
                    for (int a = 0; a < NP; a++)
#pragma unroll
                        for (int b = 0; b < NP; b++) Js[a][b] = sh.JtJ[a * NP + b];
                    const bool spd = ldlt_reg<NP>(Js, Lj, iDj); // one factorisation for the NP unit vectors
#pragma unroll 1
                    for (int i = 0; i < NP; i++) {
                        double Abi[NP][NP + 1], col[NP];
                        if (spd) {
                            double ei[NP];
#pragma unroll
                            for (int a = 0; a < NP; a++) ei[a] = a == i ? 1. : 0.;
                            ldlt_solve_reg<NP>(Lj, iDj, ei, col);
                        } else {
#pragma unroll
                            for (int a = 0; a < NP; a++) {
#pragma unroll
                                for (int b = 0; b < NP; b++) Abi[a][b] = sh.JtJ[a * NP + b];
                                Abi[a][NP] = a == i ? 1. : 0.;
                            }
                            solve_ge_reg<NP>(Abi, col);
                        }
                        double ci = 0;
#pragma unroll
                        for (int a = 0; a < NP; a++)
                            if (a == i) ci = col[a];
                        maxval = fmax(maxval, fabs(ci));
                    }
                    sh.lambda = sh.lc = 1. / maxval;
                    nu *= 0.5;
                }
                sh.lambda *= nu;
            }
            sh.Sd = Sd;
            sh.proceed = dinf >= eps ? 1 : 0;
            if (Sd < S)
                for (int i = 0; i < NP; i++) sh.x[i] = sh.xd[i];
        }
        __syncthreads();
        if (sh.Sd < sh.S) lm_normal_equations<NP>(sh, src, dst, mask, n, lane, rinf); // sets sh.S = Sd
        iter++;
        const bool go = iter < max_iters && sh.proceed && rinf >= eps;
        __syncthreads();
        if (!go) break;
    }
}

#define REFINE_LDS_POINTS 3072
__global__ void __launch_bounds__(64) ransac_refine_kernel(int model, double *__restrict__ M, const float2 *src,
                                                           const float2 *dst, int n, float thr2, int refit,
                                                           int max_iters, uint8_t *mask,
                                                           int32_t *__restrict__ info) {
    __shared__ RefineShared sh;
    // The points and the inlier mask are staged in LDS (up to REFINE_LDS_POINTS of them): this kernel is ONE wavefront
    // that walks the point set about ten times (mask, normal equations and cost of every LM iteration), and from global
    // memory each walk cost a full memory round trip with nothing else in flight -- that latency, not arithmetic, was
    // most of its 31-35 us.  Same values, same order of operations.
    __shared__ float2 l_src[REFINE_LDS_POINTS], l_dst[REFINE_LDS_POINTS];
    __shared__ uint8_t l_mask[REFINE_LDS_POINTS];
    const int lane = threadIdx.x;
    const bool homog = model == MODEL_H;
    const int msz = homog ? 9 : 6;
    float F[8];
#pragma unroll
    for (int i = 0; i < 8; i++) F[i] = i < msz ? (float)M[i] : 0.f;
    const bool in_lds = n <= REFINE_LDS_POINTS;
    uint8_t *const gmask = mask;
    if (in_lds) {
        for (int i = lane; i < n; i += 64) l_src[i] = src[i], l_dst[i] = dst[i];
        __syncthreads();
        src = l_src, dst = l_dst, mask = l_mask; // (generic pointers: the helpers below read LDS through them)
    }
    // inlier mask of the winning hypothesis (cv findInliers)
    int cnt = 0;
    for (int i = lane; i < n; i += 64) {
        const float2 s = src[i], d = dst[i];
        const int f = reproj_err_f32(homog, F, s.x, s.y, d.x, d.y) <= thr2;
        mask[i] = (uint8_t)f;
        if (in_lds) gmask[i] = (uint8_t)f;
        cnt += f;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) cnt += __shfl_xor(cnt, m);
    if (lane == 0) info[0] = cnt;
    __syncthreads(); // mask[] is re-read by every lane below (same wave: make the stores visible)
    __threadfence_block();
    if (cnt == 0 || max_iters < 0) return;

    if (homog) {
        if (refit) {
            // all-inlier normalised DLT: centroid, mean absolute deviation, LtL, smallest eigenvector
            double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            for (int i = lane; i < n; i += 64)
                if (mask[i]) { c0 += src[i].x; c1 += src[i].y; c2 += dst[i].x; c3 += dst[i].y; }
            Norm2 N;
            N.cMx = wsum(c0) / cnt; N.cMy = wsum(c1) / cnt; N.cmx = wsum(c2) / cnt; N.cmy = wsum(c3) / cnt;
            c0 = c1 = c2 = c3 = 0;
            for (int i = lane; i < n; i += 64)
                if (mask[i]) {
                    c0 += fabs(src[i].x - N.cMx); c1 += fabs(src[i].y - N.cMy);
                    c2 += fabs(dst[i].x - N.cmx); c3 += fabs(dst[i].y - N.cmy);
                }
            N.sMx = wsum(c0); N.sMy = wsum(c1); N.smx = wsum(c2); N.smy = wsum(c3);
            const bool degenerate = fabs(N.smx) < DBL_EPSILON || fabs(N.smy) < DBL_EPSILON ||
                                    fabs(N.sMx) < DBL_EPSILON || fabs(N.sMy) < DBL_EPSILON;
            if (!degenerate) {
                N.smx = cnt / N.smx; N.smy = cnt / N.smy; N.sMx = cnt / N.sMx; N.sMy = cnt / N.sMy;
                double acc[45];
#pragma unroll
                for (int i = 0; i < 45; i++) acc[i] = 0;
                for (int i = lane; i < n; i += 64) {
                    if (!mask[i]) continue;
                    const double x = (dst[i].x - N.cmx) * N.smx, y = (dst[i].y - N.cmy) * N.smy;
                    const double X = (src[i].x - N.cMx) * N.sMx, Y = (src[i].y - N.cMy) * N.sMy;
                    const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
                    const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
                    int u = 0;
#pragma unroll
                    for (int j = 0; j < 9; j++) {
#pragma unroll
                        for (int k = j; k < 9; k++) acc[u++] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
                    }
                }
                wave_sums<45>(sh, acc, lane);
#pragma unroll
                for (int i = 0; i < 45; i++) acc[i] = sh.sum[i]; // (every lane: smallest_eigvec9 runs redundantly)
                double hv[9];
                const bool quick = smallest_eigvec9(acc, hv); // (wave-uniform: every lane holds the same sums)
                if (!quick) {
                    if (lane == 0) {
                        int u = 0;
#pragma unroll
                        for (int j = 0; j < 9; j++) {
#pragma unroll
                            for (int k = j; k < 9; k++) {
                                sh.A[j * 9 + k] = acc[u];
                                sh.A[k * 9 + j] = acc[u];
                                u++;
                            }
                        }
                    }
                    __syncthreads();
                    jacobi_lds(sh.A, sh.V, sh.W, 9, lane);
                }
                if (lane == 0) {
                    double h[9], H[9];
                    if (quick) {
#pragma unroll
                        for (int i = 0; i < 9; i++) h[i] = hv[i];
                    } else {
                        int best = 0;
                        for (int i = 1; i < 9; i++)
                            if (sh.W[i] < sh.W[best]) best = i;
                        for (int i = 0; i < 9; i++) h[i] = sh.V[best * 9 + i];
                    }
                    denormalise(h, N, H);
                    for (int i = 0; i < 9; i++) M[i] = H[i];
                }
                __syncthreads();
                __threadfence_block();
            }
        }
        if (max_iters > 0) {
            if (lane == 0)
                for (int i = 0; i < 8; i++) sh.x[i] = M[i];
            __syncthreads();
            lm_run<8>(sh, src, dst, mask, n, lane, max_iters);
            if (lane == 0)
                for (int i = 0; i < 8; i++) M[i] = sh.x[i];
        }
    } else if (max_iters > 0) {
        if (lane == 0) {
            sh.x[0] = M[0]; sh.x[1] = M[3]; sh.x[2] = M[2]; sh.x[3] = M[5];
        }
        __syncthreads();
        lm_run<4>(sh, src, dst, mask, n, lane, max_iters);
        if (lane == 0) {
            M[0] = sh.x[0]; M[4] = sh.x[0]; M[1] = -sh.x[1]; M[2] = sh.x[2]; M[3] = sh.x[1]; M[5] = sh.x[3];
        }
    }
}

extern "C" int isx_ransac_refine(int model, double *M_io_dev, const float *src_dev, const float *dst_dev, int n,
                                 double thr, int refit, int max_iters, uint8_t *mask_out_dev, int32_t *info_out_dev,
                                 void *stream) {
    ISX_REQUIRE(model == ISX_MODEL_AFFINE_PARTIAL || model == MODEL_H, "bad model %d", model);
    ISX_REQUIRE(n > 0, "n must be positive");
    ISX_REQUIRE(M_io_dev && src_dev && dst_dev && mask_out_dev && info_out_dev, "null device pointer");
    hipLaunchKernelGGL(ransac_refine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, model, M_io_dev,
                       (const float2 *)src_dev, (const float2 *)dst_dev, n, (float)(thr * thr), refit, max_iters,
                       mask_out_dev, info_out_dev);
    return isx_launch_status("ransac_refine_kernel");
}

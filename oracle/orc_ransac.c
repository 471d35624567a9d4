/*
 * oracle/orc_ransac.c -- TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * Robust transform estimation as the reference calls it (stitcher.py:135-141):
 *   type 0: M, _    = cv2.estimateAffinePartial2D(src_pts, dst_pts)          (all defaults)
 *   type 1: M, mask = cv2.findHomography(src_pts, dst_pts, cv2.RANSAC, 5.0)
 *
 * [OCV-K] Restated from the published OpenCV 4.x algorithm (modules/calib3d/src/ptsetreg.cpp,
 * fundam.cpp, levmarq.cpp; modules/core/src/lapack.cpp, rand.cpp).  PARITY UNPINNED: cv2 is absent
 * and the reference holds no vectors for this stage.  The analytic tests (exact 4-point homography,
 * exact 2-point similarity, planted inlier sets) do not depend on this restatement being faithful.
 *
 * Defaults encoded by the callers in oracle/orc.py:
 *   findHomography(RANSAC, 5.0): maxIters 2000, confidence 0.995, 4-point samples
 *   estimateAffinePartial2D    : threshold 3.0, maxIters 2000, confidence 0.99, refineIters 10
 */
#include "orc.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- cv::RNG (multiply-with-carry) ------------------------------------------------------------- */
typedef struct { uint64_t state; } cvrng;
static inline unsigned rng_next(cvrng *r) {
    r->state = (uint64_t)(unsigned)r->state * 4164903690U + (unsigned)(r->state >> 32);
    return (unsigned)r->state;
}
static inline int rng_uniform(cvrng *r, int a, int b) { return a == b ? a : (int)(rng_next(r) % (unsigned)(b - a) + a); }

/* ---- checkSubset -------------------------------------------------------------------------------- */
/* only the LAST point is tested against the lines through pairs of earlier points (4.x getSubset calls
 * checkSubset once with count = modelPoints) */
static int have_collinear(const float *p, int count) {
    int i = count - 1;
    for (int j = 0; j < i; j++) {
        double dx1 = p[2 * j] - p[2 * i], dy1 = p[2 * j + 1] - p[2 * i + 1];
        for (int k = 0; k < j; k++) {
            double dx2 = p[2 * k] - p[2 * i], dy2 = p[2 * k + 1] - p[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return 1;
        }
    }
    return 0;
}

static double det3(const double *a) {
    return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}

static int check_subset(int model, const float *ms1, const float *ms2, int count) {
    if (have_collinear(ms1, count) || have_collinear(ms2, count)) return 0;
    if (model == ORC_MODEL_HOMOGRAPHY && count == 4) {
        /* orientation consistency of every 3-subset ("Speeding-up homography estimation in mobile devices") */
        static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
        int negative = 0;
        for (int i = 0; i < 4; i++) {
            const int *t = tt[i];
            double A[9] = {ms1[2 * t[0]], ms1[2 * t[0] + 1], 1., ms1[2 * t[1]], ms1[2 * t[1] + 1], 1., ms1[2 * t[2]], ms1[2 * t[2] + 1], 1.};
            double B[9] = {ms2[2 * t[0]], ms2[2 * t[0] + 1], 1., ms2[2 * t[1]], ms2[2 * t[1] + 1], 1., ms2[2 * t[2]], ms2[2 * t[2] + 1], 1.};
            negative += det3(A) * det3(B) < 0;
        }
        if (negative != 0 && negative != 4) return 0;
    }
    return 1;
}

static int model_points(int model) { return model == ORC_MODEL_HOMOGRAPHY ? 4 : 2; }

/* RANSACPointSetRegistrator::getSubset: draw modelPoints distinct indices, retry (<= maxAttempts) until
 * checkSubset passes.  idx_out receives the accepted indices. */
static int get_subset(int model, const float *m1, const float *m2, int count, cvrng *rng, int max_attempts,
                      int *idx_out, float *ms1, float *ms2) {
    int mp = model_points(model), iters = 0;
    for (; iters < max_attempts; iters++) {
        for (int i = 0; i < mp; ++i) {
            int idx_i;
            for (;;) {
                idx_i = rng_uniform(rng, 0, count);
                int dup = 0;
                for (int k = 0; k < i; k++) dup |= idx_out[k] == idx_i;
                if (!dup) break;
            }
            idx_out[i] = idx_i;
            ms1[2 * i] = m1[2 * idx_i]; ms1[2 * i + 1] = m1[2 * idx_i + 1];
            ms2[2 * i] = m2[2 * idx_i]; ms2[2 * i + 1] = m2[2 * idx_i + 1];
        }
        if (check_subset(model, ms1, ms2, mp)) break;
    }
    return iters < max_attempts;
}

/* The subset sequence does not depend on the models (only on the points), so it can be listed up front.
 * Returns the number of subsets produced (< max_iters only if an attempt budget of 10000 ran out). */
int orc_ransac_sample(int model, const float *src, const float *dst, int count, int max_iters,
                      int32_t *subsets_out) {
    cvrng rng = {(uint64_t)-1};
    int mp = model_points(model);
    float ms1[8], ms2[8];
    int idx[4];
    if (count < mp) return 0;
    if (count == mp) {
        for (int i = 0; i < mp; i++) subsets_out[i] = i;
        return 1;
    }
    for (int it = 0; it < max_iters; it++) {
        if (!get_subset(model, src, dst, count, &rng, 10000, idx, ms1, ms2)) return it;
        for (int i = 0; i < mp; i++) subsets_out[it * mp + i] = idx[i];
    }
    return max_iters;
}

/* ---- Jacobi eigen-solver for symmetric matrices (cv::eigen without Eigen/LAPACK), n <= 9 ---------- */
static void jacobi(double *A, double *W, double *V, int n) {
    const double eps = DBL_EPSILON;
    int indR[16], indC[16];
    int i, j, k, m;
    for (i = 0; i < n; i++) {
        for (j = 0; j < n; j++) V[i * n + j] = 0;
        V[i * n + i] = 1;
    }
    double mv = 0;
    for (k = 0; k < n; k++) {
        W[k] = A[(n + 1) * k];
        if (k < n - 1) {
            for (m = k + 1, mv = fabs(A[n * k + m]), i = k + 2; i < n; i++) {
                double val = fabs(A[n * k + i]);
                if (mv < val) mv = val, m = i;
            }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = fabs(A[k]), i = 1; i < k; i++) {
                double val = fabs(A[n * i + k]);
                if (mv < val) mv = val, m = i;
            }
            indC[k] = m;
        }
    }
    if (n > 1)
        for (int iters = 0, maxIters = n * n * 30; iters < maxIters; iters++) {
            for (k = 0, mv = fabs(A[indR[0]]), i = 1; i < n - 1; i++) {
                double val = fabs(A[n * i + indR[i]]);
                if (mv < val) mv = val, k = i;
            }
            int l = indR[k];
            for (i = 1; i < n; i++) {
                double val = fabs(A[n * indC[i] + i]);
                if (mv < val) mv = val, k = indC[i], l = i;
            }
            double p = A[n * k + l];
            if (fabs(p) <= eps) break;
            double y = (W[l] - W[k]) * 0.5;
            double t = fabs(y) + hypot(p, y);
            double s = hypot(p, t);
            double c = t / s;
            s = p / s;
            t = (p / t) * p;
            if (y < 0) s = -s, t = -t;
            A[n * k + l] = 0;
            W[k] -= t;
            W[l] += t;
            double a0, b0;
#define ROT(v0, v1) (a0 = (v0), b0 = (v1), (v0) = a0 * c - b0 * s, (v1) = a0 * s + b0 * c)
            for (i = 0; i < k; i++) ROT(A[n * i + k], A[n * i + l]);
            for (i = k + 1; i < l; i++) ROT(A[n * k + i], A[n * i + l]);
            for (i = l + 1; i < n; i++) ROT(A[n * k + i], A[n * l + i]);
            for (i = 0; i < n; i++) ROT(V[n * k + i], V[n * l + i]);
#undef ROT
            for (j = 0; j < 2; j++) {
                int idx = j == 0 ? k : l;
                if (idx < n - 1) {
                    for (m = idx + 1, mv = fabs(A[n * idx + m]), i = idx + 2; i < n; i++) {
                        double val = fabs(A[n * idx + i]);
                        if (mv < val) mv = val, m = i;
                    }
                    indR[idx] = m;
                }
                if (idx > 0) {
                    for (m = 0, mv = fabs(A[idx]), i = 1; i < idx; i++) {
                        double val = fabs(A[n * i + idx]);
                        if (mv < val) mv = val, m = i;
                    }
                    indC[idx] = m;
                }
            }
        }
    /* sort eigenvalues (descending) with their eigenvectors (rows of V) */
    for (k = 0; k < n - 1; k++) {
        m = k;
        for (i = k + 1; i < n; i++)
            if (W[m] < W[i]) m = i;
        if (k != m) {
            double t = W[m]; W[m] = W[k]; W[k] = t;
            for (i = 0; i < n; i++) { t = V[n * m + i]; V[n * m + i] = V[n * k + i]; V[n * k + i] = t; }
        }
    }
}

/* cv::solve(A, b, x, DECOMP_EIG) for symmetric A (n <= 8): x = sum_i (v_i.b / w_i) v_i, skipping
 * eigenvalues below 2*eps*sum|w| */
static void solve_eig(const double *A_in, const double *b, double *x, int n) {
    double A[64], W[8], V[64];
    memcpy(A, A_in, sizeof(double) * n * n);
    jacobi(A, W, V, n);
    double thr = 0;
    for (int i = 0; i < n; i++) thr += fabs(W[i]);
    thr *= DBL_EPSILON * 2;
    for (int i = 0; i < n; i++) x[i] = 0;
    for (int i = 0; i < n; i++) {
        if (fabs(W[i]) <= thr) continue;
        double s = 0;
        for (int k = 0; k < n; k++) s += V[n * i + k] * b[k];
        s /= W[i];
        for (int k = 0; k < n; k++) x[k] += s * V[n * i + k];
    }
}

/* ---- model fitting -------------------------------------------------------------------------------- */
/* HomographyEstimatorCallback::runKernel: normalised DLT (centroid + mean absolute deviation), 9x9 LtL,
 * eigenvector of the smallest eigenvalue, de-normalise, divide by H[8].  m1 = src (M), m2 = dst (m). */
static int fit_homography(const float *M, const float *m, int count, double *H) {
    double LtL[81], W[9], V[81];
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; i++) {
        cmx += m[2 * i]; cmy += m[2 * i + 1];
        cMx += M[2 * i]; cMy += M[2 * i + 1];
    }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; i++) {
        smx += fabs(m[2 * i] - cmx); smy += fabs(m[2 * i + 1] - cmy);
        sMx += fabs(M[2 * i] - cMx); sMy += fabs(M[2 * i + 1] - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON)
        return 0;
    smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
    double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    memset(LtL, 0, sizeof(LtL));
    for (int i = 0; i < count; i++) {
        double x = (m[2 * i] - cmx) * smx, y = (m[2 * i + 1] - cmy) * smy;
        double X = (M[2 * i] - cMx) * sMx, Y = (M[2 * i + 1] - cMy) * sMy;
        double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; j++)
            for (int k = j; k < 9; k++) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; j++)
        for (int k = 0; k < j; k++) LtL[j * 9 + k] = LtL[k * 9 + j]; /* completeSymm */
    jacobi(LtL, W, V, 9);
    const double *H0 = V + 72; /* eigenvector of the smallest eigenvalue */
    double Ht[9], Hf[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += invHnorm[r * 3 + k] * H0[k * 3 + c];
            Ht[r * 3 + c] = s;
        }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += Ht[r * 3 + k] * Hnorm2[k * 3 + c];
            Hf[r * 3 + c] = s;
        }
    double sc = 1. / Hf[8];
    for (int i = 0; i < 9; i++) H[i] = Hf[i] * sc;
    return 1;
}

/* AffinePartial2DEstimatorCallback::runKernel: closed form from the first two correspondences */
static int fit_affine_partial(const float *from, const float *to, double *M) {
    double x1 = from[0], y1 = from[1], x2 = from[2], y2 = from[3];
    double X1 = to[0], Y1 = to[1], X2 = to[2], Y2 = to[3];
    double d = 1. / ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
    double S0 = d * ((X1 - X2) * (x1 - x2) + (Y1 - Y2) * (y1 - y2));
    double S1 = d * ((Y1 - Y2) * (x1 - x2) - (X1 - X2) * (y1 - y2));
    double S2 = d * ((Y1 - Y2) * (x1 * y2 - x2 * y1) - (X1 * y2 - X2 * y1) * (y1 - y2) - (X1 * x2 - X2 * x1) * (x1 - x2));
    double S3 = d * (-(X1 - X2) * (x1 * y2 - x2 * y1) - (Y1 * x2 - Y2 * x1) * (x1 - x2) - (Y1 * y2 - Y2 * y1) * (y1 - y2));
    M[0] = M[4] = S0;
    M[1] = -S1;
    M[2] = S2;
    M[3] = S1;
    M[5] = S3;
    return 1;
}

int orc_fit_model(int model, const float *src, const float *dst, int n, double *model_out) {
    if (model == ORC_MODEL_HOMOGRAPHY) return n >= 4 ? fit_homography(src, dst, n, model_out) : 0;
    return n >= 2 ? fit_affine_partial(src, dst, model_out) : 0;
}

/* computeError: float32 arithmetic, no FMA (the file is compiled with -ffp-contract=off) */
void orc_reproj_err(int model, const double *M, const float *src, const float *dst, int n, float *err) {
    if (model == ORC_MODEL_HOMOGRAPHY) {
        float Hf[8];
        for (int i = 0; i < 8; i++) Hf[i] = (float)M[i];
        for (int i = 0; i < n; i++) {
            float Mx = src[2 * i], My = src[2 * i + 1];
            float ww = 1.f / (Hf[6] * Mx + Hf[7] * My + 1.f);
            float dx = (Hf[0] * Mx + Hf[1] * My + Hf[2]) * ww - dst[2 * i];
            float dy = (Hf[3] * Mx + Hf[4] * My + Hf[5]) * ww - dst[2 * i + 1];
            err[i] = dx * dx + dy * dy;
        }
    } else {
        float F0 = (float)M[0], F1 = (float)M[1], F2 = (float)M[2], F3 = (float)M[3], F4 = (float)M[4], F5 = (float)M[5];
        for (int i = 0; i < n; i++) {
            float fx = src[2 * i], fy = src[2 * i + 1];
            float a = F0 * fx + F1 * fy + F2 - dst[2 * i];
            float b = F3 * fx + F4 * fy + F5 - dst[2 * i + 1];
            err[i] = a * a + b * b;
        }
    }
}

/* findInliers: err <= (float)(thr*thr) */
int orc_score(int model, const double *M, const float *src, const float *dst, int n, double thr, uint8_t *mask_out) {
    float *err = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    orc_reproj_err(model, M, src, dst, n, err);
    float t = (float)(thr * thr);
    int nz = 0;
    for (int i = 0; i < n; i++) {
        int f = err[i] <= t;
        if (mask_out) mask_out[i] = (uint8_t)f;
        nz += f;
    }
    free(err);
    return nz;
}

static int update_num_iters(double p, double ep, int mp, int max_iters) {
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, mp);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* RANSACPointSetRegistrator::run.  Optional trace outputs list every iteration's hypothesis. */
static int ransac_run(int model, const float *m1, const float *m2, int count, double thr, int max_iters,
                      double confidence, double *best_model, uint8_t *best_mask, double *hyp_out,
                      int32_t *cnt_out, int32_t *valid_out, int *n_iter_out, int *best_iter_out) {
    int mp = model_points(model), msz = model == ORC_MODEL_HOMOGRAPHY ? 9 : 6;
    int niters = max_iters > 1 ? max_iters : 1, max_good = 0, iter;
    cvrng rng = {(uint64_t)-1};
    float ms1[8], ms2[8];
    int idx[4];
    double mdl[9];
    if (n_iter_out) *n_iter_out = 0;
    if (best_iter_out) *best_iter_out = -1;
    if (count < mp) return 0;
    uint8_t *mask = (uint8_t *)malloc((size_t)count);
    if (count == mp) {
        int ok = orc_fit_model(model, m1, m2, count, best_model) > 0;
        if (ok && best_mask) memset(best_mask, 1, (size_t)count);
        free(mask);
        return ok;
    }
    for (iter = 0; iter < niters; iter++) {
        if (!get_subset(model, m1, m2, count, &rng, 10000, idx, ms1, ms2)) {
            if (iter == 0) { free(mask); return 0; }
            break;
        }
        int nm = orc_fit_model(model, ms1, ms2, mp, mdl);
        if (valid_out) valid_out[iter] = nm > 0;
        if (hyp_out) memcpy(hyp_out + (size_t)iter * msz, mdl, sizeof(double) * msz);
        if (cnt_out) cnt_out[iter] = 0;
        if (n_iter_out) *n_iter_out = iter + 1;
        if (nm <= 0) continue;
        int good = orc_score(model, mdl, m1, m2, count, thr, mask);
        if (cnt_out) cnt_out[iter] = good;
        if (good > (max_good > mp - 1 ? max_good : mp - 1)) {
            if (best_mask) memcpy(best_mask, mask, (size_t)count);
            memcpy(best_model, mdl, sizeof(double) * msz);
            max_good = good;
            if (best_iter_out) *best_iter_out = iter;
            niters = update_num_iters(confidence, (double)(count - good) / count, mp, niters);
        }
    }
    free(mask);
    return max_good > 0;
}

int orc_ransac_trace(int model, const float *src, const float *dst, int n, double thr, int max_iters,
                     double confidence, double *hyp_out, int32_t *cnt_out, int32_t *valid_out,
                     int *n_iter_out, int *best_iter_out) {
    double best[9];
    uint8_t *mask = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    int r = ransac_run(model, src, dst, n, thr, max_iters, confidence, best, mask, hyp_out, cnt_out, valid_out,
                       n_iter_out, best_iter_out);
    free(mask);
    return r;
}

/* ---- Levenberg-Marquardt (cv::LMSolver, levmarq.cpp) ------------------------------------------------- */
/* residuals r (2n) and Jacobian J (2n x np) of the refine callbacks */
static void lm_compute(int model, const double *h, const float *src, const float *dst, int n, double *r, double *J) {
    int np_ = model == ORC_MODEL_HOMOGRAPHY ? 8 : 4;
    for (int i = 0; i < n; i++) {
        double Mx = src[2 * i], My = src[2 * i + 1];
        if (model == ORC_MODEL_HOMOGRAPHY) {
            double ww = h[6] * Mx + h[7] * My + 1.;
            ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
            double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
            double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
            r[2 * i] = xi - dst[2 * i];
            r[2 * i + 1] = yi - dst[2 * i + 1];
            if (J) {
                double *Jp = J + (size_t)2 * i * np_;
                Jp[0] = Mx * ww; Jp[1] = My * ww; Jp[2] = ww;
                Jp[3] = Jp[4] = Jp[5] = 0.;
                Jp[6] = -Mx * ww * xi; Jp[7] = -My * ww * xi;
                Jp[8] = Jp[9] = Jp[10] = 0.;
                Jp[11] = Mx * ww; Jp[12] = My * ww; Jp[13] = ww;
                Jp[14] = -Mx * ww * yi; Jp[15] = -My * ww * yi;
            }
        } else {
            double xi = h[0] * Mx - h[1] * My + h[2];
            double yi = h[1] * Mx + h[0] * My + h[3];
            r[2 * i] = xi - dst[2 * i];
            r[2 * i + 1] = yi - dst[2 * i + 1];
            if (J) {
                double *Jp = J + (size_t)2 * i * np_;
                Jp[0] = Mx; Jp[1] = -My; Jp[2] = 1.; Jp[3] = 0.;
                Jp[4] = My; Jp[5] = Mx; Jp[6] = 0.; Jp[7] = 1.;
            }
        }
    }
}

static double sumsq(const double *v, int n) { double s = 0; for (int i = 0; i < n; i++) s += v[i] * v[i]; return s; }
static double maxabs(const double *v, int n) { double s = 0; for (int i = 0; i < n; i++) s = fabs(v[i]) > s ? fabs(v[i]) : s; return s; }

static void normal_eqs(const double *J, const double *r, int rows, int np_, double *A, double *v) {
    for (int a = 0; a < np_; a++) {
        for (int b = 0; b < np_; b++) {
            double s = 0;
            for (int i = 0; i < rows; i++) s += J[(size_t)i * np_ + a] * J[(size_t)i * np_ + b];
            A[a * np_ + b] = s;
        }
        double s = 0;
        for (int i = 0; i < rows; i++) s += J[(size_t)i * np_ + a] * r[i];
        v[a] = s;
    }
}

static int lm_run(int model, double *x, const float *src, const float *dst, int n, int max_iters) {
    int np_ = model == ORC_MODEL_HOMOGRAPHY ? 8 : 4, rows = 2 * n;
    const double eps = FLT_EPSILON; /* LMSolver::create(cb, maxIters): eps = FLT_EPSILON for x and f */
    double *r = (double *)malloc(sizeof(double) * rows), *rd = (double *)malloc(sizeof(double) * rows);
    double *J = (double *)malloc(sizeof(double) * (size_t)rows * np_);
    double A[64], Ap[64], v[8], d[8], xd[8], D[8], tmp[8];
    lm_compute(model, x, src, dst, n, r, J);
    double S = sumsq(r, rows);
    normal_eqs(J, r, rows, np_, A, v);
    for (int i = 0; i < np_; i++) D[i] = A[i * np_ + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        memcpy(Ap, A, sizeof(double) * np_ * np_);
        for (int i = 0; i < np_; i++) Ap[i * np_ + i] += lambda * D[i];
        solve_eig(Ap, v, d, np_);
        for (int i = 0; i < np_; i++) xd[i] = x[i] - d[i];
        lm_compute(model, xd, src, dst, n, rd, NULL);
        double Sd = sumsq(rd, rows);
        for (int a = 0; a < np_; a++) { /* temp_d = -A*d + 2*v */
            double s = 0;
            for (int b = 0; b < np_; b++) s += A[a * np_ + b] * d[b];
            tmp[a] = -s + 2 * v[a];
        }
        double dS = 0;
        for (int i = 0; i < np_; i++) dS += d[i] * tmp[i];
        double R = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            double t = 0;
            for (int i = 0; i < np_; i++) t += d[i] * v[i];
            double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = nu > 2. ? nu : 2.;
            nu = nu < 10. ? nu : 10.;
            if (lambda == 0) {
                /* invert(A, Ap, DECOMP_EIG): only the diagonal of the inverse is used */
                double maxval = DBL_EPSILON;
                for (int i = 0; i < np_; i++) {
                    double e[8] = {0}, col[8];
                    e[i] = 1;
                    solve_eig(A, e, col, np_);
                    maxval = fabs(col[i]) > maxval ? fabs(col[i]) : maxval;
                }
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            memcpy(x, xd, sizeof(double) * np_);
            lm_compute(model, x, src, dst, n, r, J);
            normal_eqs(J, r, rows, np_, A, v);
        }
        iter++;
        int proceed = iter < max_iters && maxabs(d, np_) >= eps && maxabs(r, rows) >= eps;
        if (!proceed) break;
    }
    free(r); free(rd); free(J);
    return iter;
}

/* findHomography tail (all-inlier DLT refit + LM on 8 parameters) or estimateAffinePartial2D tail
 * (LM on a,b,tx,ty starting from the RANSAC model).  src/dst: the n INLIER correspondences. */
int orc_refine(int model, const float *src, const float *dst, int n, int refit, int max_iters, double *M) {
    if (n <= 0) return 0;
    if (model == ORC_MODEL_HOMOGRAPHY) {
        if (refit) fit_homography(src, dst, n, M); /* a failed refit leaves M unchanged, as cv does */
        lm_run(model, M, src, dst, n, max_iters);  /* M[8] stays 1 */
    } else {
        double h[4] = {M[0], M[3], M[2], M[5]};
        lm_run(model, h, src, dst, n, max_iters);
        M[0] = M[4] = h[0];
        M[1] = -h[1];
        M[2] = h[2];
        M[3] = h[1];
        M[5] = h[3];
    }
    return 1;
}

static int compress(const float *p, const uint8_t *mask, int n, float *out) {
    int k = 0;
    for (int i = 0; i < n; i++)
        if (mask[i]) { out[2 * k] = p[2 * i]; out[2 * k + 1] = p[2 * i + 1]; k++; }
    return k;
}

/* cv::findHomography(src, dst, RANSAC, thr) */
int orc_find_homography_ransac(const float *src, const float *dst, int n, double thr, int max_iters,
                               double confidence, double *M_out, uint8_t *mask_out) {
    if (n < 4) return 0; /* cv raises / returns empty */
    uint8_t *mask = (uint8_t *)malloc((size_t)n);
    int result;
    if (n == 4) {
        memset(mask, 1, (size_t)n);
        result = fit_homography(src, dst, n, M_out) > 0;
    } else {
        result = ransac_run(ORC_MODEL_HOMOGRAPHY, src, dst, n, thr, max_iters, confidence, M_out, mask, NULL,
                            NULL, NULL, NULL, NULL);
    }
    if (result && n > 4) {
        float *s = (float *)malloc(sizeof(float) * 2 * n), *d = (float *)malloc(sizeof(float) * 2 * n);
        compress(src, mask, n, s);
        int k = compress(dst, mask, n, d);
        if (k > 0) orc_refine(ORC_MODEL_HOMOGRAPHY, s, d, k, 1, 10, M_out);
        free(s); free(d);
    }
    if (mask_out) { if (result) memcpy(mask_out, mask, (size_t)n); else memset(mask_out, 0, (size_t)n); }
    free(mask);
    return result;
}

/* cv::estimateAffinePartial2D(from, to) */
int orc_estimate_affine_partial(const float *src, const float *dst, int n, double thr, int max_iters,
                                double confidence, int refine_iters, double *M_out, uint8_t *mask_out) {
    if (n < 2) return 0;
    uint8_t *mask = (uint8_t *)malloc((size_t)n);
    int result = ransac_run(ORC_MODEL_AFFINE_PARTIAL, src, dst, n, thr, max_iters, confidence, M_out, mask, NULL,
                            NULL, NULL, NULL, NULL);
    if (result && n > 2 && refine_iters) {
        float *s = (float *)malloc(sizeof(float) * 2 * n), *d = (float *)malloc(sizeof(float) * 2 * n);
        compress(src, mask, n, s);
        int k = compress(dst, mask, n, d);
        if (k > 0) orc_refine(ORC_MODEL_AFFINE_PARTIAL, s, d, k, 0, refine_iters, M_out);
        free(s); free(d);
    }
    if (mask_out) { if (result) memcpy(mask_out, mask, (size_t)n); else memset(mask_out, 0, (size_t)n); }
    free(mask);
    return result;
}

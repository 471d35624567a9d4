/*
 * oracle/orc_match.c -- TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * kNN(k=2) brute-force matching + Lowe ratio test as the reference runs them:
 *   stitcher.py:85      cv2.BFMatcher()  => NORM_L2, crossCheck=False for every algorithm
 *   stitcher.py:106-116 get_matches      => matcher.knnMatch(des1, des2, k=2)
 *   stitcher.py:118-127 filter_matches   => Lowe ratio in Python floats (float64)
 *
 * [OCV-K] restated from OpenCV 4.x modules/core/src/batch_distance.cpp (parity unpinned, cv2 absent):
 *   - distances are float32: sqrtf of the squared L2 sum (uint8: exact integer sum; float32:
 *     float accumulate), or the popcount converted to float for NORM_HAMMING;
 *   - per query row the K-best list starts as dist=FLT_MAX, idx=-1; for j ascending:
 *       if d < best[K-1]: shift while best[k] > d, insert   (float bit patterns compared as int32)
 *     => ties keep the lower train index in both slots; K = min(2, nt).
 */
#include "orc.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int32_t f2i(float f) { int32_t i; memcpy(&i, &f, 4); return i; }

/* batch_distance.cpp BatchDistInvoker K>0 branch */
static inline void insert_k2(int K, float d, int j, float *bd, int32_t *bi) {
    int32_t di = f2i(d);
    if (di < f2i(bd[K - 1])) {
        int k;
        for (k = K - 2; k >= 0 && f2i(bd[k]) > di; k--) {
            bi[k + 1] = bi[k];
            bd[k + 1] = bd[k];
        }
        bi[k + 1] = j;
        bd[k + 1] = d;
    }
}

__attribute__((target_clones("arch=skylake-avx512", "avx2", "default")))
static int l2sqr_u8(const uint8_t *a, const uint8_t *b, int d) {
    int s = 0;
    for (int k = 0; k < d; k++) {
        int v = (int)a[k] - (int)b[k];
        s += v * v;
    }
    return s;
}

static int hamming_u8(const uint8_t *a, const uint8_t *b, int d) {
    int s = 0;
    for (int k = 0; k < d; k++) s += __builtin_popcount((unsigned)(a[k] ^ b[k]));
    return s;
}

/* [OCV-K] normL2Sqr_(const float*, const float*, int): SIMD-lane partial sums then a scalar tail.
 * 4 independent accumulators stand in for the vector lanes; exact for integer-valued data. */
__attribute__((target_clones("arch=skylake-avx512", "avx2", "default")))
static float l2sqr_f32(const float *a, const float *b, int d) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= d; k += 4) {
        float v0 = a[k] - b[k], v1 = a[k + 1] - b[k + 1], v2 = a[k + 2] - b[k + 2], v3 = a[k + 3] - b[k + 3];
        s0 += v0 * v0; s1 += v1 * v1; s2 += v2 * v2; s3 += v3 * v3;
    }
    float s = (s0 + s1) + (s2 + s3);
    for (; k < d; k++) { float v = a[k] - b[k]; s += v * v; }
    return s;
}

void orc_knn2_u8(const uint8_t *q, int nq, const uint8_t *t, int nt, int d, int norm,
                 int32_t *idx_out, float *dist_out, int nthreads) {
    int K = nt < 2 ? nt : 2;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (int i = 0; i < nq; i++) {
        float bd[2] = {FLT_MAX, FLT_MAX};
        int32_t bi[2] = {-1, -1};
        const uint8_t *a = q + (size_t)i * d;
        if (K > 0) {
            for (int j = 0; j < nt; j++) {
                const uint8_t *b = t + (size_t)j * d;
                float dd = norm == ORC_NORM_HAMMING ? (float)hamming_u8(a, b, d)
                                                    : sqrtf((float)l2sqr_u8(a, b, d));
                insert_k2(K, dd, j, bd, bi);
            }
        }
        idx_out[2 * i] = bi[0]; idx_out[2 * i + 1] = bi[1];
        dist_out[2 * i] = bd[0]; dist_out[2 * i + 1] = bd[1];
    }
    (void)nthreads;
}

void orc_knn2_f32(const float *q, int nq, const float *t, int nt, int d,
                  int32_t *idx_out, float *dist_out, int nthreads) {
    int K = nt < 2 ? nt : 2;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (int i = 0; i < nq; i++) {
        float bd[2] = {FLT_MAX, FLT_MAX};
        int32_t bi[2] = {-1, -1};
        const float *a = q + (size_t)i * d;
        if (K > 0) {
            for (int j = 0; j < nt; j++) {
                float dd = sqrtf(l2sqr_f32(a, t + (size_t)j * d, d));
                insert_k2(K, dd, j, bd, bi);
            }
        }
        idx_out[2 * i] = bi[0]; idx_out[2 * i + 1] = bi[1];
        dist_out[2 * i] = bd[0]; dist_out[2 * i + 1] = bd[1];
    }
    (void)nthreads;
}

/* stitcher.py:118-127.  A row with a single match (nt == 1) is kept unconditionally (:125-126);
 * a row with no match at all (nt == 0 never reaches here: knnMatch returns [] for an empty train set). */
int orc_ratio_filter(const int32_t *idx, const float *dist, int nq, double ratio,
                     int32_t *good_q, int32_t *good_t) {
    int g = 0;
    for (int i = 0; i < nq; i++) {
        int keep;
        if (idx[2 * i] < 0) continue;
        if (idx[2 * i + 1] >= 0)
            keep = (double)dist[2 * i] < ratio * (double)dist[2 * i + 1];
        else
            keep = 1;
        if (keep) {
            if (good_q) good_q[g] = i;
            if (good_t) good_t[g] = idx[2 * i];
            g++;
        }
    }
    return g;
}

int orc_match_count_u8(const uint8_t *q, int nq, const uint8_t *t, int nt, int d, int norm,
                       double ratio, int nthreads) {
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(nq > 0 ? nq : 1));
    float *dist = (float *)malloc(sizeof(float) * 2 * (size_t)(nq > 0 ? nq : 1));
    orc_knn2_u8(q, nq, t, nt, d, norm, idx, dist, nthreads);
    int g = orc_ratio_filter(idx, dist, nq, ratio, NULL, NULL);
    free(idx); free(dist);
    return g;
}

int orc_match_count_f32(const float *q, int nq, const float *t, int nt, int d,
                        double ratio, int nthreads) {
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(nq > 0 ? nq : 1));
    float *dist = (float *)malloc(sizeof(float) * 2 * (size_t)(nq > 0 ? nq : 1));
    orc_knn2_f32(q, nq, t, nt, d, idx, dist, nthreads);
    int g = orc_ratio_filter(idx, dist, nq, ratio, NULL, NULL);
    free(idx); free(dist);
    return g;
}

int orc_version(void) { return 1; }

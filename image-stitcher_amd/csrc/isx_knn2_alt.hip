// The two non-default matchers of SURVEY.md section 8a-1:
//
//  (c) Hamming kNN-2 -- XOR + v_bcnt_u32_b32.  Opt-in: the reference builds cv2.BFMatcher() with no
//      arguments (stitcher.py:85), i.e. NORM_L2 even for ORB/AKAZE; this kernel is the parity target of
//      cv2.BFMatcher(cv2.NORM_HAMMING).knnMatch(k=2): distance = popcount as float32, ties to the lowest
//      train index in both slots.  One lane owns one query (its words stay in registers), the train rows
//      stream through LDS and are read as wave-wide broadcasts; top-2 on the packed key dist<<20 | j.
//
//  (b) float32 L2 kNN-2 -- for float descriptors that are NOT integer valued (cv2.SIFT's are, and take
//      the exact int8 path).  ||a-b||^2 = na + nb - 2 a.b with a.b on v_mfma_f32_32x32x2_f32 (exact f32
//      FMA chain), ranking on that value; the distances of the two winners are then recomputed directly
//      as sum (a-b)^2 in the oracle's order, so reported distances carry no cancellation error.
//      Gate: 1e-4 relative distance (BASELINE.json north_star); indices may differ from cv2 only between
//      rows whose distances agree to that tolerance.
#include "isx_common.h"
#include <float.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// Hamming
// ------------------------------------------------------------------------------------------------
template <int W> // W = 32-bit words per descriptor (8 = ORB's 256 bits, 16 = AKAZE's 61 bytes padded)
__global__ void __launch_bounds__(256)
knn2_hamming_kernel(const uint32_t *__restrict__ desc, const isx_knn_item *__restrict__ items, int n_items,
                    int nqb, double ratio, int32_t *__restrict__ idx_out, float *__restrict__ dist_out,
                    const int64_t *__restrict__ out_start, int32_t *__restrict__ count_out) {
    constexpr int CHUNK = 256;
    __shared__ __align__(16) uint32_t sm[CHUNK * W];
    const int item_i = blockIdx.x / nqb, qb = blockIdx.x % nqb;
    if (item_i >= n_items) return;
    const isx_knn_item it = items[item_i];
    if (qb * 256 >= it.nq) return; // uniform per block
    const int tid = threadIdx.x;
    const int qi = qb * 256 + tid;
    const bool live = qi < it.nq;
    uint32_t q[W];
    {
        const v4i *p = (const v4i *)(desc + (int64_t)(it.q_start + (live ? qi : 0)) * W);
#pragma unroll
        for (int u = 0; u < W / 4; u++) {
            const v4i v = p[u];
            q[4 * u] = v[0], q[4 * u + 1] = v[1], q[4 * u + 2] = v[2], q[4 * u + 3] = v[3];
        }
    }
    uint32_t m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu; // key = dist << 20 | j   (j < 2^18, dist <= 512)
    for (int c0 = 0; c0 < it.nt; c0 += CHUNK) {
        const int rows = min(CHUNK, it.nt - c0);
        __syncthreads();
        for (int u = tid; u < rows * (W / 4); u += 256)
            ((v4i *)sm)[u] = ((const v4i *)(desc + (int64_t)(it.t_start + c0) * W))[u];
        __syncthreads();
        for (int r = 0; r < rows; r++) {
            const v4i *tp = (const v4i *)(sm + r * W); // same address on every lane: LDS broadcast
            uint32_t d = 0;
#pragma unroll
            for (int u = 0; u < W / 4; u++) {
                const v4i t = tp[u];
                d += __popc(q[4 * u] ^ (uint32_t)t[0]) + __popc(q[4 * u + 1] ^ (uint32_t)t[1]) +
                     __popc(q[4 * u + 2] ^ (uint32_t)t[2]) + __popc(q[4 * u + 3] ^ (uint32_t)t[3]);
            }
            const uint32_t key = (d << 20) | (uint32_t)(c0 + r);
            const uint32_t lo = m1 < key ? m1 : key, hi = m1 < key ? key : m1;
            m2 = hi < m2 ? hi : m2; // second smallest of {m1, m2, key}
            m1 = lo;
        }
    }
    bool good = false;
    if (live) {
        const bool has0 = m1 != 0xFFFFFFFFu, has1 = m2 != 0xFFFFFFFFu;
        const float d0 = has0 ? (float)(m1 >> 20) : FLT_MAX, d1 = has1 ? (float)(m2 >> 20) : FLT_MAX;
        good = has0 && (!has1 || (double)d0 < ratio * (double)d1);
        if (idx_out) {
            const int64_t o = (out_start[item_i] + qi) * 2;
            idx_out[o] = has0 ? (int)(m1 & 0xFFFFFu) : -1;
            idx_out[o + 1] = has1 ? (int)(m2 & 0xFFFFFu) : -1;
            dist_out[o] = d0;
            dist_out[o + 1] = d1;
        }
    }
    const int cnt = __popcll(__ballot(good));
    if (count_out && (tid & 63) == 0 && cnt) atomicAdd(&count_out[item_i], cnt);
}

extern "C" int isx_knn2_hamming_items(const uint8_t *desc_dev, int d_pad, const isx_knn_item *items_dev, int n_items,
                                      int max_nq, int max_nt, double ratio, int32_t *idx_out_dev, float *dist_out_dev,
                                      const int64_t *out_start_dev, int32_t *count_out_dev, void *stream) {
    ISX_REQUIRE(d_pad == 32 || d_pad == 64, "d_pad must be 32 or 64 bytes (got %d)", d_pad);
    ISX_REQUIRE(n_items >= 0 && max_nq >= 0 && max_nt >= 0, "negative size");
    ISX_REQUIRE(max_nt < (1 << 18), "train set too large: cv2 asserts rows < 2^18 as well (got %d)", max_nt);
    ISX_REQUIRE((idx_out_dev == nullptr) == (dist_out_dev == nullptr), "idx_out and dist_out must be given together");
    ISX_REQUIRE(idx_out_dev == nullptr || out_start_dev != nullptr, "out_start_dev is required with idx/dist outputs");
    if (n_items == 0 || max_nq == 0) return ISX_OK;
    ISX_REQUIRE(desc_dev && items_dev, "null device pointer");
    ISX_REQUIRE(((uintptr_t)desc_dev & 15) == 0, "desc_dev must be 16-byte aligned");
    const int nqb = (max_nq + 255) / 256;
    const int per = isx_items_per_launch(nqb, 256);
    ISX_REQUIRE(per > 0, "max_nq too large (%d)", max_nq);
    hipStream_t st = (hipStream_t)stream;
    for (int off = 0; off < n_items; off += per) { // one launch unless the list is longer than a grid can be
        const int n = n_items - off < per ? n_items - off : per;
        const isx_knn_item *it = items_dev + off;
        const int64_t *os = out_start_dev ? out_start_dev + off : nullptr;
        int32_t *co = count_out_dev ? count_out_dev + off : nullptr;
        dim3 grid((unsigned)((int64_t)n * nqb));
        if (d_pad == 32)
            hipLaunchKernelGGL((knn2_hamming_kernel<8>), grid, dim3(256), 0, st, (const uint32_t *)desc_dev, it, n, nqb, ratio, idx_out_dev, dist_out_dev, os, co);
        else
            hipLaunchKernelGGL((knn2_hamming_kernel<16>), grid, dim3(256), 0, st, (const uint32_t *)desc_dev, it, n, nqb, ratio, idx_out_dev, dist_out_dev, os, co);
    }
    return isx_launch_status("knn2_hamming_kernel");
}

// ------------------------------------------------------------------------------------------------
// Exact L2 kNN-2 for byte descriptors of ANY norm (saturated AKAZE / ORB rows, dense synthetic bytes).  The MFMA
// kernels rank on the integer squared distance s, which orders like cv2's float distances only while sqrtf is
// injective on the integers in range (ISX_L2_EXACT_LIMIT).  Beyond it two different s can round to the SAME
// float32 distance, and cv2 -- which compares the float distances, `d < best[K-1]` strictly, in train order --
// then keeps the LOWER train index, whatever the integers were.  This kernel ranks on what cv2 compares: the
// key is  bits(sqrtf((float)s)) << 20 | j  (positive floats order like their bit patterns; s < 2^24 is exact in
// float32 and sqrtf is correctly rounded), so equal float distances resolve to the lowest index in both slots.
// One lane owns one query (its int8 row stays in registers), train rows stream through LDS as wave-wide
// broadcasts, a.b on v_dot4_i32_i8.  VALU-bound (no MFMA): a fallback for rare sets, ~10x the MFMA kernel's time.
// ------------------------------------------------------------------------------------------------
template <int W> // W = 32-bit words of a prepared row (8 / 16 / 32 for d_pad 32 / 64 / 128)
__global__ void __launch_bounds__(256)
knn2_l2_wide_kernel(const int32_t *__restrict__ desc, const int32_t *__restrict__ nrm, const isx_knn_item *__restrict__ items,
                    int n_items, int nqb, double ratio, int32_t *__restrict__ idx_out, float *__restrict__ dist_out,
                    const int64_t *__restrict__ out_start, int32_t *__restrict__ count_out) {
    constexpr int CHUNK = 128;
    __shared__ __align__(16) int32_t sm[CHUNK * W];
    __shared__ int32_t sn[CHUNK];
    const int item_i = blockIdx.x / nqb, qb = blockIdx.x % nqb;
    if (item_i >= n_items) return;
    const isx_knn_item it = items[item_i];
    if (qb * 256 >= it.nq) return; // uniform per block
    const int tid = threadIdx.x;
    const int qi = qb * 256 + tid;
    const bool live = qi < it.nq;
    int32_t q[W];
    {
        const v4i *p = (const v4i *)(desc + (int64_t)(it.q_start + (live ? qi : 0)) * W);
#pragma unroll
        for (int u = 0; u < W / 4; u++) {
            const v4i v = p[u];
            q[4 * u] = v[0], q[4 * u + 1] = v[1], q[4 * u + 2] = v[2], q[4 * u + 3] = v[3];
        }
    }
    const int na = nrm[it.q_start + (live ? qi : 0)];
    uint64_t m1 = ~0ull, m2 = ~0ull;
    for (int c0 = 0; c0 < it.nt; c0 += CHUNK) {
        const int rows = min(CHUNK, it.nt - c0);
        __syncthreads();
        for (int u = tid; u < rows * (W / 4); u += 256)
            ((v4i *)sm)[u] = ((const v4i *)(desc + (int64_t)(it.t_start + c0) * W))[u];
        if (tid < rows) sn[tid] = nrm[it.t_start + c0 + tid];
        __syncthreads();
        for (int r = 0; r < rows; r++) {
            const v4i *tp = (const v4i *)(sm + r * W); // same address on every lane: LDS broadcast
            int dot = 0;
#pragma unroll
            for (int u = 0; u < W / 4; u++) {
                const v4i t = tp[u];
                dot = __builtin_amdgcn_sdot4(q[4 * u], t[0], dot, false);
                dot = __builtin_amdgcn_sdot4(q[4 * u + 1], t[1], dot, false);
                dot = __builtin_amdgcn_sdot4(q[4 * u + 2], t[2], dot, false);
                dot = __builtin_amdgcn_sdot4(q[4 * u + 3], t[3], dot, false);
            }
            const int sq = na + sn[r] - 2 * dot; // exact ||a - b||^2 (the common -128 offset cancels)
            const uint64_t key = ((uint64_t)__float_as_uint(sqrtf((float)sq)) << 20) | (uint64_t)(c0 + r);
            const uint64_t lo = m1 < key ? m1 : key, hi = m1 < key ? key : m1;
            m2 = hi < m2 ? hi : m2; // second smallest of {m1, m2, key}
            m1 = lo;
        }
    }
    bool good = false;
    if (live) {
        const bool has0 = m1 != ~0ull, has1 = m2 != ~0ull;
        const float d0 = has0 ? __uint_as_float((uint32_t)(m1 >> 20)) : FLT_MAX, d1 = has1 ? __uint_as_float((uint32_t)(m2 >> 20)) : FLT_MAX;
        good = has0 && (!has1 || (double)d0 < ratio * (double)d1);
        if (idx_out) {
            const int64_t o = (out_start[item_i] + qi) * 2;
            idx_out[o] = has0 ? (int)(m1 & 0xFFFFFu) : -1;
            idx_out[o + 1] = has1 ? (int)(m2 & 0xFFFFFu) : -1;
            dist_out[o] = d0;
            dist_out[o + 1] = d1;
        }
    }
    const int cnt = __popcll(__ballot(good));
    if (count_out && (tid & 63) == 0 && cnt) atomicAdd(&count_out[item_i], cnt);
}

extern "C" int isx_knn2_l2_wide_items(const int8_t *desc_dev, const int32_t *nrm_dev, int d_pad, const isx_knn_item *items_dev,
                                      int n_items, int max_nq, int max_nt, double ratio, int32_t *idx_out_dev,
                                      float *dist_out_dev, const int64_t *out_start_dev, int32_t *count_out_dev,
                                      void *stream) {
    ISX_REQUIRE(d_pad == 32 || d_pad == 64 || d_pad == 128, "d_pad must be 32, 64 or 128 (got %d)", d_pad);
    ISX_REQUIRE(n_items >= 0 && max_nq >= 0 && max_nt >= 0, "negative size");
    ISX_REQUIRE(max_nt < (1 << 18), "train set too large: cv2 asserts rows < 2^18 as well (got %d)", max_nt);
    ISX_REQUIRE((idx_out_dev == nullptr) == (dist_out_dev == nullptr), "idx_out and dist_out must be given together");
    ISX_REQUIRE(idx_out_dev == nullptr || out_start_dev != nullptr, "out_start_dev is required with idx/dist outputs");
    if (n_items == 0 || max_nq == 0) return ISX_OK;
    ISX_REQUIRE(desc_dev && nrm_dev && items_dev, "null device pointer");
    ISX_REQUIRE(((uintptr_t)desc_dev & 15) == 0, "desc_dev must be 16-byte aligned");
    const int nqb = (max_nq + 255) / 256;
    const int per = isx_items_per_launch(nqb, 256);
    ISX_REQUIRE(per > 0, "max_nq too large (%d)", max_nq);
    hipStream_t st = (hipStream_t)stream;
    for (int off = 0; off < n_items; off += per) { // one launch unless the list is longer than a grid can be
        const int n = n_items - off < per ? n_items - off : per;
        const isx_knn_item *it = items_dev + off;
        const int64_t *os = out_start_dev ? out_start_dev + off : nullptr;
        int32_t *co = count_out_dev ? count_out_dev + off : nullptr;
        dim3 grid((unsigned)((int64_t)n * nqb));
        const int32_t *dw = (const int32_t *)desc_dev;
        if (d_pad == 32)
            hipLaunchKernelGGL((knn2_l2_wide_kernel<8>), grid, dim3(256), 0, st, dw, nrm_dev, it, n, nqb, ratio, idx_out_dev, dist_out_dev, os, co);
        else if (d_pad == 64)
            hipLaunchKernelGGL((knn2_l2_wide_kernel<16>), grid, dim3(256), 0, st, dw, nrm_dev, it, n, nqb, ratio, idx_out_dev, dist_out_dev, os, co);
        else
            hipLaunchKernelGGL((knn2_l2_wide_kernel<32>), grid, dim3(256), 0, st, dw, nrm_dev, it, n, nqb, ratio, idx_out_dev, dist_out_dev, os, co);
    }
    return isx_launch_status("knn2_l2_wide_kernel");
}

// ------------------------------------------------------------------------------------------------
// float32 L2.  Block = 4 waves x 32 queries.  D (multiple of 4, <= 128) floats per row, row-major.
// MFMA 32x32x2: lane (c = lane & 31, h = lane >> 5) feeds A[row c][k] and B[k][query c] with
// k = h * (D/2) + s at step s -- any bijection of the summation index works as long as A and B agree, and
// this one makes each lane's operands a contiguous half row (ds_read_b128 / dwordx4 friendly).
// ------------------------------------------------------------------------------------------------
struct FCand {
    float s; // na + nb - 2ab (smaller = nearer)
    int j;   // train row, -1 = absent
};
__device__ __forceinline__ bool fc_better(const FCand a, const FCand b) {
    return a.j >= 0 && (b.j < 0 || a.s < b.s || (a.s == b.s && a.j < b.j));
}

// oracle/orc_match.c l2sqr_f32 order (4 partial sums, then (s0+s1)+(s2+s3), scalar tail)
__device__ float l2sqr_direct(const float *__restrict__ a, const float *__restrict__ b, int d) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= d; k += 4) {
        const float v0 = a[k] - b[k], v1 = a[k + 1] - b[k + 1], v2 = a[k + 2] - b[k + 2], v3 = a[k + 3] - b[k + 3];
        s0 += v0 * v0, s1 += v1 * v1, s2 += v2 * v2, s3 += v3 * v3;
    }
    float s = (s0 + s1) + (s2 + s3);
    for (; k < d; k++) {
        const float v = a[k] - b[k];
        s += v * v;
    }
    return s;
}

template <int HALF> // HALF = D/2 floats handled by each lane half (multiple of 4)
__global__ void __launch_bounds__(256)
knn2_f32_kernel(const float *__restrict__ desc, const float *__restrict__ nrm, const isx_knn_item *__restrict__ items,
                int n_items, int nqb, double ratio, int32_t *__restrict__ idx_out, float *__restrict__ dist_out,
                const int64_t *__restrict__ out_start, int32_t *__restrict__ count_out) {
    constexpr int D = 2 * HALF;
    constexpr int STRIDE = D + 4; // floats; +16 B per row keeps the 16-lane ds_read_b128 groups conflict free
    constexpr int CHUNK = 32;
    __shared__ __align__(16) float sm[2][CHUNK * STRIDE + CHUNK];
    const int item_i = blockIdx.x / nqb, qb = blockIdx.x % nqb;
    if (item_i >= n_items) return;
    const isx_knn_item it = items[item_i];
    if (qb * 128 >= it.nq) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, col = lane & 31;
    const int qi = qb * 128 + wave * 32 + col;
    const bool qlive = qi < it.nq;
    const int nt = it.nt;

    v4f qf[HALF / 4];
    {
        const v4f *p = (const v4f *)(desc + (int64_t)(it.q_start + (qlive ? qi : 0)) * D + h * HALF);
#pragma unroll
        for (int u = 0; u < HALF / 4; u++) qf[u] = qlive ? p[u] : v4f{0.f, 0.f, 0.f, 0.f};
    }
    FCand g1{0.f, -1}, g2{0.f, -1};
    const int n_chunks = (nt + CHUNK - 1) / CHUNK;
    auto stage = [&](int c, int buf) {
        for (int u = tid; u < CHUNK * (D / 4); u += 256) {
            const int row = u / (D / 4), part = u % (D / 4);
            const int jl = c * CHUNK + row;
            const v4f v = jl < nt ? *(const v4f *)(desc + (int64_t)(it.t_start + jl) * D + part * 4) : v4f{0.f, 0.f, 0.f, 0.f};
            *(v4f *)(&sm[buf][row * STRIDE + part * 4]) = v;
        }
        if (tid < CHUNK) {
            const int jl = c * CHUNK + tid;
            sm[buf][CHUNK * STRIDE + tid] = jl < nt ? nrm[it.t_start + jl] : 0.f;
        }
    };
    if (n_chunks > 0) stage(0, 0);
    __syncthreads();
    const float na = qlive ? nrm[it.q_start + qi] : 0.f;
    for (int c = 0; c < n_chunks; c++) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) stage(c + 1, buf ^ 1);
        const float *base = sm[buf];
        v16f acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < HALF / 4; u++) {
            const v4f a = *(const v4f *)(base + col * STRIDE + h * HALF + 4 * u);
#pragma unroll
            for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qf[u][e], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int j = c * CHUNK + rl;
            if (j < nt) {
                const FCand cd{na + base[CHUNK * STRIDE + rl] - 2.0f * acc[r], j};
                if (fc_better(cd, g1)) {
                    g2 = g1;
                    g1 = cd;
                } else if (fc_better(cd, g2)) {
                    g2 = cd;
                }
            }
        }
        __syncthreads();
    }
    // merge the two lane halves (same query, disjoint train rows)
    FCand o1{__shfl_xor(g1.s, 32), __shfl_xor(g1.j, 32)}, o2{__shfl_xor(g2.s, 32), __shfl_xor(g2.j, 32)};
    FCand top, sec;
    if (fc_better(g1, o1) || o1.j < 0) {
        top = g1;
        sec = fc_better(g2, o1) ? g2 : o1;
    } else {
        top = o1;
        sec = fc_better(g1, o2) ? g1 : o2;
    }
    bool good = false;
    if (h == 0 && qlive) {
        const float *qp = desc + (int64_t)(it.q_start + qi) * D;
        const bool has0 = top.j >= 0, has1 = sec.j >= 0;
        float d0 = has0 ? sqrtf(l2sqr_direct(qp, desc + (int64_t)(it.t_start + top.j) * D, D)) : FLT_MAX;
        float d1 = has1 ? sqrtf(l2sqr_direct(qp, desc + (int64_t)(it.t_start + sec.j) * D, D)) : FLT_MAX;
        int j0 = top.j, j1 = sec.j;
        // the direct distances decide the final order of the two winners (cv's insertion rule)
        if (has1 && (d1 < d0 || (d1 == d0 && j1 < j0))) {
            const float td = d0;
            d0 = d1, d1 = td;
            const int tj = j0;
            j0 = j1, j1 = tj;
        }
        good = has0 && (!has1 || (double)d0 < ratio * (double)d1);
        if (idx_out) {
            const int64_t o = (out_start[item_i] + qi) * 2;
            idx_out[o] = has0 ? j0 : -1;
            idx_out[o + 1] = has1 ? j1 : -1;
            dist_out[o] = d0;
            dist_out[o + 1] = d1;
        }
    }
    const int cnt = __popcll(__ballot(good));
    if (count_out && lane == 0 && cnt) atomicAdd(&count_out[item_i], cnt);
}

// squared norms of float rows (sequential f32 sum per row: only used for ranking)
__global__ void __launch_bounds__(256) f32_norm_kernel(const float *__restrict__ desc, int64_t rows, int d,
                                                       float *__restrict__ nrm) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float *p = desc + r * d;
    float s = 0.f;
    for (int k = 0; k < d; k++) s += p[k] * p[k];
    nrm[r] = s;
}

extern "C" int isx_f32_norms(const float *desc_dev, int64_t rows, int d, float *nrm_out_dev, void *stream) {
    ISX_REQUIRE(rows >= 0 && d > 0, "bad shape");
    if (rows == 0) return ISX_OK;
    ISX_REQUIRE(desc_dev && nrm_out_dev, "null device pointer");
    hipLaunchKernelGGL(f32_norm_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, desc_dev,
                       rows, d, nrm_out_dev);
    return isx_launch_status("f32_norm_kernel");
}

extern "C" int isx_knn2_f32_items(const float *desc_dev, const float *nrm_dev, int d, const isx_knn_item *items_dev,
                                  int n_items, int max_nq, double ratio, int32_t *idx_out_dev, float *dist_out_dev,
                                  const int64_t *out_start_dev, int32_t *count_out_dev, void *stream) {
    ISX_REQUIRE(d == 32 || d == 64 || d == 128, "float32 matcher supports d = 32, 64 or 128 (got %d): zero-pad the columns", d);
    ISX_REQUIRE(n_items >= 0 && max_nq >= 0, "negative size");
    ISX_REQUIRE((idx_out_dev == nullptr) == (dist_out_dev == nullptr), "idx_out and dist_out must be given together");
    ISX_REQUIRE(idx_out_dev == nullptr || out_start_dev != nullptr, "out_start_dev is required with idx/dist outputs");
    if (n_items == 0 || max_nq == 0) return ISX_OK;
    ISX_REQUIRE(desc_dev && nrm_dev && items_dev, "null device pointer");
    ISX_REQUIRE(((uintptr_t)desc_dev & 15) == 0, "desc_dev must be 16-byte aligned");
    const int nqb = (max_nq + 127) / 128;
    const int per = isx_items_per_launch(nqb, 256);
    ISX_REQUIRE(per > 0, "max_nq too large (%d)", max_nq);
    hipStream_t st = (hipStream_t)stream;
    for (int off = 0; off < n_items; off += per) { // one launch unless the list is longer than a grid can be
        const int n = n_items - off < per ? n_items - off : per;
        const isx_knn_item *it = items_dev + off;
        const int64_t *os = out_start_dev ? out_start_dev + off : nullptr;
        int32_t *co = count_out_dev ? count_out_dev + off : nullptr;
        dim3 grid((unsigned)((int64_t)n * nqb)), block(256);
        if (d == 128)
            hipLaunchKernelGGL((knn2_f32_kernel<64>), grid, block, 0, st, desc_dev, nrm_dev, it, n, nqb, ratio, idx_out_dev, dist_out_dev, os, co);
        else if (d == 64)
            hipLaunchKernelGGL((knn2_f32_kernel<32>), grid, block, 0, st, desc_dev, nrm_dev, it, n, nqb, ratio, idx_out_dev, dist_out_dev, os, co);
        else
            hipLaunchKernelGGL((knn2_f32_kernel<16>), grid, block, 0, st, desc_dev, nrm_dev, it, n, nqb, ratio, idx_out_dev, dist_out_dev, os, co);
    }
    return isx_launch_status("knn2_f32_kernel");
}

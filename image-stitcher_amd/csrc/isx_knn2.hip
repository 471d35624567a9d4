// Exact kNN-2 (L2 over integer descriptors) + Lowe ratio test for gfx950.
//
// Replaces Stitcher.get_matches / filter_matches (reference stitcher.py:106-127, matcher built at
// stitcher.py:85 = cv2.BFMatcher() => NORM_L2) and the loop body of get_collections
// (stitcher.py:285-296).
//
// Arithmetic.  With a' = a-128, b' = b-128 (int8):  ||a-b||^2 = na' + nb' - 2 a'.b'   (exact int32).
// For a fixed query the ranking over train rows j needs only  val_j = 2 a'.b'_j - nb'_j + BIAS  (>0),
// larger = nearer.  One int8 MFMA chain gives a'.b' for a 32(train) x 32(query) tile; the train rows
// land in the 16 accumulator registers of a lane and the query on the lane, so each lane owns ONE
// running top-2 per query column block and updates it with 3 VALU ops per distance:
//       key = (acc << 9) + cbase_j|code      (v_lshl_add_u32)   key = val<<8 | code
//       m2  = med3(m1, m2, key)              (v_med3_u32)
//       m1  = max(m1, key)                   (v_max_u32)
// `code` (8 bit) orders the <=256 train rows a lane sees inside a 512-row segment so that equal
// distances resolve to the LOWEST train index, as cv2's insertion rule does; every 512 rows the
// (m1,m2) pair is folded into a (val, j) state with strict comparisons (earlier rows win ties).
// sqrtf is taken for the two survivors only; it is injective on the integer range admitted by
// ISX_L2_EXACT_LIMIT, so integer ranking == cv2's float ranking (checked by the caller).
#include "isx_common.h"
#include <float.h>
#include <type_traits>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define ISX_B2 4194304 /* 1 << 22: count kernel's cin bias, keeps h in [2^20, 2^23) for valid rows; 0 = absent */
#ifndef ISX_COUNT_WPS
#define ISX_COUNT_WPS 3 /* waves per SIMD the count kernel is compiled for */
#endif
#ifndef ISX_COUNT_WAVES
#define ISX_COUNT_WAVES 4 /* waves per workgroup of the count kernel: they share the staged train chunks */
#endif
#ifndef ISX_COUNT_NCB
#define ISX_COUNT_NCB 2 /* 32-query column blocks per wave of the count kernel (2 or 4) */
#endif
#ifndef ISX_STEP_FENCE
#define ISX_STEP_FENCE __builtin_amdgcn_sched_barrier(0)
#endif
#ifdef ISX_STEP_PATTERN /* experiment: force 1 MFMA : 5 VALU : 1 LDS read inside every step */
#define ISX_SGB1 __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#undef ISX_STEP_FENCE
#define ISX_STEP_FENCE ISX_SGB1 ISX_SGB1 ISX_SGB1 ISX_SGB1 ISX_SGB1 ISX_SGB1 ISX_SGB1 ISX_SGB1 __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef ISX_KNN_DIAG
#define ISX_KNN_DIAG 0 /* timing diagnostics, compile time only (tools/build_diag.sh); results are WRONG when != 0:
                          1 no global loads, 4 no barriers, 8 no fold, 16 no exact rescan of ambiguous queries
                          (knn2_count16w_kernel; needed with 1 / 4, whose garbage tiles make most queries ambiguous) */
#endif
#define ISX_BIAS 6291456 /* 3 << 21 : keeps val = 2dot - nb' + BIAS in [32768, 2^23] */

// ------------------------------------------------------------------------------------------------
// Descriptor preparation: (u8 | integer-valued f32) -> int8 (v-128), norms, key bases.
// One 16-byte output unit per thread; UPR = DP/16 threads cooperate on a row.
// ------------------------------------------------------------------------------------------------
template <typename T, int UPR>
__global__ void __launch_bounds__(256) desc_prep_kernel(const T *__restrict__ src, int64_t rows, int d,
                                                        int8_t *__restrict__ desc,
                                                        uint32_t *__restrict__ cbase,
                                                        int32_t *__restrict__ nrm,
                                                        int32_t *__restrict__ cin,
                                                        int32_t *__restrict__ stats) {
    constexpr int DP = UPR * 16;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t / UPR;
    const int unit = (int)(t % UPR);
    int sq_off = 0, sq_raw = 0, bad = 0;
    union {
        v4i v;
        int8_t b[16];
    } out;
    out.v = v4i{0, 0, 0, 0};
    if (row < rows) {
        const T *p = src + row * (int64_t)d;
        // 16 source values of this unit in one or four 16-byte loads when the row layout allows it
        union {
            v4i v;
            uint8_t b[16];
        } in8;
        union {
            v4i v[4];
            float f[16];
        } in32;
        const bool wide = (d & 15) == 0 && ((uintptr_t)src & 15) == 0; // then unit*16+16 <= d for every loaded unit
        if (wide && unit * 16 < d) {
            if constexpr (sizeof(T) == 1) {
                in8.v = *(const v4i *)(p + unit * 16);
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) in32.v[q] = *(const v4i *)(p + unit * 16 + 4 * q);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int k = unit * 16 + e;
            int v = 128; // padding: offset value 0
            if (k < d) {
                if constexpr (sizeof(T) == 1) {
                    v = wide ? (int)in8.b[e] : (int)p[k];
                } else {
                    const float f = wide ? in32.f[e] : (float)p[k];
                    const int iv = (int)f;
                    if (!((float)iv == f) || iv < 0 || iv > 255) {
                        bad = 1;
                        v = iv < 0 ? 0 : (iv > 255 ? 255 : iv);
                        if (!(f == f)) v = 0;
                    } else {
                        v = iv;
                    }
                }
                sq_raw += v * v;
            }
            const int o = v - 128;
            sq_off += o * o;
            out.b[e] = (int8_t)o;
        }
        *(v4i *)(desc + row * (int64_t)DP + unit * 16) = out.v;
    }
    // reduce over the UPR lanes of the row (UPR is a power of two <= 8, rows never straddle a wave)
#pragma unroll
    for (int m = 1; m < UPR; m <<= 1) {
        sq_off += __shfl_xor(sq_off, m);
        sq_raw += __shfl_xor(sq_raw, m);
    }
    if (row < rows && unit == 0) {
        nrm[row] = sq_off;
        cbase[row] = (uint32_t)(ISX_BIAS - sq_off) << 8;
        cin[row] = ISX_B2 - ((sq_off + 1) >> 1);
    }
    // stats: wave-level reduce then one atomic per wave
    int mx = row < rows ? sq_raw : 0;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int o = __shfl_xor(mx, m);
        mx = o > mx ? o : mx;
    }
    const unsigned long long anybad = __ballot(bad);
    // (a plain read first: 260k same-address atomics, one per wave, serialise to ~2.7 ms on a 256 MiB set;
    // almost every wave sees a maximum that is already recorded)
    if ((threadIdx.x & 63) == 0) {
        if (mx > __atomic_load_n(&stats[1], __ATOMIC_RELAXED)) atomicMax(&stats[1], mx);
        if (anybad) atomicOr(&stats[0], 1);
    }
}

extern "C" int isx_desc_pad(int d) {
    if (d <= 0) return isx_fail(ISX_E_ARG, "descriptor width must be positive (got %d)", d);
    if (d <= 32) return 32;
    if (d <= 64) return 64;
    if (d <= 128) return 128;
    return isx_fail(ISX_E_UNSUPPORTED, "descriptor width %d > 128 bytes is not supported by the integer L2 path", d);
}

template <typename T>
static int prep_launch(const void *src, int64_t rows, int d, int dp, int8_t *desc, uint32_t *cbase,
                       int32_t *nrm, int32_t *cin, int32_t *stats, hipStream_t st) {
    const int upr = dp / 16;
    const int64_t threads = rows * upr;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (upr == 2)
        hipLaunchKernelGGL((desc_prep_kernel<T, 2>), dim3(grid), dim3(256), 0, st, (const T *)src, rows, d, desc, cbase, nrm, cin, stats);
    else if (upr == 4)
        hipLaunchKernelGGL((desc_prep_kernel<T, 4>), dim3(grid), dim3(256), 0, st, (const T *)src, rows, d, desc, cbase, nrm, cin, stats);
    else
        hipLaunchKernelGGL((desc_prep_kernel<T, 8>), dim3(grid), dim3(256), 0, st, (const T *)src, rows, d, desc, cbase, nrm, cin, stats);
    return isx_launch_status("desc_prep_kernel");
}

extern "C" int isx_desc_prep(const void *src_dev, int src_dtype, int64_t rows, int d, int8_t *desc_dev,
                             uint32_t *cbase_dev, int32_t *nrm_dev, int32_t *cin_dev, int32_t *stats_dev,
                             void *stream) {
    const int dp = isx_desc_pad(d);
    if (dp < 0) return dp;
    ISX_REQUIRE(rows >= 0 && rows < ((int64_t)1 << 31) / 8, "rows=%lld out of range", (long long)rows);
    ISX_REQUIRE(src_dtype == ISX_U8 || src_dtype == ISX_F32, "src_dtype must be ISX_U8 or ISX_F32");
    if (rows == 0) return ISX_OK;
    ISX_REQUIRE(src_dev && desc_dev && cbase_dev && nrm_dev && cin_dev && stats_dev, "null device pointer");
    ISX_REQUIRE(((uintptr_t)desc_dev & 15) == 0, "desc_dev must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (src_dtype == ISX_U8) return prep_launch<uint8_t>(src_dev, rows, d, dp, desc_dev, cbase_dev, nrm_dev, cin_dev, stats_dev, st);
    return prep_launch<float>(src_dev, rows, d, dp, desc_dev, cbase_dev, nrm_dev, cin_dev, stats_dev, st);
}

// ------------------------------------------------------------------------------------------------
// kNN-2 + ratio kernel.  Block = 4 waves x 64 queries; train rows stream through LDS in 128-row
// chunks (three LDS buffers filled by LDS-DMA).  blockIdx -> (item, query block) is XCD aware: blocks
// that share blockIdx%8 share an XCD/L2, so item e is served entirely by XCD e%8 and the caller
// orders items so that consecutive items on one XCD share their train image.
// ------------------------------------------------------------------------------------------------
struct Cand {
    unsigned v; // val (0 = absent)
    int j;      // train row local to the item (-1 = absent)
};

__device__ __forceinline__ bool cand_better(const Cand a, const Cand b) {
    return a.v > b.v || (a.v == b.v && (unsigned)a.j < (unsigned)b.j);
}

template <int KS>
__global__ void __launch_bounds__(256, 2)
knn2_l2_kernel(const int8_t *__restrict__ desc, const uint32_t *__restrict__ cbase,
               const int32_t *__restrict__ nrm, const isx_knn_item *__restrict__ items, int n_items,
               int nqb, double ratio, int32_t *__restrict__ idx_out, float *__restrict__ dist_out,
               const int64_t *__restrict__ out_start, int32_t *__restrict__ count_out) {
    constexpr int DP = 32 * KS;
    constexpr int SLOTS = 2 * KS;
    constexpr int SH = (KS == 4) ? 1 : (KS == 2) ? 2 : 3;
    constexpr int CHUNK = 128;
    constexpr int BUF = CHUNK * DP + CHUNK * 4;
    __shared__ __align__(16) unsigned char smem[3 * BUF];

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int k = id >> 3;
    const int qb = k % nqb;
    const int item_i = (k / nqb) * 8 + xcd;
    if (item_i >= n_items) return;
    const isx_knn_item it = items[item_i];
    if (qb * 256 >= it.nq) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, col = lane & 31;
    const int q0 = qb * 256 + wave * 64;
    const int nt = it.nt;

    // query fragments (MFMA B operand): lane holds query column `col`, k bytes [32ks+16h, +16)
    v4i qf[2][KS];
#pragma unroll
    for (int cb = 0; cb < 2; cb++) {
        const int qi = q0 + cb * 32 + col;
        const v4i *p = (const v4i *)(desc + (int64_t)(it.q_start + (qi < it.nq ? qi : 0)) * DP);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) qf[cb][ks] = (qi < it.nq) ? p[2 * ks + h] : v4i{0, 0, 0, 0};
    }
    const bool cb_live[2] = {q0 < it.nq, q0 + 32 < it.nq};

    unsigned m1[2] = {0u, 0u}, m2[2] = {0u, 0u};
    Cand g1[2], g2[2];
#pragma unroll
    for (int cb = 0; cb < 2; cb++) {
        g1[cb] = Cand{0u, -1};
        g2[cb] = Cand{0u, -1};
    }

    const int n_chunks = (nt + CHUNK - 1) / CHUNK;

    // Descriptor rows go to LDS by LDS-DMA exactly as in the count kernels below (no staging registers, swizzle
    // applied on the global side, rows beyond nt zero-filled by the buffer range check); three buffers, chunk c+2
    // in flight while chunk c is consumed.  The 128 ranking codes of a chunk (cbase + position code) still pass
    // through a register: the position code is added on the way, once per row instead of once per distance.
    const __amdgpu_buffer_rsrc_t rs_desc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(desc + (int64_t)it.t_start * DP), 0, nt * DP, 0x00020000);
    constexpr int ROWS_PP = 1024 / DP;
    constexpr int PIECES = CHUNK * DP / 1024;
    constexpr int PPW = (PIECES + 3) / 4;
    const int r_in = lane / SLOTS;
    const int lane_voff = r_in * DP + (((lane % SLOTS) ^ (((wave * ROWS_PP + r_in) >> SH) & (SLOTS - 1))) << 4);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    const unsigned smem_lds = (unsigned)(size_t)smem;
    auto stage_desc = [&](int c, int bufi) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int k = 0; k < PPW; k++) {
            const int piece = wv + 4 * k;
            if (PIECES % 4 != 0 && piece >= PIECES) break;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_desc, (lds_ptr_t)(smem_lds + bufi * BUF + piece * 1024), 16, lane_voff,
                                                     c * (CHUNK * DP) + piece * 1024, 0, 0);
        }
    };
#pragma clang diagnostic pop
    unsigned st_cb = 0;
    auto load_cb = [&](int c) {
        if (tid < CHUNK) {
            const int jl = c * CHUNK + tid;
            const int tt = (c & 3) * 4 + (tid >> 5);
            const int rho = tid & 31;
            const int code = 255 - 16 * tt - ((rho & 3) + 4 * (rho >> 3));
            st_cb = (jl < nt) ? (cbase[it.t_start + jl] + (unsigned)code) : 0u;
        }
    };
    auto store_cb = [&](int bufi) {
        if (tid < CHUNK) *(unsigned *)(smem + bufi * BUF + CHUNK * DP + tid * 4) = st_cb;
    };

    if (n_chunks > 0) {
        stage_desc(0, 0);
        load_cb(0);
        store_cb(0);
    }
    if (n_chunks > 1) {
        stage_desc(1, 1);
        load_cb(1);
        store_cb(1);
    }
    __syncthreads();

    int buf = 0;
    for (int c = 0; c < n_chunks; c++) {
        const int buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf1 == 2 ? 0 : buf1 + 1;
        if (c + 2 < n_chunks) { // buf2 was last read before the previous barrier
            stage_desc(c + 2, buf2);
            load_cb(c + 2);
        }
        const unsigned char *base = smem + buf * BUF;
#pragma unroll
        for (int tl = 0; tl < 4; tl++) {
            if (c * CHUNK + tl * 32 < nt) {
                const int row = tl * 32 + col;
                v4i a[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    const int slot = 2 * ks + h;
                    a[ks] = *(const v4i *)(base + row * DP + ((slot ^ ((row >> SH) & (SLOTS - 1))) << 4));
                }
                v4i cv[4];
#pragma unroll
                for (int g = 0; g < 4; g++)
                    cv[g] = *(const v4i *)(base + CHUNK * DP + (tl * 32 + 8 * g + 4 * h) * 4);
#pragma unroll
                for (int cb = 0; cb < 2; cb++) {
                    if (cb_live[cb]) {
                        v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                        for (int ks = 0; ks < KS; ks++)
                            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks], qf[cb][ks], acc, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const unsigned key = ((unsigned)acc[r] << 9) + (unsigned)cv[r >> 2][r & 3];
                            const unsigned lo = m1[cb] < key ? m1[cb] : key;
                            const unsigned hi = m1[cb] < key ? key : m1[cb];
                            const unsigned t2 = hi < m2[cb] ? hi : m2[cb];
                            m2[cb] = lo > t2 ? lo : t2; // med3(m1, m2, key)
                            m1[cb] = hi;
                        }
                    }
                }
            }
        }
        if (c + 2 < n_chunks) store_cb(buf2);
        __syncthreads();
        buf = buf1;

        if ((c & 3) == 3 || c == n_chunks - 1) {
            const int seg = c >> 2;
#pragma unroll
            for (int cb = 0; cb < 2; cb++) {
#pragma unroll
                for (int w = 0; w < 2; w++) {
                    const unsigned key = w == 0 ? m1[cb] : m2[cb];
                    const unsigned v = key >> 8;
                    const int pos = 255 - (int)(key & 255u);
                    const int r = pos & 15;
                    const int j = seg * 512 + (pos >> 4) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (v > g1[cb].v) {
                        g2[cb] = g1[cb];
                        g1[cb] = Cand{v, j};
                    } else if (v > g2[cb].v) {
                        g2[cb] = Cand{v, j};
                    }
                }
                m1[cb] = 0u;
                m2[cb] = 0u;
            }
        }
    }

    // merge the two lane halves (same query, disjoint train rows), then lanes 0..31 finish the row
    int good_cnt = 0;
#pragma unroll
    for (int cb = 0; cb < 2; cb++) {
        Cand o1, o2;
        o1.v = __shfl_xor(g1[cb].v, 32);
        o1.j = __shfl_xor(g1[cb].j, 32);
        o2.v = __shfl_xor(g2[cb].v, 32);
        o2.j = __shfl_xor(g2[cb].j, 32);
        Cand top, sec;
        if (cand_better(g1[cb], o1)) {
            top = g1[cb];
            sec = cand_better(g2[cb], o1) ? g2[cb] : o1;
        } else {
            top = o1;
            sec = cand_better(g1[cb], o2) ? g1[cb] : o2;
        }
        const int qi = q0 + cb * 32 + col;
        bool good = false;
        if (h == 0 && qi < it.nq) {
            const int na = nrm[it.q_start + qi];
            const bool has0 = top.v != 0u, has1 = sec.v != 0u;
            const float d0 = has0 ? sqrtf((float)(na + ISX_BIAS - (int)top.v)) : FLT_MAX;
            const float d1 = has1 ? sqrtf((float)(na + ISX_BIAS - (int)sec.v)) : FLT_MAX;
            good = has0 && (!has1 || (double)d0 < ratio * (double)d1);
            if (idx_out) {
                const int64_t o = (out_start[item_i] + qi) * 2;
                idx_out[o] = has0 ? top.j : -1;
                idx_out[o + 1] = has1 ? sec.j : -1;
                dist_out[o] = d0;
                dist_out[o + 1] = d1;
            }
        }
        good_cnt += __popcll(__ballot(good));
    }
    if (count_out && lane == 0 && good_cnt) atomicAdd(&count_out[item_i], good_cnt);
}

extern "C" int isx_knn2_l2_items(const int8_t *desc_dev, const uint32_t *cbase_dev, const int32_t *nrm_dev,
                                 int d_pad, const isx_knn_item *items_dev, int n_items, int max_nq,
                                 double ratio, int32_t *idx_out_dev, float *dist_out_dev,
                                 const int64_t *out_start_dev, int32_t *count_out_dev, void *stream) {
    ISX_REQUIRE(d_pad == 32 || d_pad == 64 || d_pad == 128, "d_pad must be 32, 64 or 128 (got %d)", d_pad);
    ISX_REQUIRE(n_items >= 0 && max_nq >= 0, "negative size");
    ISX_REQUIRE((idx_out_dev == nullptr) == (dist_out_dev == nullptr), "idx_out and dist_out must be given together");
    ISX_REQUIRE(idx_out_dev == nullptr || out_start_dev != nullptr, "out_start_dev is required with idx/dist outputs");
    if (n_items == 0 || max_nq == 0) return ISX_OK;
    ISX_REQUIRE(desc_dev && cbase_dev && nrm_dev && items_dev, "null device pointer");
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
        dim3 grid((unsigned)((n + 7) / 8 * 8 * nqb)), block(256);
        if (d_pad == 128)
            hipLaunchKernelGGL((knn2_l2_kernel<4>), grid, block, 0, st, desc_dev, cbase_dev, nrm_dev, it, n, nqb, ratio, idx_out_dev, dist_out_dev, os, co);
        else if (d_pad == 64)
            hipLaunchKernelGGL((knn2_l2_kernel<2>), grid, block, 0, st, desc_dev, cbase_dev, nrm_dev, it, n, nqb, ratio, idx_out_dev, dist_out_dev, os, co);
        else
            hipLaunchKernelGGL((knn2_l2_kernel<1>), grid, block, 0, st, desc_dev, cbase_dev, nrm_dev, it, n, nqb, ratio, idx_out_dev, dist_out_dev, os, co);
    }
    return isx_launch_status("knn2_l2_kernel");
}

// ------------------------------------------------------------------------------------------------
// Count-only kNN-2 + ratio kernel: the loop body of ArbitraryStitcher.get_collections
// (reference stitcher.py:292-296 needs only len(good_matches) per ordered pair).
//
// 1.25 VALU ops per distance, no train indices.  The MFMA chain starts from C = cin_j = B2 - ceil(nb'_j/2),
// so the accumulator IS the ranking value
//       h_j = a'.b'_j - ceil(nb'_j/2) + B2 > 0,      2 a'.b'_j - nb'_j = 2 (h_j - B2) + p_j,  p_j = nb'_j & 1
// A lane owns one query and sees its train rows in 8 streams (accumulator registers r and r+8).  Two
// values (x, y) are folded at a time with three-input ops:
//       t    = med3(m[r], x[r], y[r])      second largest of {stream max so far, x, y}
//       m[r] = max3(m[r], x[r], y[r])      stream maximum
//       G    = max3(G, t[r], t[r'])        largest runner-up seen in ANY stream
// The runner-up of a stream never exceeds its maximum, so at the end
//       H1 = max_r m[r],   H2 = max(second largest of {m[r]}, G)
// are exactly the two largest h of the query.  The unknown parity bits p leave each squared distance
// known to within {s-1, s}; the ratio test is monotone in both, so it is evaluated at the two extreme
// corners and only if they disagree (~1e-4 of the queries) the wave recomputes that query exactly
// (v_dot4 over every train row).  The count is bit-identical to the index kernel's and the oracle's.
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ bool ratio_good(int s0, int s1, double ratio) {
    return (double)sqrtf((float)s0) < ratio * (double)sqrtf((float)s1);
}

__device__ __forceinline__ unsigned umax3(unsigned a, unsigned b, unsigned c) {
    const unsigned ab = a > b ? a : b;
    return ab > c ? ab : c;
}
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    const unsigned t = hi < c ? hi : c;
    return lo > t ? lo : t;
}

// Exact two largest values 2a'.b'-nb' of one query over all nt train rows, computed by the whole wave
// (INT32_MIN = absent).
template <int DP>
__device__ __noinline__ void knn2_exact_scan(const int8_t *__restrict__ desc, const int32_t *__restrict__ nrm,
                                             int q_row, int t_start, int nt, int lane, int &v1, int &v2) {
    const int *qp = (const int *)(desc + (int64_t)q_row * DP);
    v1 = INT32_MIN;
    v2 = INT32_MIN;
    for (int j = lane; j < nt; j += 64) {
        const int *tp = (const int *)(desc + (int64_t)(t_start + j) * DP);
        int dot = 0;
#pragma unroll 8
        for (int w = 0; w < DP / 4; w++) dot = __builtin_amdgcn_sdot4(qp[w], tp[w], dot, false);
        const int v = 2 * dot - nrm[t_start + j];
        if (v > v1) {
            v2 = v1;
            v1 = v;
        } else if (v > v2) {
            v2 = v;
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int o1 = __shfl_xor(v1, m), o2 = __shfl_xor(v2, m);
        const int hi = v1 > o1 ? v1 : o1, lo = v1 > o1 ? o1 : v1;
        const int s = v2 > o2 ? v2 : o2;
        v1 = hi;
        v2 = lo > s ? lo : s;
    }
}

// The same for 128-byte rows with coalesced loads: 8 lanes share a train row (16 bytes each), so one dwordx4 instruction
// reads 8 whole rows (1 KB contiguous) and 16 of them are in flight per step.  The scan above walks one row per lane
// (64 cache lines per load instruction, 8 loads in flight) and is bound by memory latency: ~350 us per 4096 rows against
// ~25 us here, which matters where many queries are recounted (knn2_pair_filter_kernel).
__device__ __noinline__ void knn2_exact_scan128(const int8_t *__restrict__ desc, const int32_t *__restrict__ nrm,
                                                int q_row, int t_start, int nt, int lane, int &v1, int &v2) {
    constexpr int UNR = 16;
    const int slot = lane & 7, sub = lane >> 3;
    const v4i q = *(const v4i *)(desc + (int64_t)q_row * 128 + slot * 16);
    v1 = INT32_MIN;
    v2 = INT32_MIN;
    for (int j0 = 0; j0 < nt; j0 += 8 * UNR) {
        v4i x[UNR];
        int nb[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int row = j0 + u * 8 + sub;
            x[u] = row < nt ? *(const v4i *)(desc + (int64_t)(t_start + row) * 128 + slot * 16) : v4i{0, 0, 0, 0};
            nb[u] = (row < nt && slot == 0) ? nrm[t_start + row] : 0;
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            int dot = __builtin_amdgcn_sdot4(q[0], x[u][0], 0, false);
            dot = __builtin_amdgcn_sdot4(q[1], x[u][1], dot, false);
            dot = __builtin_amdgcn_sdot4(q[2], x[u][2], dot, false);
            dot = __builtin_amdgcn_sdot4(q[3], x[u][3], dot, false);
            dot += __shfl_xor(dot, 1);
            dot += __shfl_xor(dot, 2);
            dot += __shfl_xor(dot, 4);
            if (slot == 0 && j0 + u * 8 + sub < nt) { // one lane per row keeps it: the merge below must see every row once
                const int v = 2 * dot - nb[u];
                if (v > v1) {
                    v2 = v1;
                    v1 = v;
                } else if (v > v2) {
                    v2 = v;
                }
            }
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int o1 = __shfl_xor(v1, m), o2 = __shfl_xor(v2, m);
        const int hi = v1 > o1 ? v1 : o1, lo = v1 > o1 ? o1 : v1;
        const int s = v2 > o2 ? v2 : o2;
        v1 = hi;
        v2 = lo > s ? lo : s;
    }
}

// KIND 0: int8 L2 (the text above).  KIND 1: Hamming distance of bit vectors on the matrix pipe -- the rows hold
// one FP4 (E2M1) nibble per bit, 0.0 or 1.0, and the chain is v_mfma_scale_f32_32x32x64_f8f6f4 (64 nibbles per
// lane pair and instruction, scales 2^0): popcount(a ^ b) = |a| + |b| - 2 a.b.  The chain starts from
// C = 512 - |b|/2 (float), so the accumulator is h = 512 + a.b - |b|/2 > 0, exact in float32 (multiples of
// 1/2 below 1024), and positive floats order like their bit patterns: the same unsigned fold serves both kinds.
// No parity ambiguity exists here, so KIND 1 has no exact-scan fallback.
template <int KS, int KIND = 0>
__global__ void __launch_bounds__(64 * ISX_COUNT_WAVES, ISX_COUNT_WPS)
knn2_count_kernel(const int8_t *__restrict__ desc, const int32_t *__restrict__ nrm,
                  const int32_t *__restrict__ cin, const isx_knn_item *__restrict__ items, int n_items, int nqb, double ratio,
                  int32_t *__restrict__ count_out, int flags) {
    constexpr int DP = 32 * KS;
    constexpr int SLOTS = 2 * KS;
    constexpr int SH = (KS == 4) ? 1 : (KS == 2) ? 2 : 3;
    constexpr int CHUNK = 128;
    constexpr int BUF = CHUNK * DP + CHUNK * 4;
    __shared__ __align__(16) unsigned char smem[3 * BUF];

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int k = id >> 3;
    const int qb = k % nqb;
    const int item_i = (k / nqb) * 8 + xcd;
    if (item_i >= n_items) return;
    const isx_knn_item it = items[item_i];
    constexpr int NCB = ISX_COUNT_NCB, NP = NCB / 2; // column blocks per wave, handled two at a time
    constexpr int WAVES = ISX_COUNT_WAVES;
    constexpr int QPW = 32 * NCB, QPB = WAVES * QPW; // queries per wave / per workgroup
    if (qb * QPB >= it.nq) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, col = lane & 31;
    const int q0 = qb * QPB + wave * QPW;
    const int nt = it.nt;
    const bool wave_live = q0 < it.nq;

    v4i qf[NCB][KS];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        const int qi = q0 + cb * 32 + col;
        const v4i *p = (const v4i *)(desc + (int64_t)(it.q_start + (qi < it.nq ? qi : 0)) * DP);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) qf[cb][ks] = (qi < it.nq) ? p[2 * ks + h] : v4i{0, 0, 0, 0};
    }

    unsigned m[NCB][8], G[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        G[cb] = 0u;
#pragma unroll
        for (int r = 0; r < 8; r++) m[cb][r] = 0u;
    }

    const int n_chunks = (nt + CHUNK - 1) / CHUNK;

    // Staging by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write, no VALU address
    // work; rows beyond nt are zero-filled by the buffer range check.  One instruction fills a 1 KiB piece
    // of LDS linearly (lane i -> byte 16 i), so the XOR swizzle of the 16-byte slots is applied on the
    // GLOBAL side: lane i fetches the slot that belongs at its LDS position.  Wave w stages pieces
    // w, w+4, ... of a chunk; their swizzle phase is the same, so one per-lane offset serves them all.
    const __amdgpu_buffer_rsrc_t rs_desc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(desc + (int64_t)it.t_start * DP), 0, nt * DP, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cin =
        __builtin_amdgcn_make_buffer_rsrc((void *)(cin + it.t_start), 0, nt * 4, 0x00020000);
    constexpr int ROWS_PP = 1024 / DP;            // rows per 1 KiB piece
    constexpr int PIECES = CHUNK * DP / 1024;
    constexpr int PPW = (PIECES + ISX_COUNT_WAVES - 1) / ISX_COUNT_WAVES; // pieces per wave and chunk
    const int r_in = lane / SLOTS;
    const int lane_voff = r_in * DP + (((lane % SLOTS) ^ (((wave * ROWS_PP + r_in) >> SH) & (SLOTS - 1))) << 4);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    const unsigned smem_lds = (unsigned)(size_t)smem;
    auto stage_chunk = [&](int c, int bufi) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int k = 0; k < PPW; k++) {
            const int piece = wv + ISX_COUNT_WAVES * k;
            if (PIECES % ISX_COUNT_WAVES != 0 && piece >= PIECES) break;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_desc, (lds_ptr_t)(smem_lds + bufi * BUF + piece * 1024), 16, lane_voff,
                                                     c * (CHUNK * DP) + piece * 1024, 0, 0);
        }
        if (wv < 2) // 128 cin values = 2 x 256 bytes
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_cin, (lds_ptr_t)(smem_lds + bufi * BUF + CHUNK * DP + wv * 256), 4,
                                                     lane * 4, c * (CHUNK * 4) + wv * 256, 0, 0);
    };
#pragma clang diagnostic pop
    // LDS fragments of one 32-row tile: the MFMA A operand (KS x 16 bytes of row `col`) and the 16 cin values
    struct Frag {
        v4i a[KS];
        v4i cv[4];
    };
    auto read_frag = [&](const unsigned char *base, int tl, Frag &f) {
        const int row = tl * 32 + col;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int slot = 2 * ks + h;
            f.a[ks] = *(const v4i *)(base + row * DP + ((slot ^ ((row >> SH) & (SLOTS - 1))) << 4));
        }
#pragma unroll
        for (int g = 0; g < 4; g++) f.cv[g] = *(const v4i *)(base + CHUNK * DP + (tl * 32 + 8 * g + 4 * h) * 4);
    };
    // one 32-row tile against the pair of query column blocks (2p, 2p+1)
    auto mfma_tile = [&](const Frag &f, int p, v16i (&acc)[2]) {
#pragma unroll
        for (int cb = 0; cb < 2; cb++) {
            v16i ac = {f.cv[0][0], f.cv[0][1], f.cv[0][2], f.cv[0][3], f.cv[1][0], f.cv[1][1], f.cv[1][2], f.cv[1][3],
                       f.cv[2][0], f.cv[2][1], f.cv[2][2], f.cv[2][3], f.cv[3][0], f.cv[3][1], f.cv[3][2], f.cv[3][3]};
            if constexpr (KIND == 0) {
#pragma unroll
                for (int ks = 0; ks < KS; ks++)
                    ac = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.a[ks], qf[2 * p + cb][ks], ac, 0, 0, 0);
            } else {
                v16f af = __builtin_bit_cast(v16f, ac);
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    const v4i &a = f.a[ks], &b = qf[2 * p + cb][ks];
                    // FP4 operands use the low 128 bits of the 256-bit operand; cbsz = blgp = 4 selects E2M1
                    af = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{a[0], a[1], a[2], a[3], 0, 0, 0, 0},
                                                                       v8i{b[0], b[1], b[2], b[3], 0, 0, 0, 0}, af, 4, 4, 0,
                                                                       0x7f7f7f7f, 0, 0x7f7f7f7f);
                }
                ac = __builtin_bit_cast(v16i, af);
            }
            acc[cb] = ac;
        }
    };
    // fold one tile into the running stream maxima / runner-up (20 VALU ops per column block)
    auto fold_tile = [&](const v16i (&acc)[2], int p) {
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++) {
            const int cb = 2 * p + c2;
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                unsigned x0 = (unsigned)acc[c2][r], y0 = (unsigned)acc[c2][r + 8];
                unsigned x1 = (unsigned)acc[c2][r + 1], y1 = (unsigned)acc[c2][r + 9];
                const unsigned t0 = umed3(m[cb][r], x0, y0);
                const unsigned t1 = umed3(m[cb][r + 1], x1, y1);
                // opaque re-definitions: keep the optimiser from sharing max(m, x) between the med3 and
                // the max3 (which would split the max3 into two v_max_u32); they emit no instruction
                asm("" : "+v"(x0));
                asm("" : "+v"(x1));
                m[cb][r] = umax3(x0, y0, m[cb][r]);
                m[cb][r + 1] = umax3(x1, y1, m[cb][r + 1]);
                G[cb] = umax3(t0, t1, G[cb]);
            }
        }
    };

    // Three LDS buffers: chunk c+2 is staged while chunk c is consumed, so chunk c+1 is already visible
    // (published by the previous barrier) and its first tile can be prefetched across the barrier.
    if (n_chunks > 0) stage_chunk(0, 0);
    if (n_chunks > 1) stage_chunk(1, 1);
    __syncthreads();

    // software pipeline: fragments of tile t+1 are read and the MFMAs of tile t+1 issued before the fold of
    // tile t, so that LDS, matrix and vector work of ONE wave overlap; accB is carried across the chunk
    // boundary (zeros = absent rows)
    v16i accA[2], accB[2];
#pragma unroll
    for (int cb = 0; cb < 2; cb++) accB[cb] = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    Frag f0, f1;
    if (wave_live) read_frag(smem, 0, f0);
    int buf = 0;
    for (int c = 0; c < n_chunks; c++) {
        const int buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf1 == 2 ? 0 : buf1 + 1;
        if (c + 2 < n_chunks && !(ISX_KNN_DIAG & 1)) stage_chunk(c + 2, buf2); // last read before the previous barrier
        const unsigned char *base = smem + buf * BUF;
        if (wave_live) {
            // the scheduling fences keep each step's live ranges inside the step (without them the
            // scheduler hoists every LDS read to the top of the chunk and spills)
            // unit u = (tile u / NP, column-block pair u % NP): accA on even units, accB on odd ones; the fold
            // of unit u-1 runs under the MFMAs of unit u (the last unit of the previous chunk for u = 0)
            if constexpr (NP == 1) {
                read_frag(base, 1, f1);
                mfma_tile(f0, 0, accA);
                fold_tile(accB, 0);
                ISX_STEP_FENCE;
                read_frag(base, 2, f0);
                mfma_tile(f1, 0, accB);
                fold_tile(accA, 0);
                ISX_STEP_FENCE;
                read_frag(base, 3, f1);
                mfma_tile(f0, 0, accA);
                fold_tile(accB, 0);
                ISX_STEP_FENCE;
                read_frag(smem + buf1 * BUF, 0, f0); // published by the previous barrier (unused after the last chunk)
                mfma_tile(f1, 0, accB);
                fold_tile(accA, 0);
                ISX_STEP_FENCE;
            } else {
#pragma unroll
                for (int tl = 0; tl < 4; tl += 2) {
                    read_frag(base, tl + 1, f1);
                    mfma_tile(f0, 0, accA);
                    fold_tile(accB, 1);
                    ISX_STEP_FENCE;
                    mfma_tile(f0, 1, accB);
                    fold_tile(accA, 0);
                    ISX_STEP_FENCE;
                    if (tl == 0)
                        read_frag(base, 2, f0);
                    else
                        read_frag(smem + buf1 * BUF, 0, f0); // published by the previous barrier
                    mfma_tile(f1, 0, accA);
                    fold_tile(accB, 1);
                    ISX_STEP_FENCE;
                    mfma_tile(f1, 1, accB);
                    fold_tile(accA, 0);
                    ISX_STEP_FENCE;
                }
            }
        }
        if (!(ISX_KNN_DIAG & 4)) __syncthreads();
        buf = buf1;
    }
    if (!wave_live) return;
    fold_tile(accB, NP - 1);

    int good_cnt = 0;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        // two largest h of this lane's half of the train rows, then of the lane pair (col, 0|1)
        unsigned b1 = 0u, b2 = 0u;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const unsigned v = m[cb][r];
            b2 = umed3(b1, b2, v); // b1 >= b2 throughout
            b1 = b1 > v ? b1 : v;
        }
        b2 = b2 > G[cb] ? b2 : G[cb]; // a stream's runner-up never exceeds b1
        const unsigned o1 = __shfl_xor(b1, 32), o2 = __shfl_xor(b2, 32);
        const unsigned H1 = b1 > o1 ? b1 : o1;
        const unsigned mn = b1 > o1 ? o1 : b1, mx2 = b2 > o2 ? b2 : o2;
        const unsigned H2 = mn > mx2 ? mn : mx2;
        const int qi = q0 + cb * 32 + col;
        const bool mine = h == 0 && qi < it.nq;
        bool good = false, amb = false;
        int na = 0;
        if (KIND == 1) {
            if (mine && H1 != 0u) {
                good = true; // single-match row: kept unconditionally (stitcher.py:125-126)
                if (H2 != 0u) {
                    const float naf = (float)nrm[it.q_start + qi];
                    const float d0 = naf - 2.f * (__uint_as_float(H1) - 512.f), d1 = naf - 2.f * (__uint_as_float(H2) - 512.f);
                    good = (double)d0 < ratio * (double)d1; // Hamming distances as float32, ratio in float64
                }
            }
        } else if (mine && H1 != 0u) {
            if (H2 == 0u) {
                good = true; // single-match row: kept unconditionally (stitcher.py:125-126)
            } else {
                na = nrm[it.q_start + qi];
                const int s0_hi = na - 2 * ((int)H1 - ISX_B2), s1_hi = na - 2 * ((int)H2 - ISX_B2);
                const int s0_lo = s0_hi > 0 ? s0_hi - 1 : 0, s1_lo = s1_hi > 0 ? s1_hi - 1 : 0;
                const bool g_best = ratio_good(s0_lo, s1_hi, ratio), g_worst = ratio_good(s0_hi, s1_lo, ratio);
                good = g_best;
                amb = (g_best != g_worst) || (flags & 1);
            }
        }
        unsigned long long am = KIND == 0 ? __ballot(amb) : 0ull;
        while (KIND == 0 && am) {
            const int L = __ffsll((long long)am) - 1;
            am &= am - 1;
            int v1, v2;
            knn2_exact_scan<DP>(desc, nrm, it.q_start + __shfl(qi, L), it.t_start, nt, lane, v1, v2);
            const int nal = __shfl(na, L);
            if (lane == L) good = ratio_good(nal - v1, nal - v2, ratio);
        }
        good_cnt += __popcll(__ballot(good));
    }
    if (lane == 0 && good_cnt) atomicAdd(&count_out[item_i], good_cnt);
}

// ------------------------------------------------------------------------------------------------
// The same count kernel on v_mfma_i32_16x16x64_i8 (128- and 64-byte rows; 32-byte rows stay on knn2_count_kernel<1>: one
// 32x32x32 instruction already covers the whole row and the kernel is bound by the fold there, not by the matrix pipe).
// KIND 1 = the FP4 Hamming rows on v_mfma_scale_f32_16x16x128_f8f6f4: built, bit-identical, and measured 5.5 % slower
// than the 32x32x64 chain (profiles/r02_shape_ab.md), so it is compiled in only as an experiment.  Same staging, same LDS layout,
// same fold arithmetic and epilogue; what changes is the tile shape: a unit is two 16-row train tiles against
// the wave's four 16-query column blocks (16 instructions of 16K MACs = the 8 x 32K of the 32x32x32 kernel),
// a lane owns query (lane & 15) of each column block and sees train rows 4 (lane >> 4) + i of a tile, so the
// four lanes that share a query merge at the end.  Per MAC the instruction writes half as many accumulator
// values as 32x32x32; tools/mfma_rate.hip ("shape") measures it 9-11 % faster under the power limit.
// ------------------------------------------------------------------------------------------------
template <int KS16, int KIND = 0>
__global__ void __launch_bounds__(64 * ISX_COUNT_WAVES, ISX_COUNT_WPS)
knn2_count16_kernel(const int8_t *__restrict__ desc, const int32_t *__restrict__ nrm, const int32_t *__restrict__ cin,
                    const isx_knn_item *__restrict__ items, int n_items, int nqb, double ratio,
                    int32_t *__restrict__ count_out, int flags) {
    // KS16 = 64-byte K steps per row: 2 for 128-byte rows (SIFT; the 256 FP4 nibbles of KIND 1), 1 for 64-byte rows (AKAZE)
    constexpr int DP = 64 * KS16, SLOTS = DP / 16, SH = KS16 == 2 ? 1 : 2, CHUNK = 128;
    static_assert(KIND == 0 || KS16 == 2, "the FP4 Hamming rows are 128 bytes");
    constexpr int BUF = CHUNK * DP + CHUNK * 4;
    __shared__ __align__(16) unsigned char smem[3 * BUF];

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int k = id >> 3;
    const int qb = k % nqb;
    const int item_i = (k / nqb) * 8 + xcd;
    if (item_i >= n_items) return;
    const isx_knn_item it = items[item_i];
    constexpr int NCB = 4, WAVES = ISX_COUNT_WAVES;
    constexpr int QPW = 16 * NCB, QPB = WAVES * QPW;
    if (qb * QPB >= it.nq) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, col = lane & 15;
    const int q0 = qb * QPB + wave * QPW;
    const int nt = it.nt;
    const bool wave_live = q0 < it.nq;

    v4i qf[NCB][KS16];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        const int qi = q0 + cb * 16 + col;
        const v4i *p = (const v4i *)(desc + (int64_t)(it.q_start + (qi < it.nq ? qi : 0)) * DP);
#pragma unroll
        for (int ks = 0; ks < KS16; ks++) qf[cb][ks] = (qi < it.nq) ? p[4 * ks + g] : v4i{0, 0, 0, 0};
    }
    unsigned m[NCB][4], G[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        G[cb] = 0u;
#pragma unroll
        for (int r = 0; r < 4; r++) m[cb][r] = 0u;
    }
    const int n_chunks = (nt + CHUNK - 1) / CHUNK;

    // staging: exactly knn2_count_kernel<4>'s (LDS-DMA, swizzle applied on the global side)
    const __amdgpu_buffer_rsrc_t rs_desc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(desc + (int64_t)it.t_start * DP), 0, nt * DP, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cin =
        __builtin_amdgcn_make_buffer_rsrc((void *)(cin + it.t_start), 0, nt * 4, 0x00020000);
    constexpr int ROWS_PP = 1024 / DP;
    constexpr int PIECES = CHUNK * DP / 1024;
    constexpr int PPW = (PIECES + ISX_COUNT_WAVES - 1) / ISX_COUNT_WAVES;
    const int r_in = lane / SLOTS;
    const int lane_voff = r_in * DP + (((lane % SLOTS) ^ (((wave * ROWS_PP + r_in) >> SH) & (SLOTS - 1))) << 4);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    const unsigned smem_lds = (unsigned)(size_t)smem;
    auto stage_chunk = [&](int c, int bufi) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int k = 0; k < PPW; k++) {
            const int piece = wv + ISX_COUNT_WAVES * k;
            if (PIECES % ISX_COUNT_WAVES != 0 && piece >= PIECES) break;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_desc, (lds_ptr_t)(smem_lds + bufi * BUF + piece * 1024), 16, lane_voff,
                                                     c * (CHUNK * DP) + piece * 1024, 0, 0);
        }
        if (wv < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_cin, (lds_ptr_t)(smem_lds + bufi * BUF + CHUNK * DP + wv * 256), 4,
                                                     lane * 4, c * (CHUNK * 4) + wv * 256, 0, 0);
    };
#pragma clang diagnostic pop
    // LDS fragments of one unit (32 train rows = two 16-row tiles): A operands and the 4 cin values per tile
    struct Frag {
        v4i a[2][KS16];
        v4i cv[2];
    };
    auto read_frag = [&](const unsigned char *base, int u, Frag &f) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int row = u * 32 + t * 16 + col;
#pragma unroll
            for (int ks = 0; ks < KS16; ks++) {
                const int slot = 4 * ks + g;
                f.a[t][ks] = *(const v4i *)(base + row * DP + ((slot ^ ((row >> SH) & (SLOTS - 1))) << 4));
            }
            f.cv[t] = *(const v4i *)(base + CHUNK * DP + (u * 32 + t * 16 + 4 * g) * 4);
        }
    };
    auto mfma_unit = [&](const Frag &f, v4i (&acc)[2][NCB]) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            // K steps outermost: the four column blocks' chains are independent, so back-to-back issue never waits
#pragma unroll
            for (int ks = 0; ks < KS16; ks++) {
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) {
                    const v4i c_in = ks == 0 ? f.cv[t] : acc[t][cb];
                    if constexpr (KIND == 0) {
                        acc[t][cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f.a[t][ks], qf[cb][ks], c_in, 0, 0, 0);
                    } else {
                        // FP4 (E2M1) operands: 128 nibbles per instruction, the low 128 bits of the 256-bit operand;
                        // cbsz = blgp = 4, scales 2^0
                        const v4i &a = f.a[t][ks], &b = qf[cb][ks];
                        const v4f r = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                            v8i{a[0], a[1], a[2], a[3], 0, 0, 0, 0}, v8i{b[0], b[1], b[2], b[3], 0, 0, 0, 0},
                            __builtin_bit_cast(v4f, c_in), 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                        acc[t][cb] = __builtin_bit_cast(v4i, r);
                    }
                }
            }
        }
    };
    // fold one unit: per column block 4 streams take (x, y) = the two tiles' values (10 VALU ops per 8 distances)
    auto fold_unit = [&](const v4i (&acc)[2][NCB]) {
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                unsigned x0 = (unsigned)acc[0][cb][r], y0 = (unsigned)acc[1][cb][r];
                unsigned x1 = (unsigned)acc[0][cb][r + 1], y1 = (unsigned)acc[1][cb][r + 1];
                const unsigned t0 = umed3(m[cb][r], x0, y0);
                const unsigned t1 = umed3(m[cb][r + 1], x1, y1);
                asm("" : "+v"(x0));
                asm("" : "+v"(x1));
                m[cb][r] = umax3(x0, y0, m[cb][r]);
                m[cb][r + 1] = umax3(x1, y1, m[cb][r + 1]);
                G[cb] = umax3(t0, t1, G[cb]);
            }
        }
    };

    if (n_chunks > 0) stage_chunk(0, 0);
    if (n_chunks > 1) stage_chunk(1, 1);
    __syncthreads();

    v4i accA[2][NCB], accB[2][NCB];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) accB[t][cb] = v4i{0, 0, 0, 0};
    Frag f0, f1;
    if (wave_live) read_frag(smem, 0, f0);
    int buf = 0;
    for (int c = 0; c < n_chunks; c++) {
        const int buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf1 == 2 ? 0 : buf1 + 1;
        if (c + 2 < n_chunks) stage_chunk(c + 2, buf2);
        const unsigned char *base = smem + buf * BUF;
        if (wave_live) {
            read_frag(base, 1, f1);
            mfma_unit(f0, accA);
            fold_unit(accB);
            ISX_STEP_FENCE;
            read_frag(base, 2, f0);
            mfma_unit(f1, accB);
            fold_unit(accA);
            ISX_STEP_FENCE;
            read_frag(base, 3, f1);
            mfma_unit(f0, accA);
            fold_unit(accB);
            ISX_STEP_FENCE;
            read_frag(smem + buf1 * BUF, 0, f0); // published by the previous barrier (unused after the last chunk)
            mfma_unit(f1, accB);
            fold_unit(accA);
            ISX_STEP_FENCE;
        }
        __syncthreads();
        buf = buf1;
    }
    if (!wave_live) return;
    fold_unit(accB);

    int good_cnt = 0;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        unsigned b1 = 0u, b2 = 0u;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const unsigned v = m[cb][r];
            b2 = umed3(b1, b2, v); // b1 >= b2 throughout
            b1 = b1 > v ? b1 : v;
        }
        b2 = b2 > G[cb] ? b2 : G[cb];
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1) { // the four lanes (col, g) of a query
            const unsigned o1 = __shfl_xor(b1, sh), o2 = __shfl_xor(b2, sh);
            const unsigned mn = b1 > o1 ? o1 : b1, mx2 = b2 > o2 ? b2 : o2;
            b1 = b1 > o1 ? b1 : o1;
            b2 = mn > mx2 ? mn : mx2;
        }
        const unsigned H1 = b1, H2 = b2;
        const int qi = q0 + cb * 16 + col;
        const bool mine = g == 0 && qi < it.nq;
        bool good = false, amb = false;
        int na = 0;
        if (KIND == 1) {
            if (mine && H1 != 0u) {
                good = true; // single-match row: kept unconditionally (stitcher.py:125-126)
                if (H2 != 0u) {
                    const float naf = (float)nrm[it.q_start + qi];
                    const float d0 = naf - 2.f * (__uint_as_float(H1) - 512.f), d1 = naf - 2.f * (__uint_as_float(H2) - 512.f);
                    good = (double)d0 < ratio * (double)d1; // Hamming distances as float32, ratio in float64
                }
            }
        } else if (mine && H1 != 0u) {
            if (H2 == 0u) {
                good = true; // single-match row: kept unconditionally (stitcher.py:125-126)
            } else {
                na = nrm[it.q_start + qi];
                const int s0_hi = na - 2 * ((int)H1 - ISX_B2), s1_hi = na - 2 * ((int)H2 - ISX_B2);
                const int s0_lo = s0_hi > 0 ? s0_hi - 1 : 0, s1_lo = s1_hi > 0 ? s1_hi - 1 : 0;
                const bool g_best = ratio_good(s0_lo, s1_hi, ratio), g_worst = ratio_good(s0_hi, s1_lo, ratio);
                good = g_best;
                amb = (g_best != g_worst) || (flags & 1);
            }
        }
        unsigned long long am = KIND == 0 ? __ballot(amb) : 0ull;
        while (KIND == 0 && am) {
            const int L = __ffsll((long long)am) - 1;
            am &= am - 1;
            int v1, v2;
            knn2_exact_scan<DP>(desc, nrm, it.q_start + __shfl(qi, L), it.t_start, nt, lane, v1, v2);
            const int nal = __shfl(na, L);
            if (lane == L) good = ratio_good(nal - v1, nal - v2, ratio);
        }
        good_cnt += __popcll(__ballot(good));
    }
    if (lane == 0 && good_cnt) atomicAdd(&count_out[item_i], good_cnt);
}

// ------------------------------------------------------------------------------------------------
// knn2_count16w_kernel: the 16x16x64 count kernel with EIGHT query column blocks per wave (128 queries) and a unit of
// ONE 16-row train tile.  Every A fragment read from LDS now feeds 8 matrix instructions instead of 4: half the LDS
// fragment reads per MAC, which on this power-limited kernel is what separates it from its register-only ingredient
// loop (profiles/r02_bidir_probe.md: 142 ns against 185 ns per unit).  The fold pairs accumulator registers (0,2) and
// (1,3) of the SAME tile -- two streams per column block instead of four -- so it still costs 5 three-input ops per
// 4 distances.  190 VGPRs: 2 waves/SIMD.  Same staging, LDS layout and epilogue as knn2_count16_kernel<2, 0>.
// Measured -5.4 to -6.0 % per launch against it (tools/count16_ab.py; 8 waves per workgroup instead of 4: +4 %, dropped).
// ------------------------------------------------------------------------------------------------
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 8 / WAVES)
knn2_count16w_kernel(const int8_t *__restrict__ desc, const int32_t *__restrict__ nrm, const int32_t *__restrict__ cin,
                     const isx_knn_item *__restrict__ items, int n_items, int nqb, double ratio,
                     int32_t *__restrict__ count_out, int flags) {
    constexpr int DP = 128, SLOTS = 8, SH = 1, CHUNK = 128;
    constexpr int BUF = CHUNK * DP + CHUNK * 4;
    __shared__ __align__(16) unsigned char smem[3 * BUF];

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int k = id >> 3;
    const int qb = k % nqb;
    const int item_i = (k / nqb) * 8 + xcd;
    if (item_i >= n_items) return;
    const isx_knn_item it = items[item_i];
    constexpr int NCB = 8;
    constexpr int QPW = 16 * NCB, QPB = WAVES * QPW;
    if (qb * QPB >= it.nq) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, col = lane & 15;
    const int q0 = qb * QPB + wave * QPW;
    const int nt = it.nt;
    const bool wave_live = q0 < it.nq;

    v4i qf[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        const int qi = q0 + cb * 16 + col;
        const v4i *p = (const v4i *)(desc + (int64_t)(it.q_start + (qi < it.nq ? qi : 0)) * DP);
#pragma unroll
        for (int ks = 0; ks < 2; ks++) qf[cb][ks] = (qi < it.nq) ? p[4 * ks + g] : v4i{0, 0, 0, 0};
    }
    unsigned m[NCB][2], G[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) G[cb] = m[cb][0] = m[cb][1] = 0u;
    const int n_chunks = (nt + CHUNK - 1) / CHUNK;

    const __amdgpu_buffer_rsrc_t rs_desc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(desc + (int64_t)it.t_start * DP), 0, nt * DP, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cin =
        __builtin_amdgcn_make_buffer_rsrc((void *)(cin + it.t_start), 0, nt * 4, 0x00020000);
    constexpr int ROWS_PP = 1024 / DP;
    constexpr int PIECES = CHUNK * DP / 1024;
    constexpr int PPW = (PIECES + WAVES - 1) / WAVES;
    const int r_in = lane / SLOTS;
    const int lane_voff = r_in * DP + (((lane % SLOTS) ^ (((wave * ROWS_PP + r_in) >> SH) & (SLOTS - 1))) << 4);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    const unsigned smem_lds = (unsigned)(size_t)smem;
    auto stage_chunk = [&](int c, int bufi) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int k = 0; k < PPW; k++) {
            const int piece = wv + WAVES * k;
            if (PIECES % WAVES != 0 && piece >= PIECES) break;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_desc, (lds_ptr_t)(smem_lds + bufi * BUF + piece * 1024), 16, lane_voff,
                                                     c * (CHUNK * DP) + piece * 1024, 0, 0);
        }
        if (wv < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_cin, (lds_ptr_t)(smem_lds + bufi * BUF + CHUNK * DP + wv * 256), 4,
                                                     lane * 4, c * (CHUNK * 4) + wv * 256, 0, 0);
    };
#pragma clang diagnostic pop
    struct Frag { // one 16-row tile: the A operand (2 K steps) and this lane's 4 cin values
        v4i a[2];
        v4i cv;
    };
    auto read_frag = [&](const unsigned char *base, int t, Frag &f) {
        const int row = t * 16 + col;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int slot = 4 * ks + g;
            f.a[ks] = *(const v4i *)(base + row * DP + ((slot ^ ((row >> SH) & (SLOTS - 1))) << 4));
        }
        f.cv = *(const v4i *)(base + CHUNK * DP + (t * 16 + 4 * g) * 4);
    };
    auto mfma_unit = [&](const Frag &f, v4i (&acc)[NCB]) {
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc[cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f.a[0], qf[cb][0], f.cv, 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc[cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f.a[1], qf[cb][1], acc[cb], 0, 0, 0);
    };
    auto fold_unit = [&](const v4i (&acc)[NCB]) {
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            unsigned x0 = (unsigned)acc[cb][0], y0 = (unsigned)acc[cb][2];
            unsigned x1 = (unsigned)acc[cb][1], y1 = (unsigned)acc[cb][3];
            const unsigned t0 = umed3(m[cb][0], x0, y0);
            const unsigned t1 = umed3(m[cb][1], x1, y1);
            asm("" : "+v"(x0));
            asm("" : "+v"(x1));
            m[cb][0] = umax3(x0, y0, m[cb][0]);
            m[cb][1] = umax3(x1, y1, m[cb][1]);
            G[cb] = umax3(t0, t1, G[cb]);
        }
    };

    if (n_chunks > 0) stage_chunk(0, 0);
    if (n_chunks > 1) stage_chunk(1, 1);
    __syncthreads();

    v4i accA[NCB], accB[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) accB[cb] = v4i{0, 0, 0, 0};
    Frag f0, f1;
    if (wave_live) read_frag(smem, 0, f0);
    int buf = 0;
    for (int c = 0; c < n_chunks; c++) {
        const int buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf1 == 2 ? 0 : buf1 + 1;
        if (c + 2 < n_chunks && !(ISX_KNN_DIAG & 1)) stage_chunk(c + 2, buf2);
        const unsigned char *base = smem + buf * BUF;
        if (wave_live) {
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
                read_frag(base, t + 1, f1);
                mfma_unit(f0, accA);
                if (!(ISX_KNN_DIAG & 8)) fold_unit(accB);
                ISX_STEP_FENCE;
                if (t + 2 < 8)
                    read_frag(base, t + 2, f0);
                else
                    read_frag(smem + buf1 * BUF, 0, f0); // published by the previous barrier (unused after the last chunk)
                mfma_unit(f1, accB);
                if (!(ISX_KNN_DIAG & 8)) fold_unit(accA);
                ISX_STEP_FENCE;
            }
        }
        if (!(ISX_KNN_DIAG & 4)) __syncthreads();
        buf = buf1;
    }
    if (!wave_live) return;
    fold_unit(accB);

    int good_cnt = 0;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        unsigned b1 = m[cb][0] > m[cb][1] ? m[cb][0] : m[cb][1];
        unsigned b2 = m[cb][0] > m[cb][1] ? m[cb][1] : m[cb][0];
        b2 = b2 > G[cb] ? b2 : G[cb];
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1) { // the four lanes (col, g) of a query
            const unsigned o1 = __shfl_xor(b1, sh), o2 = __shfl_xor(b2, sh);
            const unsigned mn = b1 > o1 ? o1 : b1, mx2 = b2 > o2 ? b2 : o2;
            b1 = b1 > o1 ? b1 : o1;
            b2 = mn > mx2 ? mn : mx2;
        }
        const unsigned H1 = b1, H2 = b2;
        const int qi = q0 + cb * 16 + col;
        const bool mine = g == 0 && qi < it.nq;
        bool good = false, amb = false;
        int na = 0;
        if (mine && H1 != 0u) {
            if (H2 == 0u) {
                good = true; // single-match row: kept unconditionally (stitcher.py:125-126)
            } else {
                na = nrm[it.q_start + qi];
                const int s0_hi = na - 2 * ((int)H1 - ISX_B2), s1_hi = na - 2 * ((int)H2 - ISX_B2);
                const int s0_lo = s0_hi > 0 ? s0_hi - 1 : 0, s1_lo = s1_hi > 0 ? s1_hi - 1 : 0;
                const bool g_best = ratio_good(s0_lo, s1_hi, ratio), g_worst = ratio_good(s0_hi, s1_lo, ratio);
                good = g_best;
                amb = (g_best != g_worst) || (flags & 1);
            }
        }
        unsigned long long am = (ISX_KNN_DIAG & 16) ? 0ull : __ballot(amb);
        while (am) {
            const int L = __ffsll((long long)am) - 1;
            am &= am - 1;
            int v1, v2;
            knn2_exact_scan<DP>(desc, nrm, it.q_start + __shfl(qi, L), it.t_start, nt, lane, v1, v2);
            const int nal = __shfl(na, L);
            if (lane == L) good = ratio_good(nal - v1, nal - v2, ratio);
        }
        good_cnt += __popcll(__ballot(good));
    }
    if (lane == 0 && good_cnt) atomicAdd(&count_out[item_i], good_cnt);
}

extern "C" int isx_knn2_l2_count_items(const int8_t *desc_dev, const int32_t *nrm_dev, const int32_t *cin_dev,
                                       int d_pad, const isx_knn_item *items_dev, int n_items, int max_nq,
                                       double ratio, int32_t *count_out_dev, int flags, void *stream) {
    ISX_REQUIRE(d_pad == 32 || d_pad == 64 || d_pad == 128, "d_pad must be 32, 64 or 128 (got %d)", d_pad);
    ISX_REQUIRE(n_items >= 0 && max_nq >= 0, "negative size");
    if (n_items == 0 || max_nq == 0) return ISX_OK;
    ISX_REQUIRE(desc_dev && nrm_dev && cin_dev && items_dev && count_out_dev, "null device pointer");
    ISX_REQUIRE(((uintptr_t)desc_dev & 15) == 0, "desc_dev must be 16-byte aligned");
    // 128-byte rows run on knn2_count16w_kernel (128 queries per wave); flags bit 2 selects its predecessor
    // knn2_count16_kernel<2> (64 queries per wave), bit 1 the 32x32x32 kernel: the A/B references
    const bool wide = d_pad == 128 && !(flags & 6);
    const int qpb = wide ? 16 * 8 * ISX_COUNT_WAVES : 32 * ISX_COUNT_WAVES * ISX_COUNT_NCB; // queries per workgroup
    const int nqb = (max_nq + qpb - 1) / qpb;
    const int per = isx_items_per_launch(nqb, 64 * ISX_COUNT_WAVES);
    ISX_REQUIRE(per > 0, "max_nq too large (%d)", max_nq);
    hipStream_t st = (hipStream_t)stream;
    for (int off = 0; off < n_items; off += per) { // one launch unless the list is longer than a grid can be
        const int n = n_items - off < per ? n_items - off : per;
        const isx_knn_item *it = items_dev + off;
        int32_t *co = count_out_dev + off;
        dim3 grid((unsigned)((n + 7) / 8 * 8 * nqb)), block(64 * ISX_COUNT_WAVES);
        if (wide)
            hipLaunchKernelGGL((knn2_count16w_kernel<ISX_COUNT_WAVES>), grid, block, 0, st, desc_dev, nrm_dev, cin_dev, it, n, nqb, ratio, co, flags);
        else if (d_pad == 128 && !(flags & 2))
            hipLaunchKernelGGL((knn2_count16_kernel<2>), grid, block, 0, st, desc_dev, nrm_dev, cin_dev, it, n, nqb, ratio, co, flags);
        else if (d_pad == 64 && !(flags & 2))
            hipLaunchKernelGGL((knn2_count16_kernel<1>), grid, block, 0, st, desc_dev, nrm_dev, cin_dev, it, n, nqb, ratio, co, flags);
        else if (d_pad == 128)
            hipLaunchKernelGGL((knn2_count_kernel<4>), grid, block, 0, st, desc_dev, nrm_dev, cin_dev, it, n, nqb, ratio, co, flags);
        else if (d_pad == 64)
            hipLaunchKernelGGL((knn2_count_kernel<2>), grid, block, 0, st, desc_dev, nrm_dev, cin_dev, it, n, nqb, ratio, co, flags);
        else
            hipLaunchKernelGGL((knn2_count_kernel<1>), grid, block, 0, st, desc_dev, nrm_dev, cin_dev, it, n, nqb, ratio, co, flags);
    }
    return isx_launch_status("knn2_count_kernel");
}

// ------------------------------------------------------------------------------------------------
// knn2_pair_filter_kernel (round 3): BOTH directions of an unordered image pair from ONE distance matrix -- half the
// matrix instructions of the two-pass count kernels above.  reference stitcher.py:285-296 visits (i, j) and (j, i);
// both need the same 4096 x 4096 products.
//
// What made "both directions from one tile" lose in round 2 (profiles/r02_bidir_probe.md) was the vector work of two
// EXACT top-2 folds (3.5 three-input ops per value).  This kernel does not rank exactly.  It keeps, per direction, only
// MAXIMA OF GROUPS of candidates (0.5 op per value and direction) and proves from them that a query is certainly NOT a
// good match, which is what almost every query of almost every pair of get_collections is:
//     H1      = the largest group maximum          = the query's best value, exactly
//     H2lo    = the second largest group maximum  <= its true second best value (two distinct candidates reach it)
//     good needs d1 < ratio * d2; with d2 <= d(H2lo) the test can only get harder, so "not good at (d(H1), d(H2lo))"
//     is final.  Whoever passes is a CANDIDATE: up to ISX_PAIR_CAP per direction are recounted exactly, 16 per wave and
//     walk of the other image on the matrix pipe (knn2_exact_recount16); a direction with more (overlapping images) is
//     reported as -1 and the caller runs it on the two-pass kernel.  Counts are therefore the exact counts, bit for bit,
//     whatever the filter lets through.
// Column direction (query = item.q image, train = item.t image): as in knn2_count16w_kernel the train norm rides in the
// MFMA's C input, a lane owns 8 x 16 queries, two max-only streams per column block.  Row direction (query = a row of
// item.t, candidates = the rows of item.q): the value that ranks along a row is raw - n_q/2; instead of one subtraction
// per value the element-wise maximum over the wave's 8 column blocks (128 queries = one CLASS) is corrected ONCE by the
// class's smallest half norm (an upper bound) and tagged with the class, reduced 16 lanes x 4 registers -> 1 register
// (a reduce-scatter over the quads, six DPP ops) and folded into a per-row table in LDS with one ds_max_u32; the class's norm spread is taken off again where a LOWER bound is
// needed.  Rows sorted by norm inside an image (ops.sort_rows_by_norm; counts do not depend on the
// row order) make that spread small; unsorted input is still exact, only more candidates pass.
// One workgroup (8 waves, 2 per SIMD) per pair, the query image in 4 passes of 1024; the train image streams through the
// four LDS-DMA buffers once per pass.
// ------------------------------------------------------------------------------------------------
#ifndef ISX_PAIR_CAP
#define ISX_PAIR_CAP 128
#endif
#ifndef ISX_PAIR_SPREAD
#define ISX_PAIR_SPREAD 16384
#endif /* norm spread of a class (128 query rows) beyond which it takes the per-value correction */
#ifndef ISX_PAIR_T
#define ISX_PAIR_T 0 /* compile-time timing diagnostics (tools/build_variant.sh), results WRONG when != 0: 4 no exact recounts,
                        16 report the candidate counts, 32 exactly 16 recounts per direction, 64 no chunk barrier */
#endif
#define ISX_PAIR_MAX_ROWS 4096   /* rows of either image for the four-group table (NG = 4) */
#define ISX_PAIR_BIG_TRAIN 8192  /* NG = 2: rows of the smaller image of a pair (it becomes the train side) ... */
#define ISX_PAIR_BIG_QUERY 16384 /* ... and of the larger one (128 classes of 128 query rows) */
#ifndef ISX_PAIR_X
#define ISX_PAIR_X 0 /* timing experiments on the step (WRONG results when != 0): 1 no sched_barrier per group, 2 no LDS atomics,
                        4 no DPP lane reduction, 8 no row-direction maxima, 16 the waves' LDS atomics land on different rows,
                        32 no LDS fragment reads, 64 no column epilogue (the compiler then drops the column folds of the loop as well:
                        this measures both), 128 the first pass's query rows in every pass */
#endif
#define ISX_B3 (1 << 22) /* bias of the packed row-direction bound: keeps it positive */
#define ISX_PAIR_NBUF 4 /* staging buffers: the chunk being read, the next one (its first tile is read a step early), two in flight */
#define ISX_PAIR_LDS (ISX_PAIR_NBUF * (128 * 128 + 128 * 4) + 256 + ISX_PAIR_MAX_ROWS * 16 + 128 * 2 * 4 + (4 + 2 * ISX_PAIR_CAP) * 4)

__device__ __forceinline__ unsigned dpp_quad_max(unsigned v) { // maximum over each group of 4 lanes, in every lane
    unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true); // quad_perm [1,0,3,2]
    v = v > o ? v : o;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);          // quad_perm [2,3,0,1]
    return v > o ? v : o;
}

template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
// minimum / maximum over each row of 16 lanes, in every lane of the row (quad_perm, quad_perm, row_half_mirror, row_mirror)
__device__ __forceinline__ int row16_min(int v) {
    int o = dpp_i32<0xB1>(v); v = o < v ? o : v;
    o = dpp_i32<0x4E>(v); v = o < v ? o : v;
    o = dpp_i32<0x141>(v); v = o < v ? o : v;
    o = dpp_i32<0x140>(v); return o < v ? o : v;
}
__device__ __forceinline__ int row16_max(int v) {
    int o = dpp_i32<0xB1>(v); v = o > v ? o : v;
    o = dpp_i32<0x4E>(v); v = o > v ? o : v;
    o = dpp_i32<0x141>(v); v = o > v ? o : v;
    o = dpp_i32<0x140>(v); return o > v ? o : v;
}
// (largest, second largest) of the values the four lanes {l, l ^ 16, l ^ 32, l ^ 48} hold as (b1 >= b2) pairs, in each of them:
// two v_permlane*_swap exchanges instead of four LDS-crossbar shuffles
__device__ __forceinline__ void top2_over_rows(unsigned &b1, unsigned &b2) {
    {
        const auto p = __builtin_amdgcn_permlane16_swap(b1, b1, false, false); // {own, partner} in some order
        const auto q = __builtin_amdgcn_permlane16_swap(b2, b2, false, false);
        const unsigned mn = p[0] < p[1] ? p[0] : p[1], mx2 = q[0] > q[1] ? q[0] : q[1];
        b1 = p[0] > p[1] ? p[0] : p[1];
        b2 = mn > mx2 ? mn : mx2;
    }
    {
        const auto p = __builtin_amdgcn_permlane32_swap(b1, b1, false, false);
        const auto q = __builtin_amdgcn_permlane32_swap(b2, b2, false, false);
        const unsigned mn = p[0] < p[1] ? p[0] : p[1], mx2 = q[0] > q[1] ? q[0] : q[1];
        b1 = p[0] > p[1] ? p[0] : p[1];
        b2 = mn > mx2 ? mn : mx2;
    }
}
// A float that is certainly above ratio^2: (float)s0 > (float)s1 * ratio2_up(ratio) implies !ratio_good(s0, s1, ratio) --
// sqrtf, the two int -> float conversions and the float product are each off by at most 2^-24 relatively, the double product
// by 2^-53; 2^-19 covers them with room.  (A NaN ratio compares false: the exact test decides.)
__device__ __forceinline__ float ratio2_up(double ratio) { return __double2float_ru(ratio * ratio * (1.0 + 0x1p-19)); }

// Exact recount of up to 16 candidate queries by ONE wave on the matrix pipe: the wave walks the other image once (16-row
// tiles straight from global memory, 8 tiles in flight) and keeps, per query, the exact two largest values 2 a.b - n_b of its
// rows -- the same values, hence the same verdicts, as knn2_exact_scan128, which walks the image once PER QUERY: 512 KB each,
// and with eight waves of a CU recounting at once that walk is bound by the CU's load path (~90 us per query measured, 3 ms for
// a pair with two full candidate lists).  qlist: up to 16 query rows (indices inside the query image, in LDS).  Returns the
// number of good matches among them (wave-uniform).
struct __attribute__((packed, aligned(4))) isx_i4u { int v[4]; };
template <int KS> // 64-byte K steps of a row: 2 = 128-byte rows, 1 = 64-byte rows
__device__ __noinline__ int knn2_exact_recount16(const int8_t *__restrict__ desc, const int32_t *__restrict__ nrm, const int *qlist,
                                                 int n_q, int q_start, int t_start, int nt, int lane, double ratio) {
    constexpr int PF = 8;
    const int col = lane & 15, g = lane >> 4;
    const bool have = col < n_q;
    const int qrow = q_start + qlist[have ? col : 0];
    constexpr int DP = 64 * KS;
    const v4i *qp = (const v4i *)(desc + (int64_t)qrow * DP);
    const v4i b0 = qp[g], b1 = KS == 2 ? qp[4 + g] : v4i{0, 0, 0, 0};
    const int n_tiles = (nt + 15) >> 4;
    v4i a0[PF], a1[PF];
    isx_i4u nb[PF];
    auto load = [&](int t, int s) {
        const int row = 16 * t + col;
        const v4i *tp = (const v4i *)(desc + (int64_t)(t_start + (row < nt ? row : nt - 1)) * DP);
        a0[s] = tp[g];
        if (KS == 2) a1[s] = tp[4 + g];
        const int r0 = 16 * t + 4 * g; // this lane's four output rows
        if (r0 + 3 < nt)
            nb[s] = *(const isx_i4u *)(nrm + t_start + r0);
        else // the image's last rows: one guarded load per row (never past the image)
#pragma unroll
            for (int i = 0; i < 4; i++) nb[s].v[i] = r0 + i < nt ? nrm[t_start + r0 + i] : 0;
    };
#pragma unroll
    for (int s = 0; s < PF; s++)
        if (s < n_tiles) load(s, s);
    int m1 = INT32_MIN, m2 = INT32_MIN;
    for (int t0 = 0; t0 < n_tiles; t0 += PF) {
#pragma unroll
        for (int s = 0; s < PF; s++) {
            const int t = t0 + s;
            if (t >= n_tiles) break;
            v4i acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[s], b0, v4i{0, 0, 0, 0}, 0, 0, 0);
            if (KS == 2) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[s], b1, acc, 0, 0, 0);
            const int r0 = 16 * t + 4 * g;
            const int nbv[4] = {nb[s].v[0], nb[s].v[1], nb[s].v[2], nb[s].v[3]};
            if (t + PF < n_tiles) load(t + PF, s);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int v = (r0 + i < nt) ? 2 * acc[i] - nbv[i] : INT32_MIN;
                const int lo = m1 < v ? m1 : v;
                m1 = m1 < v ? v : m1;
                m2 = m2 < lo ? lo : m2;
            }
        }
    }
    // merge the four row groups of a column (signed values through the unsigned helper: x ^ 0x80000000 keeps the order)
    unsigned u1 = (unsigned)m1 ^ 0x80000000u, u2 = (unsigned)m2 ^ 0x80000000u;
    top2_over_rows(u1, u2);
    const int v1 = (int)(u1 ^ 0x80000000u), v2 = (int)(u2 ^ 0x80000000u);
    bool good = false;
    if (g == 0 && have) {
        const int na = nrm[qrow];
        good = (v1 != INT32_MIN) && (v2 == INT32_MIN || ratio_good(na - v1, na - v2, ratio));
    }
    return __builtin_popcountll(__builtin_amdgcn_ballot_w64(good));
}

// NG = groups per row of the table.  4: the kernel of BASELINE's 4096-keypoint images.  2: images beyond 4096 rows -- the table
// keeps two groups per row (8 bytes: 8192 train rows in the same 64 KB; the second largest group maximum is a somewhat weaker
// lower bound, so a few more candidates are recounted), the class tag takes 7 bits (128 classes = 16384 query rows), and the
// smaller image of a pair is made the train side (results swapped back); a pair beyond those sizes is reported as (-1, -1).
// KS = 64-byte K steps of a descriptor row: 2 = 128-byte rows (SIFT), 1 = 64-byte rows (AKAZE's 61 bytes, padded): one matrix
// instruction per tile and column block, the same folds.
#ifdef ISX_PAIR_STAMP /* timing diagnostic: shader cycles of wave 0 per phase, summed over the workgroups (tools/pairs_diag.py) */
__device__ unsigned long long isx_pair_stamp[16];
extern "C" int isx_pair_stamp_read(unsigned long long *out16, int reset) {
    ISX_HIP_TRY(hipDeviceSynchronize());
    ISX_HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(isx_pair_stamp), sizeof(isx_pair_stamp)));
    if (reset) {
        unsigned long long z[16] = {};
        ISX_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(isx_pair_stamp), z, sizeof(z)));
    }
    return ISX_OK;
}
#define ISX_STAMP(k)                                                                               \
    do {                                                                                           \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                              \
        if (tid == 0) atomicAdd(&isx_pair_stamp[k], now_ - stamp_t);                               \
        stamp_t = now_;                                                                            \
    } while (0)
#else
#define ISX_STAMP(k)
#endif
template <int WAVES, int NG, int KS>
__global__ void __launch_bounds__(64 * WAVES)
knn2_pair_filter_kernel(const int8_t *__restrict__ desc, const int32_t *__restrict__ nrm, const int32_t *__restrict__ cin,
                        const isx_knn_item *__restrict__ items, int n_items, double ratio,
                        int32_t *__restrict__ count_out, int flags) {
    static_assert(KS == 1 || KS == 2, "64- or 128-byte rows");
    constexpr int DP = 64 * KS, SLOTS = DP / 16, SH = KS == 2 ? 1 : 2, CHUNK = 128, NCB = 8;
    constexpr int BUF = CHUNK * DP + CHUNK * 4;
    constexpr int QPW = 16 * NCB, QPB = WAVES * QPW;
    extern __shared__ __align__(16) unsigned char smem[];
    // (one dummy tile in front: the software pipeline's first fold runs on an all-zero accumulator of "tile -1")
    constexpr int NBUF = ISX_PAIR_NBUF;
    static_assert(NG == 4 || NG == 2, "groups per table row");
    constexpr int TAGB = NG == 4 ? 5 : 7, NCLS = 1 << TAGB; // class tag bits, classes (of 128 query rows)
    constexpr int ROWB = NG * 4;                            // table bytes per train row
    unsigned *rowtab = (unsigned *)(smem + NBUF * BUF + 256);           // [rows][NG]: packed (bound << TAGB | class)
    int *cls = (int *)(smem + NBUF * BUF + 256 + ISX_PAIR_MAX_ROWS * 16);     // [NCLS][2]: bound offset, 2 floor(nmin/2) - nmax
    int *lst = cls + 2 * 128;                                        // n_col, n_row, good_col, good_row, col list, row list

    const int item_i = blockIdx.x;
    if (item_i >= n_items) return;
    isx_knn_item it = items[item_i];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, col = lane & 15;
#ifdef ISX_PAIR_STAMP
    unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
    if (tid == 0) atomicAdd(&isx_pair_stamp[15], 1ull);
#endif
    bool swapped = false;
    if (NG == 2) {
        if (it.nt > it.nq) { // the smaller image is the train side: its rows are what the table holds
            const isx_knn_item o = it;
            it.q_start = o.t_start, it.nq = o.nt, it.t_start = o.q_start, it.nt = o.nq;
            swapped = true;
        }
        if (it.nt > ISX_PAIR_BIG_TRAIN || it.nq > ISX_PAIR_BIG_QUERY) { // (the caller runs both directions on the two-pass kernel)
            if (tid == 0) count_out[2 * (int64_t)item_i] = count_out[2 * (int64_t)item_i + 1] = -1;
            return;
        }
    }
    const int nt = it.nt, nq = it.nq;
    const int n_chunks = (nt + CHUNK - 1) / CHUNK;
    const int n_pass = (nq + QPB - 1) / QPB;
    const float r2_up = ratio2_up(ratio);

    for (int i = tid; i < n_chunks * CHUNK * ROWB / 16; i += 64 * WAVES) ((uint4 *)rowtab)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 4) lst[tid] = 0;

    const __amdgpu_buffer_rsrc_t rs_desc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(desc + (int64_t)it.t_start * DP), 0, nt * DP, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cin =
        __builtin_amdgcn_make_buffer_rsrc((void *)(cin + it.t_start), 0, nt * 4, 0x00020000);
    constexpr int ROWS_PP = 1024 / DP;
    constexpr int PIECES = CHUNK * DP / 1024;
    constexpr int PPW = (PIECES + WAVES - 1) / WAVES;
    const int r_in = lane / SLOTS;
    const int lane_voff = r_in * DP + (((lane % SLOTS) ^ (((wave * ROWS_PP + r_in) >> SH) & (SLOTS - 1))) << 4);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    const unsigned smem_lds = (unsigned)(size_t)smem;
    auto stage_chunk = [&](int c, int bufi) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int k = 0; k < PPW; k++) {
            const int piece = wv + WAVES * k;
            if (PIECES % WAVES != 0 && piece >= PIECES) break;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_desc, (lds_ptr_t)(smem_lds + bufi * BUF + piece * 1024), 16, lane_voff,
                                                     c * (CHUNK * DP) + piece * 1024, 0, 0);
        }
        // every wave loads one of the two cin pieces (the same bytes up to four times): each wave then has exactly
        // PPW + 1 DMA instructions per stage in flight, which the counted wait before the chunk barrier relies on
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_cin, (lds_ptr_t)(smem_lds + bufi * BUF + CHUNK * DP + (wv & 1) * 256), 4,
                                                 lane * 4, c * (CHUNK * 4) + (wv & 1) * 256, 0, 0);
    };
    static_assert(PIECES % WAVES == 0 && PPW + 1 == KS + 1, "the counted waits below assume KS + 1 DMA instructions per wave and stage");
    auto stage_first = [&]() { // the chunks a pass starts on
#pragma unroll
        for (int c = 0; c < NBUF - 1; c++)
            if (c < n_chunks) stage_chunk(c, c);
    };
#pragma clang diagnostic pop
    struct Frag {
        v4i a[2];
        v4i cv;
    };
    // Fragment addresses: row = 16 t + col and the XOR swizzle ((row >> 1) & 7) = ((col >> 1) & 7) does not depend on the
    // tile, so a lane's three addresses are (buffer base + a per-lane constant) + an immediate per tile -- no address
    // arithmetic in the loop.
    const unsigned fo0 = col * DP + (((unsigned)g ^ ((col >> SH) & (SLOTS - 1))) << 4);
    const unsigned fo1 = col * DP + (((unsigned)(4 + g) ^ ((col >> SH) & (SLOTS - 1))) << 4);
    const unsigned foc = CHUNK * DP + 16 * g;
    auto read_frag = [&](const unsigned char *base, auto tc, Frag &f) {
        constexpr int t = decltype(tc)::value;
        if (ISX_PAIR_X & 32) { // (timing diagnostic: no fragment reads)
            asm volatile("" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.cv));
            return;
        }
        f.a[0] = *(const v4i *)(base + fo0 + t * (16 * DP));
        if (KS == 2) f.a[1] = *(const v4i *)(base + fo1 + t * (16 * DP));
        if (!(ISX_PAIR_X & 256) || t == 0) f.cv = *(const v4i *)(base + foc + t * 64); // (256: timing diagnostic, one tile's norms for all)
    };
    // this lane's slot of the row table in tile 0 of chunk 0: row 4g + its quad (lane >> 2) & 3, group = its position lane & 3
    // as an LDS byte address, one tile back (the dummy tile): the fold adds (tile + 1) * 256 + 16 r as an immediate
    // (NG == 2: group (lane >> 1) & 1, written by the lanes at even positions only)
    const unsigned row_slot_lds = smem_lds + NBUF * BUF + 256 - 16 * ROWB + (4 * g + ((lane >> 2) & 3)) * ROWB + (NG == 4 ? lane & 3 : (lane >> 1) & 1) * 4;

    v4i qf[NCB][2];
    unsigned phantom = 0; // bit cb: this lane's column of block cb is beyond nq
    // norms of this lane's 8 queries, for the pass that runs (na) and the one whose rows are already requested (na_nx):
    // every global load of a pass is issued in one batch a pass ahead -- eight dependent loads in the column epilogue,
    // and two more for the class's norm range, were 13 us of every pass (profiles/r03_knn2_count_pmc.md)
    int na[NCB], na_nx[NCB];
    auto load_queries = [&](int pass) {
        const int q0 = pass * QPB + wave * QPW;
        phantom = 0;
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            const int qi = q0 + cb * 16 + col;
            const v4i *p = (const v4i *)(desc + (int64_t)(it.q_start + (qi < nq ? qi : 0)) * DP);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) qf[cb][ks] = (qi < nq) ? p[4 * ks + g] : v4i{0, 0, 0, 0};
            na_nx[cb] = qi < nq ? nrm[it.q_start + qi] : -1;
            if (qi >= nq) phantom |= 1u << cb;
        }
    };
#ifdef ISX_PAIR_PRIO /* experiment: static priority for the second-dispatched half of the waves */
    if (wave >= WAVES / 2) __builtin_amdgcn_s_setprio(ISX_PAIR_PRIO);
#endif
    load_queries(0);
    stage_first();
    ISX_STAMP(0); // prologue: item, table clear, first requests

    for (int pass = 0; pass < n_pass; pass++) {
        const int q0 = pass * QPB + wave * QPW;
        const bool wave_live = q0 < nq;
        const int cls_id = pass * WAVES + wave;
        // 0: the common class.  1: wide norm spread -- every value is corrected by its own query's half norm (one
        // subtraction per value).  2: the last, partly filled class of an image -- as 1, and the columns beyond nq are
        // kept out of the row table (a select per value).  1 and 2 are kept out of the common loop.
        int mode = q0 + QPW > nq ? 2 : 0;
        // the class's norm range (whatever the row order is: sortedness only makes it narrow)
        int kc = 0;
        unsigned dq[NCB]; // modes 1, 2: this lane's query of block cb has half norm floor(nmin / 2) + dq
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) na[cb] = na_nx[cb];
        if (wave_live) {
            int nmin = INT32_MAX, nmax = 0;
#pragma unroll
            for (int cb = 0; cb < NCB; cb++)
                if (na[cb] >= 0) {
                    nmin = na[cb] < nmin ? na[cb] : nmin;
                    nmax = na[cb] > nmax ? na[cb] : nmax;
                }
            nmin = row16_min(nmin), nmax = row16_max(nmax); // (the 16 columns; the four row groups g hold the same values)
            const int off = ISX_B3 - (nmin >> 1);
            kc = __builtin_amdgcn_readfirstlane((off << TAGB) | cls_id);
            // a class whose norms spread widely (the two tails of a sorted image, every class of an unsorted one) would
            // make every row a candidate: it takes the per-value correction instead, which is exact to 1
            if (mode == 0 && __builtin_amdgcn_readfirstlane(nmax - nmin) > ISX_PAIR_SPREAD) mode = 1;
            if (lane == 0) {
                cls[2 * cls_id] = off;
                cls[2 * cls_id + 1] = mode ? -1 : 2 * (nmin >> 1) - nmax;
            }
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                const int qi = q0 + cb * 16 + col;
                dq[cb] = (mode && qi < nq) ? (unsigned)((na[cb] >> 1) - (nmin >> 1)) : 0u;
            }
        }
        unsigned m[NCB][2];
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) m[cb][0] = m[cb][1] = 0u;

        auto run_chunks = [&](auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            // One step = the 16 matrix instructions of tile X interleaved BY HAND with the fold of the previous tile P (39
            // vector ops + 1 LDS atomic), two to three vector ops in the shadow of each matrix instruction: a wave that is left
            // to the scheduler issues its fold in blocks, and with two lock-stepped waves per SIMD the matrix pipe then
            // idles while both fold (counters: matrix pipe 51 % busy, 16 % of the cycles co-issued).
            // tile_c: compile-time index of tile P in the chunk, -1 .. 7 (-1 = the previous chunk's last tile, or the dummy tile)
            auto step = [&](const Frag &f, v4i (&X)[NCB], const v4i (&P)[NCB], unsigned slot_addr, auto tile_c) {
                constexpr int TB = (decltype(tile_c)::value + 1) * (16 * ROWB);
                auto x = [&](int cb, int r) -> unsigned { // value of P for the row direction
                    unsigned v = (unsigned)P[cb][r];
                    if (MODE >= 1) v -= dq[cb];
                    if (MODE == 2) v = (phantom & (1u << cb)) ? 0u : v;
                    return v;
                };
                unsigned w[4];
                const int kcl = kc; // (a plain use: clang does not capture a variable that only appears as an asm operand)
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    // (64-byte rows: eight matrix instructions per step, spread over the sixteen slots of the fold)
                    if (KS == 2)
                        X[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f.a[0], qf[k][0], f.cv, 0, 0, 0);
                    else if ((k & 1) == 0)
                        X[k / 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f.a[0], qf[k / 2][0], f.cv, 0, 0, 0);
                    // column direction: two max-only streams per column block
                    m[k][0] = umax3(m[k][0], (unsigned)P[k][0], (unsigned)P[k][2]);
                    m[k][1] = umax3(m[k][1], (unsigned)P[k][1], (unsigned)P[k][3]);
                    // row direction: max over the class's 8 column blocks, register r = k & 3
                    if (ISX_PAIR_X & 8) {
                        if (k < 4) w[k] = 0;
                    } else if (k < 4)
                        w[k] = umax3(x(0, k), x(1, k), x(2, k));
                    else
                        w[k - 4] = umax3(w[k - 4], x(3, k - 4), x(4, k - 4));
                    if (!(ISX_PAIR_X & 1)) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (KS == 2)
                        X[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f.a[1], qf[k][1], X[k], 0, 0, 0);
                    else if ((k & 1) == 0)
                        X[4 + k / 2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f.a[0], qf[4 + k / 2][0], f.cv, 0, 0, 0);
                    if (ISX_PAIR_X & 8) {
                    } else if (k < 2) {
                        w[2 * k] = umax3(w[2 * k], x(5, 2 * k), x(6, 2 * k));
                        w[2 * k + 1] = umax3(w[2 * k + 1], x(5, 2 * k + 1), x(6, 2 * k + 1));
                    } else if (k < 4) {
                        const int r = 2 * (k - 2);
                        const unsigned a7 = x(7, r), b7 = x(7, r + 1);
                        w[r] = w[r] > a7 ? w[r] : a7;
                        w[r + 1] = w[r + 1] > b7 ? w[r + 1] : b7;
                    } else if (k == 4 && !(ISX_PAIR_X & 4)) {
                        // 16 lanes x 4 registers -> 1 register, SIX ops: a reduce-scatter over the four quads of a row.  The
                        // groups of the table are the four lane positions inside a quad; quad q ends up with row register q's
                        // maxima over the four quads.  DPP's bank mask (one bit per quad) writes only the quads that keep the
                        // other register of a pair, so no select is needed: quads {0, 2} keep w0 / w2, quads {1, 3} w1 / w3, each
                        // combined with its neighbour quad (row_ror:12 reads quad q + 1, row_ror:4 quad q - 1) ...
                        asm volatile("v_max_u32_dpp %0, %0, %0 row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
                                     "v_max_u32_dpp %2, %2, %2 row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
                                     "v_max_u32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                                     "v_max_u32_dpp %2, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xa"
                                     : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
                    } else if (k == 5 && !(ISX_PAIR_X & 4)) {
                        // ... then the quads two apart: quads {0, 1} finish w0 / w1, quads {2, 3} take w2 / w3
                        asm volatile("v_max_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                                     "v_max_u32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc"
                                     : "+v"(w[0]), "+v"(w[2]));
                    } else if (k == 6 && (ISX_PAIR_X & 2)) {
                        asm volatile("" ::"v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
                    } else if (k == 6 && NG == 4) {
                        // the class's correction and tag, then the LDS max into the row table: lane -> (row 4g + its quad,
                        // group = its position in the quad), 64 consecutive words per tile
                        asm volatile("v_lshl_add_u32 %0, %0, 5, %2\n\t"
                                     "ds_max_u32 %1, %0 offset:%3"
                                     : "+v"(w[0])
                                     : "v"(slot_addr), "s"(kcl), "n"(TB)
                                     : "memory");
                    } else if (k == 6) {
                        // two groups per row: the even positions take their odd neighbours' maxima and write, 32 words per tile
                        unsigned long long keep;
                        asm volatile("v_max_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                                     "v_lshl_add_u32 %1, %1, 7, %3\n\t"
                                     "s_mov_b64 %0, exec\n\t"
                                     "s_mov_b64 exec, %4\n\t"
                                     "ds_max_u32 %2, %1 offset:%5\n\t"
                                     "s_mov_b64 exec, %0"
                                     : "=&s"(keep), "+v"(w[0])
                                     : "v"(slot_addr), "s"(kcl), "s"(0x5555555555555555ull), "n"(TB)
                                     : "memory");
                    }
                    if (!(ISX_PAIR_X & 1)) __builtin_amdgcn_sched_barrier(0);
                }
            };
            v4i accA[NCB], accB[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) accB[cb] = v4i{0, 0, 0, 0};
            Frag f0{}, f1{};
            if (wave_live) read_frag(smem, std::integral_constant<int, 0>{}, f0);
            int buf = 0;
            for (int c = 0; c < n_chunks; c++) {
                const int buf1 = (buf + 1) & (NBUF - 1);
                if (c + NBUF - 1 < n_chunks) stage_chunk(c + NBUF - 1, (buf + NBUF - 1) & (NBUF - 1));
                const unsigned char *base = smem + buf * BUF;
                const unsigned slot_addr = (ISX_PAIR_X & 16) ? row_slot_lds + ((c * (CHUNK * ROWB) + wave * 8192) & 0xFFFF & ~2047)  // (diag: every wave its own rows)
                                                             : row_slot_lds + c * (CHUNK * ROWB);
                if (wave_live) {
                    auto two_tiles = [&](auto tc) {
                        constexpr int t = decltype(tc)::value;
                        read_frag(base, std::integral_constant<int, t + 1>{}, f1);
                        step(f0, accA, accB, slot_addr, std::integral_constant<int, t - 1>{});
                        if (t + 2 < 8)
                            read_frag(base, std::integral_constant<int, (t + 2) & 7>{}, f0);
                        else
                            read_frag(smem + buf1 * BUF, std::integral_constant<int, 0>{}, f0);
                        step(f1, accB, accA, slot_addr, std::integral_constant<int, t>{});
                    };
                    two_tiles(std::integral_constant<int, 0>{});
                    two_tiles(std::integral_constant<int, 2>{});
                    two_tiles(std::integral_constant<int, 4>{});
                    two_tiles(std::integral_constant<int, 6>{});
                }
                // Chunks up to c + 2 must have landed and every wave must be done with chunk c: the next iteration reads
                // chunk c + 1 and, one step early, the first tile of chunk c + 2.  A __syncthreads() here waits for ALL
                // memory operations (vmcnt(0)), i.e. also for chunk c + 3, issued a moment ago; wait for all but the
                // newest stage instead (chunk c + 2 has been travelling for two iterations).  The LDS reads of chunk c
                // were consumed by this iteration's matrix instructions, the table atomics are only read after the last pass.
                if (ISX_PAIR_T & 64)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (timing diagnostic: no chunk barrier)
                else if (c + NBUF - 1 < n_chunks) {
                    if (KS == 2)
                        asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
                }
                else
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef ISX_PAIR_SKEW
                if (wave >= WAVES / 2) __builtin_amdgcn_s_sleep(ISX_PAIR_SKEW);
#endif
                buf = buf1;
            }
            // drain: fold the last tile (the matrix instructions of this extra step run on stale fragments into accA, unused)
            if (wave_live && n_chunks > 0)
                step(f0, accA, accB, row_slot_lds + (n_chunks - 1) * (CHUNK * ROWB), std::integral_constant<int, 7>{});
        };

        ISX_STAMP(1); // class set-up (waits for the query norms)
        __syncthreads(); // the first three chunks of this pass have landed (staged before the previous pass's epilogue)
        ISX_STAMP(2); // wait for the other waves and the first chunks
        // Overlapping images: once the column direction has overflowed its candidate list the pair goes to the two-pass
        // kernel anyway -- leave now (both directions reported as -1) instead of forming the remaining products.  The
        // barrier above has drained this wave's DMA; lst[0] only changes in the column epilogue, before that barrier.
        // "Overflowed" is taken as "on course to overflow": more than pass / n_pass of the list is used after `pass` passes.  Leaving
        // is always safe (the two-pass kernel is exact), and for a pair that would have stayed just below the limit the
        // recounts it is spared cost about what the two-pass items do.
        if (pass > 0 && lst[0] * n_pass > ISX_PAIR_CAP * pass && !(ISX_PAIR_T & 20)) {
            if (tid == 0) count_out[2 * (int64_t)item_i] = count_out[2 * (int64_t)item_i + 1] = -1;
            return;
        }
#ifdef ISX_PAIR_REP /* timing diagnostic: the chunk loop of every pass ISX_PAIR_REP times (maxima are idempotent: same results) */
        for (int rep = 1; rep < ISX_PAIR_REP; rep++) {
            run_chunks(std::integral_constant<int, 0>{});
            stage_first();
            __syncthreads();
        }
#endif
        if (mode == 0)
            run_chunks(std::integral_constant<int, 0>{});
        else if (mode == 1)
            run_chunks(std::integral_constant<int, 1>{});
        else
            run_chunks(std::integral_constant<int, 2>{});
        ISX_STAMP(3); // the chunk loop
        // every wave is past the last chunk's barrier: the buffers are free.  Start the next pass's first chunks and query
        // rows now, so that they travel while this pass's column results are evaluated.
        if (pass + 1 < n_pass) {
            stage_first();
            if (!(ISX_PAIR_X & 128)) load_queries(pass + 1);
        }
        ISX_STAMP(8); // (issuing the next pass's requests)

        // column direction of this pass: candidates among this wave's 128 queries.  Almost no query is one: every lane
        // first takes the cheap conservative reject (one float product), and only a wave that has a survivor enters the
        // exact test (two IEEE sqrtf each); the four row groups are merged with v_permlane*_swap instead of 32
        // LDS-crossbar shuffles: -1 % of the kernel (profiles/r03_knn2_count_pmc.md)
        if (wave_live && !(ISX_PAIR_X & 64)) {
            unsigned maybe = 0;
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                unsigned b1 = m[cb][0] > m[cb][1] ? m[cb][0] : m[cb][1];
                unsigned b2 = m[cb][0] > m[cb][1] ? m[cb][1] : m[cb][0];
                top2_over_rows(b1, b2);
                m[cb][0] = b1, m[cb][1] = b2;
                const int s0_hi = na[cb] - 2 * ((int)b1 - ISX_B2), s1_hi = na[cb] - 2 * ((int)b2 - ISX_B2);
                // b2 == 0: fewer than two groups saw a train row -- let the exact scan decide
                const bool rejected = b2 != 0u && !(flags & 1) && (float)(s0_hi > 0 ? s0_hi - 1 : 0) > (float)s1_hi * r2_up;
                if (b1 != 0u && na[cb] >= 0 && !rejected) maybe |= 1u << cb;
            }
            if (g != 0) maybe = 0;
            if (__builtin_amdgcn_ballot_w64(maybe != 0u) != 0ull) {
                for (int cb = 0; cb < NCB; cb++) {
                    if (!(maybe & (1u << cb))) continue;
                    const unsigned b1 = m[cb][0], b2 = m[cb][1];
                    bool cand = true;
                    if (b2 != 0u && !(flags & 1)) {
                        const int s0_hi = na[cb] - 2 * ((int)b1 - ISX_B2), s1_hi = na[cb] - 2 * ((int)b2 - ISX_B2);
                        cand = ratio_good(s0_hi > 0 ? s0_hi - 1 : 0, s1_hi, ratio);
                    }
                    if (cand) {
                        const int pos = atomicAdd(&lst[0], 1);
                        if (pos < ISX_PAIR_CAP) lst[4 + pos] = q0 + cb * 16 + col;
                    }
                }
            }
        }
        ISX_STAMP(4); // next pass's requests + column epilogue
    }
    __syncthreads();
    ISX_STAMP(5);
    // row direction: candidates among the train image's rows, from the table
    for (int t = tid; t < nt; t += 64 * WAVES) {
        unsigned E1, E2; // the largest and the second largest of the row's group maxima
        if (NG == 4) {
            const uint4 e = ((const uint4 *)rowtab)[t];
            unsigned a = e.x > e.y ? e.x : e.y, b = e.x > e.y ? e.y : e.x;
            unsigned c = e.z > e.w ? e.z : e.w, d = e.z > e.w ? e.w : e.z;
            E1 = a > c ? a : c;
            const unsigned lo = a > c ? c : a, hi2 = b > d ? b : d;
            E2 = lo > hi2 ? lo : hi2;
        } else {
            const uint2 e = ((const uint2 *)rowtab)[t];
            E1 = e.x > e.y ? e.x : e.y, E2 = e.x > e.y ? e.y : e.x;
        }
        if (E1 == 0u) continue; // no query row at all
        // entry >> TAGB = raw + ISX_B3 - floor(nmin_c / 2), raw = B2 + t.q - ceil(n_t / 2); squared distance s = 2 B2 - p_t - (2 raw - n_q)
        const int c1 = E1 & (NCLS - 1), c2 = E2 & (NCLS - 1);
        const int u1 = (int)(E1 >> TAGB) - ISX_B3, u2 = (int)(E2 >> TAGB) - ISX_B3; // raw - floor(nmin_c / 2): 2 u >= 2 raw - n_q in the class
        const int raw2 = (int)(E2 >> TAGB) - cls[2 * c2];
        bool cand = true;                                      // fewer than two real entries: the exact scan decides
        if (E2 != 0u && raw2 >= (1 << 20) && !(flags & 1)) {   // (raw of a real query row is >= 2^20; masked columns give 0)
            // the two entries belong to two distinct query rows whose 2 raw - n_q is at least 2 u + (2 floor(nmin/2) - nmax)
            const int lo1 = 2 * u1 + cls[2 * c1 + 1], lo2 = 2 * u2 + cls[2 * c2 + 1];
            const int lo = lo1 < lo2 ? lo1 : lo2;
            const int s0_lo = 2 * ISX_B2 - 1 - 2 * u1, s1_hi = 2 * ISX_B2 - lo;
            cand = ratio_good(s0_lo > 0 ? s0_lo : 0, s1_hi, ratio);
        }
        if (cand) {
            const int pos = atomicAdd(&lst[1], 1);
            if (pos < ISX_PAIR_CAP) lst[4 + ISX_PAIR_CAP + pos] = t;
        }
    }
    __syncthreads();
    ISX_STAMP(6); // table scan
    int n_col = lst[0], n_row = lst[1];
    if (ISX_PAIR_T & 32) { // timing diagnostic: exactly 16 recounts per direction, whatever the filter found
        n_col = n_row = 16;
        if (tid < 16) lst[4 + tid] = lst[4 + ISX_PAIR_CAP + tid] = tid;
        __syncthreads();
    }
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    int good_col = 0, good_row = 0;
    // exact recounts, up to 16 candidates per wave and walk of the other image; the lists are spread evenly over the waves
    if (n_col <= ISX_PAIR_CAP && !(ISX_PAIR_T & 4)) {
        const int gsz = n_col > 16 * WAVES ? 16 : (n_col + WAVES - 1) / WAVES;
        for (int k0 = wv * gsz; k0 < n_col; k0 += WAVES * gsz)
            good_col += knn2_exact_recount16<KS>(desc, nrm, lst + 4 + k0, n_col - k0 < gsz ? n_col - k0 : gsz, it.q_start, it.t_start,
                                             nt, lane, ratio);
    }
    if (n_row <= ISX_PAIR_CAP && !(ISX_PAIR_T & 4)) {
        const int gsz = n_row > 16 * WAVES ? 16 : (n_row + WAVES - 1) / WAVES;
        for (int k0 = wv * gsz; k0 < n_row; k0 += WAVES * gsz)
            good_row += knn2_exact_recount16<KS>(desc, nrm, lst + 4 + ISX_PAIR_CAP + k0, n_row - k0 < gsz ? n_row - k0 : gsz, it.t_start,
                                             it.q_start, nq, lane, ratio);
    }
    if (lane == 0) {
        if (good_col) atomicAdd(&lst[2], good_col);
        if (good_row) atomicAdd(&lst[3], good_row);
    }
    __syncthreads();
    ISX_STAMP(7); // recounts
    if (tid == 0) {
        // ([2e] = the direction whose queries are the ITEM's q image: the column direction unless the roles were swapped)
        count_out[2 * (int64_t)item_i + (swapped ? 1 : 0)] = (ISX_PAIR_T & 16) ? n_col : n_col <= ISX_PAIR_CAP ? lst[2] : -1;
        count_out[2 * (int64_t)item_i + (swapped ? 0 : 1)] = (ISX_PAIR_T & 16) ? n_row : n_row <= ISX_PAIR_CAP ? lst[3] : -1;
    }
}

extern "C" int isx_knn2_l2_count_pairs(const int8_t *desc_dev, const int32_t *nrm_dev, const int32_t *cin_dev, int d_pad,
                                       const isx_knn_item *items_dev, int n_items, int max_rows, double ratio,
                                       int32_t *count_out_dev, int flags, void *stream) {
    ISX_REQUIRE(d_pad == 128 || d_pad == 64, "isx_knn2_l2_count_pairs needs 128- or 64-byte rows (got d_pad %d)", d_pad);
    ISX_REQUIRE(n_items >= 0 && max_rows >= 0, "negative size");
    if (n_items == 0 || max_rows == 0) return ISX_OK;
    ISX_REQUIRE(desc_dev && nrm_dev && cin_dev && items_dev && count_out_dev, "null device pointer");
    ISX_REQUIRE(((uintptr_t)desc_dev & 15) == 0, "desc_dev must be 16-byte aligned");
    constexpr int WAVES = 8;
    // images of up to 4096 rows: the four-group table.  Beyond: two groups per row (a pair whose smaller image has more than
    // 8192 rows or whose larger one more than 16384 comes back as (-1, -1), like any other direction the caller re-runs)
    const bool big = max_rows > ISX_PAIR_MAX_ROWS;
    typedef void (*pair_fn)(const int8_t *, const int32_t *, const int32_t *, const isx_knn_item *, int, double, int32_t *, int);
    const pair_fn fn = d_pad == 128 ? (big ? knn2_pair_filter_kernel<WAVES, 2, 2> : knn2_pair_filter_kernel<WAVES, 4, 2>)
                                    : (big ? knn2_pair_filter_kernel<WAVES, 2, 1> : knn2_pair_filter_kernel<WAVES, 4, 1>);
    ISX_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, ISX_PAIR_LDS));
    const int per = isx_items_per_launch(1, 64 * WAVES);
    hipStream_t st = (hipStream_t)stream;
    for (int off = 0; off < n_items; off += per) {
        const int n = n_items - off < per ? n_items - off : per;
        hipLaunchKernelGGL(fn, dim3((unsigned)n), dim3(64 * WAVES), ISX_PAIR_LDS, st, desc_dev, nrm_dev, cin_dev, items_dev + off, n,
                           ratio, count_out_dev + 2 * (int64_t)off, flags);
    }
    return isx_launch_status("knn2_pair_filter_kernel");
}

// ------------------------------------------------------------------------------------------------
// Hamming on the matrix pipe (opt-in norm="hamming"; binary descriptors of up to 256 bits).
// isx_hamming_fp4_prep: every bit of a row becomes one FP4 nibble (0x0 = 0.0, 0x2 = 1.0), 128 bytes per row
// (rows shorter than 32 bytes are zero padded: equal padding on both sides adds nothing to a distance);
// nrm = popcount, cin = float bits of 512 - popcount/2.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) hamming_fp4_prep_kernel(const uint8_t *__restrict__ src, int64_t rows, int d,
                                                               uint8_t *__restrict__ desc, int32_t *__restrict__ nrm,
                                                               int32_t *__restrict__ cin) {
    // 8 threads per row, 4 source bytes (= 16 output bytes) each
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t >> 3;
    const int unit = (int)(t & 7);
    int pc = 0;
    if (row < rows) {
        v4i out = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int k = unit * 4 + e;
            const unsigned byte = k < d ? src[row * d + k] : 0u;
            pc += __popc(byte);
            unsigned w = 0; // bit b of the byte -> nibble b of the word
#pragma unroll
            for (int b = 0; b < 8; b++) w |= ((byte >> b) & 1u) << (4 * b + 1);
            out[e] = (int)w;
        }
        *(v4i *)(desc + row * 128 + unit * 16) = out;
    }
    pc += __shfl_xor(pc, 1);
    pc += __shfl_xor(pc, 2);
    pc += __shfl_xor(pc, 4);
    if (row < rows && unit == 0) {
        nrm[row] = pc;
        cin[row] = (int32_t)__float_as_uint(512.f - 0.5f * (float)pc);
    }
}

extern "C" int isx_hamming_fp4_prep(const uint8_t *src_dev, int64_t rows, int d, uint8_t *desc_dev, int32_t *nrm_dev,
                                    int32_t *cin_dev, void *stream) {
    ISX_REQUIRE(d > 0 && d <= 32, "the FP4 Hamming path takes descriptors of up to 32 bytes (got %d)", d);
    ISX_REQUIRE(rows >= 0 && rows < ((int64_t)1 << 31) / 8, "rows=%lld out of range", (long long)rows);
    if (rows == 0) return ISX_OK;
    ISX_REQUIRE(src_dev && desc_dev && nrm_dev && cin_dev, "null device pointer");
    ISX_REQUIRE(((uintptr_t)desc_dev & 15) == 0, "desc_dev must be 16-byte aligned");
    const unsigned grid = (unsigned)((rows * 8 + 255) / 256);
    hipLaunchKernelGGL(hamming_fp4_prep_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src_dev, rows, d, desc_dev,
                       nrm_dev, cin_dev);
    return isx_launch_status("hamming_fp4_prep_kernel");
}

extern "C" int isx_knn2_hamming_count_fp4(const uint8_t *desc_dev, const int32_t *nrm_dev, const int32_t *cin_dev,
                                          const isx_knn_item *items_dev, int n_items, int max_nq, double ratio,
                                          int32_t *count_out_dev, void *stream) {
    ISX_REQUIRE(n_items >= 0 && max_nq >= 0, "negative size");
    if (n_items == 0 || max_nq == 0) return ISX_OK;
    ISX_REQUIRE(desc_dev && nrm_dev && cin_dev && items_dev && count_out_dev, "null device pointer");
    ISX_REQUIRE(((uintptr_t)desc_dev & 15) == 0, "desc_dev must be 16-byte aligned");
    const int qpb = 32 * ISX_COUNT_WAVES * ISX_COUNT_NCB;
    const int nqb = (max_nq + qpb - 1) / qpb;
    const int per = isx_items_per_launch(nqb, 64 * ISX_COUNT_WAVES);
    ISX_REQUIRE(per > 0, "max_nq too large (%d)", max_nq);
    for (int off = 0; off < n_items; off += per) {
        const int n = n_items - off < per ? n_items - off : per;
#ifdef ISX_HAMMING_SHAPE16 /* experiment macro (tools/build_variant.sh): the 16x16x128 FP4 chain; measured 5.5 % SLOWER than
                            the 32x32x64 one on 256-bit rows (profiles/r02_shape_ab.md), so it is not the default */
        hipLaunchKernelGGL((knn2_count16_kernel<2, 1>), dim3((unsigned)((n + 7) / 8 * 8 * nqb)), dim3(64 * ISX_COUNT_WAVES), 0,
                           (hipStream_t)stream, (const int8_t *)desc_dev, nrm_dev, cin_dev, items_dev + off, n, nqb, ratio,
                           count_out_dev + off, 0);
#else
        hipLaunchKernelGGL((knn2_count_kernel<4, 1>), dim3((unsigned)((n + 7) / 8 * 8 * nqb)), dim3(64 * ISX_COUNT_WAVES), 0,
                           (hipStream_t)stream, (const int8_t *)desc_dev, nrm_dev, cin_dev, items_dev + off, n, nqb, ratio,
                           count_out_dev + off, 0);
#endif
    }
    return isx_launch_status("knn2_count_kernel<4,1>");
}

// ------------------------------------------------------------------------------------------------
// Stable compaction of the rows that pass the Lowe ratio test: Stitcher.filter_matches
// (reference stitcher.py:118-127) as a list, in query order.  One workgroup walks the rows in chunks
// of 1024 with a ballot/popcount prefix sum, so the output order equals the reference's append order.
// ------------------------------------------------------------------------------------------------
// n_seg > 1: idx / dist hold n_seg kNN-2 results of the SAME queries against consecutive segments of seg_rows train
// rows each (result s at entry offset s * nq * 2, indices relative to its segment); they are merged per query first.
// The global two nearest rows are the two smallest of the union of the per-segment pairs under (distance, train index)
// -- distances are float32 square roots of integers, equal integers give equal floats, and cv2 keeps the lower train
// index on a tie, within a segment (the kernel's rule) and therefore across segments too.
#define RC_MAX_SEG 16
__global__ void __launch_bounds__(1024) ratio_compact_kernel(const int32_t *__restrict__ idx,
                                                             const float *__restrict__ dist, int nq, double ratio,
                                                             int n_seg, int seg_rows,
                                                             int32_t *__restrict__ good_q,
                                                             int32_t *__restrict__ good_t,
                                                             int32_t *__restrict__ count_out) {
    __shared__ int wave_tot[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int start = 0; start < nq; start += 1024) {
        const int i = start + tid;
        bool good = false;
        int t0 = -1;
        if (i < nq) {
            // (all segments' pairs are fetched before the first compare: one memory round trip instead of n_seg)
            int2 js[RC_MAX_SEG];
            float2 ds[RC_MAX_SEG];
#pragma unroll
            for (int sg = 0; sg < RC_MAX_SEG; sg++) {
                const int64_t o = (int64_t)(sg < n_seg ? sg : 0) * nq + i;
                js[sg] = ((const int2 *)idx)[o];
                ds[sg] = ((const float2 *)dist)[o];
            }
            t0 = js[0].x;
            int t1 = js[0].y;
            float d0 = ds[0].x, d1 = ds[0].y;
#pragma unroll
            for (int sg = 1; sg < RC_MAX_SEG; sg++) { // (segments in index order: a later candidate only wins if strictly nearer)
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int j = e ? js[sg].y : js[sg].x;
                    if (sg >= n_seg || j < 0) continue; // (no break: the arrays stay in registers, statically indexed)
                    const int jg = sg * seg_rows + j;
                    const float dj = e ? ds[sg].y : ds[sg].x;
                    if (t0 < 0 || dj < d0) {
                        t1 = t0, d1 = d0, t0 = jg, d0 = dj;
                    } else if (t1 < 0 || dj < d1) {
                        t1 = jg, d1 = dj;
                    }
                }
            }
            if (t0 >= 0) good = t1 < 0 || (double)d0 < ratio * (double)d1;
        }
        const unsigned long long b = __ballot(good);
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(b);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; w++) off += wave_tot[w];
        if (good) {
            good_q[off + before] = i;
            good_t[off + before] = t0;
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; w++) tot += wave_tot[w];
            base_s += tot;
        }
        __syncthreads();
    }
    if (tid == 0) count_out[0] = base_s;
}

extern "C" int isx_ratio_compact(const int32_t *idx_dev, const float *dist_dev, int nq, double ratio,
                                 int32_t *good_q_out_dev, int32_t *good_t_out_dev, int32_t *count_out_dev,
                                 void *stream) {
    ISX_REQUIRE(nq >= 0, "negative nq");
    ISX_REQUIRE(count_out_dev, "null count_out_dev");
    ISX_REQUIRE(nq == 0 || (idx_dev && dist_dev && good_q_out_dev && good_t_out_dev), "null device pointer");
    hipLaunchKernelGGL(ratio_compact_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, idx_dev, dist_dev, nq, ratio, 1, 0,
                       good_q_out_dev, good_t_out_dev, count_out_dev);
    return isx_launch_status("ratio_compact_kernel");
}

extern "C" int isx_ratio_compact_segments(const int32_t *idx_dev, const float *dist_dev, int nq, int n_seg, int seg_rows,
                                          double ratio, int32_t *good_q_out_dev, int32_t *good_t_out_dev,
                                          int32_t *count_out_dev, void *stream) {
    ISX_REQUIRE(nq >= 0 && n_seg >= 1 && n_seg <= RC_MAX_SEG && seg_rows >= 0, "bad sizes (1 <= n_seg <= %d)", RC_MAX_SEG);
    ISX_REQUIRE((((uintptr_t)idx_dev | (uintptr_t)dist_dev) & 7) == 0, "idx_dev / dist_dev must be 8-byte aligned");
    ISX_REQUIRE(count_out_dev, "null count_out_dev");
    ISX_REQUIRE(nq == 0 || (idx_dev && dist_dev && good_q_out_dev && good_t_out_dev), "null device pointer");
    hipLaunchKernelGGL(ratio_compact_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, idx_dev, dist_dev, nq, ratio, n_seg,
                       seg_rows, good_q_out_dev, good_t_out_dev, count_out_dev);
    return isx_launch_status("ratio_compact_kernel");
}

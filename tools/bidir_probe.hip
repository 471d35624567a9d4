// Micro-probe (gfx950): would "both directions from one tile" beat two passes in the kNN count kernel?
// One UNIT = 32 train rows x the wave's 64 queries on v_mfma_i32_16x16x64_i8 (16 instructions), as knn2_count16_kernel.
//   mode 0  two-pass form   : 16 MFMA + 40 three-input fold ops (column direction, in-lane)        -> x2 per unordered unit
//   mode 1  both directions : 16 MFMA + the same 40 fold ops
//                             + 32 v_sub (make the value symmetric: subtract the query's half norm)
//                             + 40 ops element-wise top-2 over the 4 column blocks (8 row positions x 5)
//                             + 8 positions x 4 DPP steps x 4 ops (top-2 across the 16 lanes that hold one train row)
//                             + 8 ds_write for the cross-wave merge
//   mode 2  mode 1 without the DPP steps (how much of the cost is the cross-lane part)
//   mode 3  round 3: both directions with MAX-ONLY FILTER folds (0.5 op per value each way, exact top-2 left to a rescan of
//           the few "maybe good" queries): 16 MFMA + 16 max3 (column direction, two streams per column block)
//           + 32 v_sub + 16 ops element-wise max over the column blocks + 8 DPP max (16 lanes of a row -> 4 groups)
//           + 4 ds_max_u32 into a per-row table in LDS (amounts of the 8-column-block form: DPP and atomics per 8 values)
//   mode 4  mode 3 without v_sub (norm carried some other way), mode 5 = mode 3 without DPP + atomics
// Same operand bytes (SIFT-like), every CU busy, 3 waves per SIMD as the kernel: the number printed is ns per unit
// per SIMD, directly comparable with the kernel's 185 ns per directed unit.
// Build: hipcc --offload-arch=gfx950 -O3 tools/bidir_probe.hip -o tools/bin/bidir_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int rnd(int x) {
    unsigned h = (unsigned)x * 2654435761u + 4u * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return (int)((h & 0x3f3f3f3fu) | 0x80808080u);
}

template <int MODE>
__global__ void __launch_bounds__(256, 3) probe(int iters, unsigned *out) {
    __shared__ unsigned lds[4][64];
    __shared__ unsigned rowtab[128][4];
    if (MODE >= 3) { for (int i = threadIdx.x; i < 512; i += 256) (&rowtab[0][0])[i] = 0; __syncthreads(); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4i a[2][2], q[4][2], cv[2];
    for (int t = 0; t < 2; t++) for (int k = 0; k < 2; k++) for (int e = 0; e < 4; e++) a[t][k][e] = rnd(lane * 131 + t * 17 + k * 5 + e);
    for (int c = 0; c < 4; c++) for (int k = 0; k < 2; k++) for (int e = 0; e < 4; e++) q[c][k][e] = rnd(lane * 977 + c * 29 + k * 3 + e + 1000);
    for (int t = 0; t < 2; t++) for (int e = 0; e < 4; e++) cv[t][e] = 4194304 - (lane * 7 + e);
    unsigned m[4][4], G[4], qn[4];
    for (int c = 0; c < 4; c++) { G[c] = 0; qn[c] = lane + c; for (int r = 0; r < 4; r++) m[c][r] = 0; }
    v4i accA[2][4], accB[2][4];
    for (int t = 0; t < 2; t++) for (int c = 0; c < 4; c++) accB[t][c] = v4i{0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            v4i (&X)[2][4] = half == 0 ? accA : accB;
            v4i (&Y)[2][4] = half == 0 ? accB : accA;
#pragma unroll
            for (int t = 0; t < 2; t++) {
#pragma unroll
                for (int c = 0; c < 4; c++) X[t][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t][0], q[c][0], cv[t], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 4; c++) X[t][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t][1], q[c][1], X[t][c], 0, 0, 0);
            }
            if (MODE >= 3) {
                // column direction, filter form: two max-only streams per column block (16 ops for 32 values)
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(m[c][0]) : "v"(Y[t][c][0]), "v"(Y[t][c][1]));
                        asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(m[c][1]) : "v"(Y[t][c][2]), "v"(Y[t][c][3]));
                    }
                if (MODE != 4)
#pragma unroll
                    for (int t = 0; t < 2; t++)
#pragma unroll
                        for (int c = 0; c < 4; c++)
#pragma unroll
                            for (int r = 0; r < 4; r++) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(Y[t][c][r]) : "v"(qn[c]));
                // row direction: element-wise max over the column blocks (2 ops per 4 values), running over both halves
                // of the iteration as the 8-column-block kernel would (rr persists; reduced + flushed every second half)
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        unsigned mx;
                        asm volatile("v_max3_u32 %0, %1, %2, %3" : "=v"(mx) : "v"(Y[t][0][r]), "v"(Y[t][1][r]), "v"(Y[t][2][r]));
                        if (half == 0) asm volatile("v_max_u32 %0, %1, %2" : "=v"(G[r]) : "v"(mx), "v"(Y[t][3][r]));   // (G, qn reused as scratch state)
                        else asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(G[r]) : "v"(mx), "v"(Y[t][3][r]));
                        if (MODE != 5 && half == 1 && t == 1) {
                            asm volatile("v_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(G[r]));
                            asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(G[r]));
                        }
                    }
                if (MODE != 5 && half == 1) {
                    if ((lane & 3) == 0)
#pragma unroll
                        for (int r = 0; r < 4; r++) atomicMax(&rowtab[((it * 4 + r) & 7) * 16 + (lane >> 4) * 4 + r][(lane >> 2) & 3], G[r]);
                }
            } else
            // column direction: the kernel's fold (40 ops)
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    unsigned t0, t1;
                    asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(t0) : "v"(m[c][r]), "v"(Y[0][c][r]), "v"(Y[1][c][r]));
                    asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(t1) : "v"(m[c][r + 1]), "v"(Y[0][c][r + 1]), "v"(Y[1][c][r + 1]));
                    asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(m[c][r]) : "v"(Y[0][c][r]), "v"(Y[1][c][r]));
                    asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(m[c][r + 1]) : "v"(Y[0][c][r + 1]), "v"(Y[1][c][r + 1]));
                    asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(G[c]) : "v"(t0), "v"(t1));
                }
            if (MODE == 1 || MODE == 2) {
                // row direction.  (a) symmetric value: subtract the query's half norm (32 ops)
                v4i (&S)[2][4] = Y; // in place: the fold above was the last reader of the raw values
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int c = 0; c < 4; c++)
#pragma unroll
                        for (int r = 0; r < 4; r++) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(S[t][c][r]) : "v"(qn[c]));
                // (b) top-2 over the 4 column blocks for each of the 8 row positions (5 ops each)
                unsigned b1[8], b2[8];
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int p = t * 4 + r;
                        unsigned mx, md, mn;
                        asm volatile("v_max3_u32 %0, %1, %2, %3" : "=v"(mx) : "v"(S[t][0][r]), "v"(S[t][1][r]), "v"(S[t][2][r]));
                        asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(md) : "v"(S[t][0][r]), "v"(S[t][1][r]), "v"(S[t][2][r]));
                        asm volatile("v_min_u32 %0, %1, %2" : "=v"(mn) : "v"(mx), "v"(S[t][3][r]));
                        asm volatile("v_max_u32 %0, %1, %2" : "=v"(b1[p]) : "v"(mx), "v"(S[t][3][r]));
                        asm volatile("v_max_u32 %0, %1, %2" : "=v"(b2[p]) : "v"(md), "v"(mn));
                    }
                if (MODE == 1) {
                    // (c) top-2 across the 16 lanes of a row: 4 DPP steps x 4 ops per position
#pragma unroll
                    for (int p = 0; p < 8; p++) {
#define DPP_STEP(CTRL)                                                                                                   \
    {                                                                                                                    \
        unsigned o2, mn;                                                                                                 \
        asm volatile("v_mov_b32_dpp %0, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(o2) : "v"(b2[p]));                \
        asm volatile("v_min_u32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(mn) : "v"(b1[p]));            \
        asm volatile("v_max_u32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(b1[p]));                      \
        asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(b2[p]) : "v"(o2), "v"(mn));                                      \
    }
                        DPP_STEP("quad_perm:[1,0,3,2]")
                        DPP_STEP("quad_perm:[2,3,0,1]")
                        DPP_STEP("row_half_mirror")
                        DPP_STEP("row_mirror")
#undef DPP_STEP
                    }
                }
                // (d) cross-wave merge through LDS: one write per position from the row's first lane
                if ((lane & 15) == 0)
#pragma unroll
                    for (int p = 0; p < 8; p += 2) // (b1, b2) of two positions per 16-byte write
                        *(uint4 *)&lds[wave][(lane >> 4) * 16 + 2 * p] = make_uint4(b1[p], b2[p], b1[p + 1], b2[p + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned acc = 0;
    for (int c = 0; c < 4; c++) { acc ^= G[c]; for (int r = 0; r < 4; r++) acc ^= m[c][r]; }
    for (int t = 0; t < 2; t++) for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) acc ^= (unsigned)accA[t][c][r] ^ (unsigned)accB[t][c][r];
    acc ^= lds[wave][lane];
    if (MODE >= 3) { __syncthreads(); acc ^= rowtab[threadIdx.x & 127][lane & 3] ^ qn[0]; }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static int g_wps = 3; // workgroups per CU = waves per SIMD (argv[1]; 3 as knn2_count16_kernel, 2 as knn2_pair_filter_kernel)
template <int MODE> static double run(int iters, unsigned *out, int blocks) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, iters / 10, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, iters, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // 3 workgroups of 4 waves per CU: 3 waves per SIMD run concurrently; units per SIMD = 3 waves x iters x 2
    return ms * 1e6 / ((double)iters * 2 * g_wps);
}

int main(int argc, char **argv) {
    if (argc > 1) g_wps = atoi(argv[1]);
    if (g_wps < 1 || g_wps > 3) { printf("waves per SIMD: 1..3\n"); return 1; }
    const int blocks = 256 * g_wps; // every CU holds g_wps workgroups (one wave of each per SIMD)
    printf("%d waves per SIMD\n", g_wps);
    unsigned *out; CK(hipMalloc(&out, blocks * 256 * 4));
    const int iters = 20000;
    const double a = run<0>(iters, out, blocks), b = run<1>(iters, out, blocks), c = run<2>(iters, out, blocks);
    const double d3 = run<3>(iters, out, blocks), d4 = run<4>(iters, out, blocks), d5 = run<5>(iters, out, blocks);
    printf("ns per unit per SIMD  (kernel today: ~185 ns per DIRECTED unit)\n");
    printf("  mode 0  two-pass ingredients (16 MFMA + 40 fold)             %7.1f ns   -> %7.1f ns per unordered unit (x2)\n", a, 2 * a);
    printf("  mode 1  both directions (+32 sub +40 top2-of-4 +128 DPP +LDS) %7.1f ns per unordered unit  = %.2fx of two-pass\n", b, b / (2 * a));
    printf("  mode 2  mode 1 without the DPP lane reduction                %7.1f ns per unordered unit  = %.2fx of two-pass\n", c, c / (2 * a));
    printf("  mode 3  both directions, max-only filter folds (+sub +DPP +LDS atomics) %7.1f ns per unordered unit  = %.2fx of two-pass\n", d3, d3 / (2 * a));
    printf("  mode 4  mode 3 without the norm subtraction                  %7.1f ns per unordered unit  = %.2fx of two-pass\n", d4, d4 / (2 * a));
    printf("  mode 5  mode 3 without DPP + LDS atomics                     %7.1f ns per unordered unit  = %.2fx of two-pass\n", d5, d5 / (2 * a));
    return 0;
}

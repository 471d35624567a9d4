// Micro-probe (gfx950): what rate does the i8 32x32x32 MFMA pipe sustain when a wave also folds the previous
// accumulators with v_max3_u32, as the kNN count kernel does?  No LDS, no global loads in the loop.
// Step = 2 chains x 4 dependent MFMAs into one accumulator pair + NV max3 ops on the other pair.
//   IL = 1: VALU ops spread between the MFMAs (pinned by sched_barrier), IL = 0: all MFMAs, then all VALU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                               \
    do {                                                                    \
        hipError_t e = (x);                                                 \
        if (e != hipSuccess) {                                              \
            printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
            exit(1);                                                        \
        }                                                                   \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// operand bytes (the matrix pipe's power, and with it the clock, depends on the operand bits that toggle):
// seed 2 -> small, mostly-zero words; seed 3 -> pseudo-random bits; seed 4 -> "SIFT-like" bytes as the count
// kernel sees them: v - 128 with v uniform in [0, 64)
__device__ __forceinline__ int rnd(int seed, int x) {
    if (seed == 2) return x & 7;
    unsigned h = (unsigned)x * 2654435761u + (unsigned)seed * 40503u;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    if (seed == 4) h = (h & 0x3f3f3f3fu) | 0x80808080u;
    return (int)h;
}

template <int NV, int IL>
__device__ __forceinline__ void step(v4i (&a)[4], v4i (&b)[2][4], const v16i &cin, v16i (&X)[2], const v16i (&Y)[2],
                                     unsigned (&m)[2][8]) {
    constexpr int PER = NV / 8; // VALU ops after each MFMA when interleaved
    int issued = 0;
#pragma unroll
    for (int cb = 0; cb < 2; cb++) {
        v16i ac = cin;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            ac = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks], b[cb][ks], ac, 0, 0, 0);
            if (IL) {
#pragma unroll
                for (int u = 0; u < PER; u++, issued++) {
                    const int c2 = (issued / 8) & 1, r = issued & 7;
                    asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(m[c2][r]) : "v"(Y[c2][r]), "v"(Y[c2][r + 8]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        X[cb] = ac;
    }
    if (!IL) {
#pragma unroll
        for (; issued < NV; issued++) {
            const int c2 = (issued / 8) & 1, r = issued & 7;
            asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(m[c2][r]) : "v"(Y[c2][r]), "v"(Y[c2][r + 8]));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int NV, int IL>
__global__ void __launch_bounds__(256, 2) mfma_kernel(unsigned *out, int iters, int seed) {
    extern __shared__ char dyn_c[];
    const char *dyn = dyn_c;
    const int lane = threadIdx.x;
    v4i a[4], b[2][4];
    for (int i = 0; i < 4; i++) {
        a[i] = v4i{rnd(seed, lane * 64 + i), rnd(seed, lane * 64 + i + 4), rnd(seed, lane * 64 + i + 8), rnd(seed, lane * 64 + i + 12)};
        b[0][i] = v4i{rnd(seed, lane * 64 + i + 16), rnd(seed, lane * 64 + i + 20), rnd(seed, lane * 64 + i + 24), rnd(seed, lane * 64 + i + 28)};
        b[1][i] = v4i{rnd(seed, lane * 64 + i + 32), rnd(seed, lane * 64 + i + 36), rnd(seed, lane * 64 + i + 40), rnd(seed, lane * 64 + i + 44)};
    }
    v16i cin, A[2], B[2];
    for (int i = 0; i < 16; i++) cin[i] = seed + i;
    A[0] = A[1] = B[0] = B[1] = cin;
    unsigned m[2][8];
    for (int i = 0; i < 8; i++) m[0][i] = m[1][i] = 0u;
    for (int it = 0; it < iters; it++) {
        for (int i = 0; i < 4; i++) asm volatile("" : "+v"(a[i])); // loop-variant as far as the optimiser knows
        step<NV, IL>(a, b, cin, A, B, m);
        for (int i = 0; i < 4; i++) asm volatile("" : "+v"(a[i]));
        step<NV, IL>(a, b, cin, B, A, m);
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += m[0][i] + m[1][i];
    for (int i = 0; i < 16; i++) s += A[0][i] + A[1][i] + B[0][i] + B[1][i];
    if (dyn[0] == 77) s++;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Same step, but the A operand (4 x 16 B) and the 16 C-in values come from LDS, read one step ahead
// (8 ds_read_b128 per step), as in the count kernel.
struct Frag {
    v4i a[4];
    v4i cv[4];
};
__device__ __forceinline__ void read_frag(const unsigned char *base, int tl, int lane, Frag &f) {
    const int row = tl * 32 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        const int slot = 2 * ks + h;
        f.a[ks] = *(const v4i *)(base + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int g = 0; g < 4; g++) f.cv[g] = *(const v4i *)(base + 128 * 128 + (tl * 32 + 8 * g + 4 * h) * 4);
}
template <int NV>
__device__ __forceinline__ void step_lds(const Frag &f, v4i (&b)[2][4], v16i (&X)[2], const v16i (&Y)[2], unsigned (&m)[2][8]) {
    constexpr int PER = NV / 8;
    int issued = 0;
#pragma unroll
    for (int cb = 0; cb < 2; cb++) {
        v16i ac = {f.cv[0][0], f.cv[0][1], f.cv[0][2], f.cv[0][3], f.cv[1][0], f.cv[1][1], f.cv[1][2], f.cv[1][3],
                   f.cv[2][0], f.cv[2][1], f.cv[2][2], f.cv[2][3], f.cv[3][0], f.cv[3][1], f.cv[3][2], f.cv[3][3]};
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            ac = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.a[ks], b[cb][ks], ac, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < PER; u++, issued++) {
                const int c2 = (issued / 8) & 1, r = issued & 7;
                asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(m[c2][r]) : "v"(Y[c2][r]), "v"(Y[c2][r + 8]));
            }
        }
        X[cb] = ac;
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int NV, int IL>
__global__ void __launch_bounds__(256, 2) mfma_lds_kernel(unsigned *out, int iters, int seed) {
    extern __shared__ char dyn_c[];
    unsigned char *dyn = (unsigned char *)dyn_c;
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < (128 * 128 + 512) / 4; i += 256) ((int *)dyn)[i] = rnd(seed, i);
    __syncthreads();
    v4i b[2][4];
    for (int i = 0; i < 4; i++) {
        b[0][i] = v4i{rnd(seed, lane * 64 + i + 16), rnd(seed, lane * 64 + i + 20), rnd(seed, lane * 64 + i + 24), rnd(seed, lane * 64 + i + 28)};
        b[1][i] = v4i{rnd(seed, lane * 64 + i + 32), rnd(seed, lane * 64 + i + 36), rnd(seed, lane * 64 + i + 40), rnd(seed, lane * 64 + i + 44)};
    }
    v16i A[2], B[2];
    for (int i = 0; i < 16; i++) A[0][i] = seed + i;
    A[1] = B[0] = B[1] = A[0];
    unsigned m[2][8];
    for (int i = 0; i < 8; i++) m[0][i] = m[1][i] = 0u;
    Frag f0, f1;
    read_frag(dyn, 0, lane, f0);
    for (int it = 0; it < iters; it += 2) {
        read_frag(dyn, 1, lane, f1);
        step_lds<NV>(f0, b, A, B, m);
        read_frag(dyn, 2, lane, f0);
        step_lds<NV>(f1, b, B, A, m);
        read_frag(dyn, 3, lane, f1);
        step_lds<NV>(f0, b, A, B, m);
        read_frag(dyn, 0, lane, f0);
        step_lds<NV>(f1, b, B, A, m);
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += m[0][i] + m[1][i];
    for (int i = 0; i < 16; i++) s += A[0][i] + A[1][i] + B[0][i] + B[1][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// The same MACs per iteration through v_mfma_i32_16x16x64_i8 (32 instructions of 16K MACs instead of 16 of 32K):
// is the energy per MAC -- and with it the clock -- different for the other int8 shape?  4 column blocks of 16
// queries, 2 k-steps of 64, NV max3 per 8 instructions-equivalents as above.
template <int NV>
__global__ void __launch_bounds__(256, 2) mfma16_kernel(unsigned *out, int iters, int seed) {
    extern __shared__ char dyn_c[];
    const char *dyn = dyn_c;
    const int lane = threadIdx.x;
    v4i a[2], b[4][2];
    for (int i = 0; i < 2; i++) {
        a[i] = v4i{rnd(seed, lane * 64 + i), rnd(seed, lane * 64 + i + 4), rnd(seed, lane * 64 + i + 8), rnd(seed, lane * 64 + i + 12)};
        for (int cb = 0; cb < 4; cb++)
            b[cb][i] = v4i{rnd(seed, lane * 64 + i + 16 + 7 * cb), rnd(seed, lane * 64 + i + 20 + 7 * cb), rnd(seed, lane * 64 + i + 24 + 7 * cb),
                           rnd(seed, lane * 64 + i + 28 + 7 * cb)};
    }
    v4i cin{seed, seed + 1, seed + 2, seed + 3};
    v4i X[2][4];
    for (int t = 0; t < 2; t++)
        for (int cb = 0; cb < 4; cb++) X[t][cb] = cin;
    unsigned m[2][8];
    for (int i = 0; i < 8; i++) m[0][i] = m[1][i] = 0u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 4; t++) { // 4 train tiles of 16 rows x 4 column blocks x 2 k-steps = 32 instructions
            for (int i = 0; i < 2; i++) asm volatile("" : "+v"(a[i]));
            const v4i(&Y)[4] = X[(t + 1) & 1];
#pragma unroll
            for (int cb = 0; cb < 4; cb++) {
                v4i ac = cin;
#pragma unroll
                for (int ks = 0; ks < 2; ks++) ac = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks], b[cb][ks], ac, 0, 0, 0);
                X[t & 1][cb] = ac;
#pragma unroll
                for (int u = 0; u < NV / 8; u++) // NV per 8 x 32K MACs = NV/8 per 2 of these instructions
                    asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(m[cb & 1][u & 7]) : "v"(Y[cb][u & 3]), "v"(Y[(cb + 1) & 3][(u + 1) & 3]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += m[0][i] + m[1][i];
    for (int t = 0; t < 2; t++)
        for (int cb = 0; cb < 4; cb++)
            for (int i = 0; i < 4; i++) s += X[t][cb][i];
    if (dyn[0] == 77) s++;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef void (*kern_t)(unsigned *, int, int);

static void run(const char *name, kern_t k, unsigned *out, int wps, int seed) {
    const int iters = 20000 / wps;
    const int blocks = 256 * wps;
    const size_t lds = (wps == 1) ? 96 * 1024 : (wps == 2) ? 64 * 1024 : 40 * 1024; // pins blocks per CU
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, out, 200, seed);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, out, iters, seed);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double mfma_per_simd = (double)iters * 16 * wps;
    const double ns = best * 1e6 / mfma_per_simd;
    printf("%-12s %s waves/SIMD %d  %.3f ms  %.2f ns per MFMA per SIMD  (32 cycles -> %.2f GHz effective; %.0f TOPS)\n", name, seed == 2 ? "sparse" : seed == 3 ? "random" : "sift  ", wps,
           best, ns, 32.0 / ns, 65536.0 / ns * 1024 / 1e3);
}

int main(int argc, char **argv) {
    unsigned *out;
    CK(hipMalloc(&out, 256 * 4 * 256 * sizeof(unsigned)));
    if (argc > 1 && argv[1][0] == 's') { // "shape": 32x32x32 against 16x16x64, same MACs ("MFMA" = 32K MACs in both)
        for (int seed = 2; seed <= 4; seed++)
            for (int wps = 2; wps <= 3; wps++) {
                run("32x32x32 nv0", mfma_kernel<0, 0>, out, wps, seed);
                run("16x16x64 nv0", mfma16_kernel<0>, out, wps, seed);
                run("32x32x32 nv40", mfma_kernel<40, 1>, out, wps, seed);
                run("16x16x64 nv40", mfma16_kernel<40>, out, wps, seed);
            }
        CK(hipFree(out));
        return 0;
    }
    for (int seed = 2; seed <= 4; seed++)
        for (int wps = 1; wps <= 3; wps++) {
            run("nv0", mfma_kernel<0, 0>, out, wps, seed);
            run("nv16 il", mfma_kernel<16, 1>, out, wps, seed);
            run("nv16 tail", mfma_kernel<16, 0>, out, wps, seed);
            run("nv40 il", mfma_kernel<40, 1>, out, wps, seed);
            run("nv40 tail", mfma_kernel<40, 0>, out, wps, seed);
            run("nv64 il", mfma_kernel<64, 1>, out, wps, seed);
            if (wps == 3) continue; // 174 VGPRs: 2 waves per SIMD at most
            run("lds nv16", mfma_lds_kernel<16, 1>, out, wps, seed);
            run("lds nv40", mfma_lds_kernel<40, 1>, out, wps, seed);
        }
    CK(hipFree(out));
    return 0;
}

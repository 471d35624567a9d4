// Micro-probe (gfx950): issue rates that bound the kNN-2 epilogue.
//   mode 0: integer VALU only (v_lshl_add_u32 + v_med3_u32 + v_max_u32 per "distance")
//   mode 1: int8 MFMA only (v_mfma_i32_32x32x32_i8, 4 per tile)
//   mode 2: both in every wave (the kernel's shape: 4 MFMA then 48 VALU on the result)
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/valu_probe
// Run:   tools/valu_probe            (prints cycles per tile per SIMD for 1..8 waves/SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256, 2) probe(unsigned *out, int iters, unsigned seed) {
    const int lane = threadIdx.x;
    v4i a = {(int)(seed + lane), (int)(seed * 3 + lane), (int)(seed * 5 + lane), (int)(seed * 7 + lane)};
    v4i b = {(int)(seed * 11 + lane), (int)(seed * 13 + lane), (int)(seed * 17 + lane), (int)(seed * 19 + lane)};
    unsigned m1 = 0, m2 = 0, m1b = 0, m2b = 0;
    unsigned cb[16];
#pragma unroll
    for (int r = 0; r < 16; r++) cb[r] = seed * (r + 1) + lane;
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        if (MODE == 1 || MODE == 2) {
            acc = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; k++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
            a[0] += 1;
        }
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const unsigned src = (MODE == 0) ? (cb[r] + it) : (unsigned)acc[r];
                const unsigned key = (src << 9) + cb[r];
                unsigned lo = m1 < key ? m1 : key, hi = m1 < key ? key : m1;
                unsigned t2 = hi < m2 ? hi : m2;
                m2 = lo > t2 ? lo : t2;
                m1 = hi;
            }
        }
        if (MODE == 1) {
            m1 += acc[0];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = m1 + m2 + m1b + m2b;
}

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e = (x);                                                   \
        if (e != hipSuccess) {                                                \
            printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);   \
            exit(1);                                                          \
        }                                                                     \
    } while (0)

template <int MODE>
static void run(const char *name, unsigned *out) {
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        // 256-thread blocks = 1 wave per SIMD; wps blocks per CU
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double tiles_per_simd = (double)iters * wps;
        const double ns_per_tile = ms * 1e6 / tiles_per_simd;
        printf("%-10s waves/SIMD=%d  %.3f ms  %.1f ns per tile per SIMD  (= %.0f cycles @2.4GHz)\n", name, wps, ms,
               ns_per_tile, ns_per_tile * 2.4);
    }
}

int main() {
    unsigned *out;
    CK(hipMalloc(&out, 256 * 8 * 256 * sizeof(unsigned)));
    run<0>("valu48", out);
    run<1>("mfma4", out);
    run<2>("mfma+valu", out);
    CK(hipFree(out));
    return 0;
}

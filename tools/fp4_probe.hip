// Probe (gfx950): v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (E2M1) 0/1 nibbles -- operand layout (lane l: row l&31, 32 nibbles of
// k-half l>>5), unit scales (E8M0 0x7f) and exactness of the dot product, as used by knn2_count_kernel<4, 1>.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp4_probe.hip -o tools/bin/fp4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
// A: 32 rows x 64 k (fp4 nibbles, 32 bytes per row), B: 32 cols x 64 k.  lane l: row/col l&31, k-half l>>5 (16 bytes)
__global__ void k(const uint8_t* A, const uint8_t* B, float* C) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    v4i a4 = *(const v4i*)(A + r * 32 + h * 16);
    v4i b4 = *(const v4i*)(B + r * 32 + h * 16);
    v8i a = {a4[0], a4[1], a4[2], a4[3], 0, 0, 0, 0};
    v8i b = {b4[0], b4[1], b4[2], b4[3], 0, 0, 0, 0};
    v16f c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int i = 0; i < 16; i++) C[lane * 16 + i] = c[i];
}
int main() {
    std::vector<uint8_t> A(32 * 32), B(32 * 32);
    std::vector<int> abit(32 * 64), bbit(32 * 64);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 16) & 1; };
    for (int r = 0; r < 32; r++) for (int kk = 0; kk < 64; kk++) { abit[r * 64 + kk] = rnd(); bbit[r * 64 + kk] = rnd(); }
    for (int r = 0; r < 32; r++) for (int by = 0; by < 32; by++) {
        A[r * 32 + by] = (abit[r * 64 + 2 * by] ? 0x2 : 0) | (abit[r * 64 + 2 * by + 1] ? 0x20 : 0);
        B[r * 32 + by] = (bbit[r * 64 + 2 * by] ? 0x2 : 0) | (bbit[r * 64 + 2 * by + 1] ? 0x20 : 0);
    }
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 64 * 16 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<float> C(64 * 16);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; lane++) for (int i = 0; i < 16; i++) {
        const int col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        int dot = 0;
        for (int kk = 0; kk < 64; kk++) dot += abit[row * 64 + kk] * bbit[col * 64 + kk];
        if (C[lane * 16 + i] != (float)dot) { if (bad < 8) printf("lane %d i %d got %g want %d\n", lane, i, C[lane * 16 + i], dot); bad++; }
    }
    printf("fp4 mfma 32x32x64: %s (%d mismatches)  sample C[0]=%g\n", bad ? "MISMATCH" : "exact", bad, C[0]);
    return 0;
}

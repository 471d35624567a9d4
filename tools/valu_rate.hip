// Micro-probe (gfx950): issue rate of the individual integer VALU ops the kNN-2 epilogue can be built from.
// 8 independent chains per lane, 4 waves per SIMD, inline asm so the compiler cannot fuse or drop anything.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/bin/valu_rate
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

#define OP2(name, INSN)                                                                     \
    __global__ void __launch_bounds__(256) name(unsigned *out, int iters, unsigned seed) {  \
        unsigned r[8], x = seed + threadIdx.x, y = seed * 3 + threadIdx.x;                  \
        for (int i = 0; i < 8; i++) r[i] = seed * (i + 2) + threadIdx.x;                    \
        for (int it = 0; it < iters; it++) {                                                \
            _Pragma("unroll") for (int u = 0; u < 4; u++) {                                 \
                _Pragma("unroll") for (int i = 0; i < 8; i++) {                             \
                    asm volatile(INSN : "+v"(r[i]) : "v"(x), "v"(y));                       \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
        unsigned s = 0;                                                                     \
        for (int i = 0; i < 8; i++) s += r[i];                                              \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                            \
    }

OP2(k_max_u32, "v_max_u32 %0, %0, %1")
OP2(k_add_u32, "v_add_u32 %0, %0, %1")
OP2(k_med3_u32, "v_med3_u32 %0, %0, %1, %2")
OP2(k_max3_u32, "v_max3_u32 %0, %0, %1, %2")
OP2(k_lshl_add, "v_lshl_add_u32 %0, %0, 9, %1")
OP2(k_pk_max_u16, "v_pk_max_u16 %0, %0, %1")
OP2(k_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
OP2(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
OP2(k_perm, "v_perm_b32 %0, %0, %1, %2")
OP2(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
OP2(k_bfe, "v_bfe_u32 %0, %0, 3, 9")
OP2(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
OP2(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
OP2(k_cmp, "v_cmp_gt_u32 vcc, %0, %1")
OP2(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
OP2(k_max_f32, "v_max_f32 %0, %0, %1")
OP2(k_max_i16, "v_max_i16 %0, %0, %1")
OP2(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
OP2(k_sad_u16, "v_sad_u16 %0, %0, %1, %2")
OP2(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
OP2(k_xor, "v_xor_b32 %0, %0, %1")
OP2(k_mov_dpp, "v_max_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
OP2(k_sdwa, "v_max_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")


OP2(k_sub_u32, "v_sub_u32 %0, %0, %1")
OP2(k_max_i32, "v_max_i32 %0, %0, %1")
OP2(k_min_u32, "v_min_u32 %0, %0, %1")
OP2(k_max_u16, "v_max_u16 %0, %0, %1")
OP2(k_min_u16, "v_min_u16 %0, %0, %1")
OP2(k_and, "v_and_b32 %0, %0, %1")
OP2(k_or, "v_or_b32 %0, %0, %1")
OP2(k_lshlrev, "v_lshlrev_b32 %0, 1, %0")
OP2(k_lshrrev, "v_lshrrev_b32 %0, %1, %0")
OP2(k_ashrrev, "v_ashrrev_i32 %0, 31, %0")
OP2(k_mul_f32, "v_mul_f32 %0, %0, %1")
OP2(k_add_f32, "v_add_f32 %0, %0, %1")
OP2(k_sub_f32, "v_sub_f32 %0, %0, %1")
OP2(k_min_f32, "v_min_f32 %0, %0, %1")
OP2(k_fmac_f32, "v_fmac_f32 %0, %1, %2")
OP2(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
OP2(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
OP2(k_or3, "v_or3_b32 %0, %0, %1, %2")
OP2(k_xad, "v_xad_u32 %0, %0, %1, %2")
OP2(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 1")
OP2(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
OP2(k_mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2")
OP2(k_mov, "v_mov_b32 %0, %1")
OP2(k_bfi, "v_bfi_b32 %0, %0, %1, %2")
OP2(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80")
OP2(k_max3_f32, "v_max3_f32 %0, %0, %1, %2")
OP2(k_med3_f32, "v_med3_f32 %0, %0, %1, %2")
OP2(k_add_co, "v_add_co_u32 %0, vcc, %0, %1")
OP2(k_sub_co, "v_sub_co_u32 %0, vcc, %0, %1")
OP2(k_addc_co, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
OP2(k_cmp_f32, "v_cmp_gt_f32 vcc, %0, %1")
OP2(k_cmp_sgpr, "v_cmp_gt_u32 s[20:21], %0, %1")
OP2(k_max_f16, "v_max_f16 %0, %0, %1")
OP2(k_add_u16, "v_add_u16 %0, %0, %1")
OP2(k_sub_i32_vop3, "v_sub_i32 %0, %0, %1")
OP2(k_dot4, "v_dot4_i32_i8 %0, %1, %2, %0")
OP2(k_msad, "v_msad_u8 %0, %0, %1, %2")
OP2(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
// the image kernels' ops (round 2): byte -> float, float -> byte, floor / round
OP2(k_cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
OP2(k_cvt_f32_ubyte3, "v_cvt_f32_ubyte3 %0, %0")
OP2(k_rndne_f32, "v_rndne_f32 %0, %0")
OP2(k_floor_f32, "v_floor_f32 %0, %0")
OP2(k_fract_f32, "v_fract_f32 %0, %0")
OP2(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
OP2(k_cvt_pk_u8_f32, "v_cvt_pk_u8_f32 %0, %1, 1, %0")
OP2(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 3")
#define OP64(name, INSN)                                                                    \
    __global__ void __launch_bounds__(256) name(unsigned *out, int iters, unsigned seed) {  \
        double r[8], x = 1.0 + 1e-9 * (seed + threadIdx.x), y = 1e-7 * (seed + threadIdx.x); \
        unsigned z = seed + threadIdx.x;                                                    \
        for (int i = 0; i < 8; i++) r[i] = 1.0 + 1e-6 * (seed * (i + 2) + threadIdx.x);     \
        for (int it = 0; it < iters; it++) {                                                \
            _Pragma("unroll") for (int u = 0; u < 4; u++) {                                 \
                _Pragma("unroll") for (int i = 0; i < 8; i++) {                             \
                    asm volatile(INSN : "+v"(r[i]) : "v"(x), "v"(y), "v"(z));               \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
        double s = 0;                                                                       \
        for (int i = 0; i < 8; i++) s += r[i];                                              \
        out[blockIdx.x * 256 + threadIdx.x] = (unsigned)s;                                  \
    }
OP64(k_mul_f64, "v_mul_f64 %0, %0, %1")
OP64(k_add_f64, "v_add_f64 %0, %0, %2")
OP64(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
OP64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
OP64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %2")
OP64(k_lshr_b64, "v_lshrrev_b64 %0, 3, %0")
OP64(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %3")

__global__ void k_pk_u8_semantics(float *in, unsigned *out, int n) {
    const int i = threadIdx.x;
    if (i < n) {
        unsigned r = 0;
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r) : "v"(in[i]));
        out[i] = r;
    }
}

typedef void (*kern_t)(unsigned *, int, unsigned);

static void run(const char *name, kern_t k, unsigned *out) {
    const int iters = 4000, wps = 4;
    const int blocks = 256 * wps;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 50, 1u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = (double)iters * 32 * wps;
    printf("%-14s %.3f ms  %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

int main() {
    unsigned *out;
    CK(hipMalloc(&out, 256 * 8 * 256 * sizeof(unsigned)));
#define R(k) run(#k, k, out)
    R(k_max_u32);
    R(k_add_u32);
    R(k_med3_u32);
    R(k_max3_u32);
    R(k_lshl_add);
    R(k_pk_max_u16);
    R(k_pk_min_u16);
    R(k_pk_add_u16);
    R(k_perm);
    R(k_and_or);
    R(k_bfe);
    R(k_mad_u32_u24);
    R(k_cndmask);
    R(k_cmp);
    R(k_fma_f32);
    R(k_max_f32);
    R(k_max_i16);
    R(k_pk_max_i16);
    R(k_sad_u16);
    R(k_bcnt);
    R(k_xor);
    R(k_mov_dpp);
    R(k_sdwa);
    R(k_sub_u32);
    R(k_max_i32);
    R(k_min_u32);
    R(k_max_u16);
    R(k_min_u16);
    R(k_and);
    R(k_or);
    R(k_lshlrev);
    R(k_lshrrev);
    R(k_ashrrev);
    R(k_mul_f32);
    R(k_add_f32);
    R(k_sub_f32);
    R(k_min_f32);
    R(k_fmac_f32);
    R(k_add3_u32);
    R(k_lshl_or);
    R(k_or3);
    R(k_xad);
    R(k_add_lshl);
    R(k_mul_u32_u24);
    R(k_mad_i32_i24);
    R(k_mov);
    R(k_bfi);
    R(k_bitop3);
    R(k_max3_f32);
    R(k_med3_f32);
    R(k_add_co);
    R(k_sub_co);
    R(k_addc_co);
    R(k_cmp_f32);
    R(k_cmp_sgpr);
    R(k_max_f16);
    R(k_add_u16);
    R(k_sub_i32_vop3);
    R(k_dot4);
    R(k_msad);
    R(k_cvt_f32_i32);
    R(k_cvt_f32_ubyte0);
    R(k_cvt_f32_ubyte3);
    R(k_rndne_f32);
    R(k_floor_f32);
    R(k_fract_f32);
    R(k_cvt_i32_f32);
    R(k_cvt_pk_u8_f32);
    R(k_alignbyte);
    R(k_mul_f64);
    R(k_add_f64);
    R(k_fma_f64);
    R(k_pk_fma_f32);
    R(k_pk_add_f32);
    R(k_lshr_b64);
    R(k_cvt_f64_u32);
    {
        const float h[12] = {0.f, 0.4999f, 0.5f, 0.5001f, 1.5f, 2.5f, 3.5f, 254.5f, 255.4f, 255.5f, 300.f, -1.f};
        float *din;
        unsigned ho[12];
        CK(hipMalloc(&din, sizeof(h)));
        CK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_pk_u8_semantics, dim3(1), dim3(64), 0, 0, din, out, 12);
        CK(hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost));
        printf("v_cvt_pk_u8_f32 (default rounding mode):");
        for (int i = 0; i < 12; i++) printf("  %g -> %u", h[i], ho[i]);
        printf("\n");
    }
    CK(hipFree(out));
    return 0;
}

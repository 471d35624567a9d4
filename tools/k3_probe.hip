// Memory-side floor of the gradient blend (K3): byte-select  out = w ? w : r  over a 2030x1190 BGR region of a pitched
// mosaic, in the tilings / access widths the kernel could use.  No bookkeeping, no feather: what the loads and stores cost.
// Build: hipcc --offload-arch=gfx950 -O3 tools/k3_probe.hip -o tools/bin/k3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct __attribute__((packed, aligned(1))) u96u { uint32_t a, b, c; };
typedef uint32_t u32u __attribute__((aligned(1)));
struct __attribute__((packed, aligned(1))) u128u { uint32_t a, b, c, d; };

__device__ __forceinline__ uint32_t nz_bytes(uint32_t w) {
    const uint32_t a = (((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w) & 0x80808080u;
    return a | (a - (a >> 7));
}
__device__ __forceinline__ uint32_t sel(uint32_t w, uint32_t r) { return w | (r & ~nz_bytes(w)); }

// MODE 0: read w, read r, write; 1: read w + read r, write nothing (one lane writes a checksum); 2: write only; 3: read r, write (copy in place)
// tile: wave = TW pixels x (64 / (TW/4)) rows; workgroup = 4 waves side by side; IT row groups
template <int TW, int IT, int MODE>
__global__ void __launch_bounds__(256) k_tile(const uint8_t *__restrict__ wb, int64_t wp, uint8_t *mq, int64_t mp, int qw, int qh, uint32_t *sink) {
    constexpr int LPR = TW / 4, RPW = 64 / LPR; // lanes per row, rows per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lx = blockIdx.x * (4 * TW) + wave * TW + 4 * (lane % LPR);
    const int ly0 = blockIdx.y * (RPW * IT) + lane / LPR;
    u96u w[IT], r[IT];
    bool ok[IT];
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int ly = ly0 + RPW * it;
        ok[it] = lx + 4 <= qw && ly < qh;
        w[it] = u96u{0, 0, 0}, r[it] = u96u{0, 0, 0};
        if (ok[it]) {
            if (MODE != 2 && MODE != 3) w[it] = *(const u96u *)(wb + ly * wp + lx * 3);
            if (MODE != 2) r[it] = *(const u96u *)(mq + ly * mp + lx * 3);
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int ly = ly0 + RPW * it;
        u96u o;
        o.a = sel(w[it].a, r[it].a), o.b = sel(w[it].b, r[it].b), o.c = sel(w[it].c, r[it].c);
        if (MODE == 1) acc += o.a ^ o.b ^ o.c;
        else if (ok[it]) *(u96u *)(mq + ly * mp + lx * 3) = o;
    }
    if (MODE == 1 && acc == 0x12345678u) sink[0] = acc;
}

// 16-byte accesses aligned on the MOSAIC side: a lane owns 16 bytes of a mosaic row (byte granular, not pixel granular);
// the warped row is read unaligned at the same byte offset.  Wave = 1024 B of one row.
template <int IT>
__global__ void __launch_bounds__(256) k_bytes(const uint8_t *__restrict__ wb, int64_t wp, uint8_t *mq, int64_t mp, int qwb, int qh, int lead) {
    // mq + lead is 16-byte aligned for every row (mp % 16 == 0); bytes [0, lead) and the tail are skipped here (a real
    // kernel would give them to a narrow path)
    const int bx = lead + (blockIdx.x * 256 + threadIdx.x) * 16;
    const int ly0 = blockIdx.y * IT;
    uint4 w[IT], r[IT];
    bool ok[IT];
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int ly = ly0 + it;
        ok[it] = bx + 16 <= qwb && ly < qh;
        if (ok[it]) {
            const u128u t = *(const u128u *)(wb + ly * wp + bx);
            w[it] = make_uint4(t.a, t.b, t.c, t.d);
            r[it] = *(const uint4 *)(mq + ly * mp + bx);
        }
    }
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int ly = ly0 + it;
        if (ok[it]) *(uint4 *)(mq + ly * mp + bx) = make_uint4(sel(w[it].x, r[it].x), sel(w[it].y, r[it].y), sel(w[it].z, r[it].z), sel(w[it].w, r[it].w));
    }
}

template <typename F> static float time_us(int reps, F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; i++) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / reps;
}

int main() {
    const int qw = 2030, qh = 1190, CW = 5760, CH = 3240, dx0 = 2170, dy0 = 1040;
    const int64_t wp = (int64_t)qw * 3, mp = (int64_t)CW * 3;
    uint8_t *wb, *arena; uint32_t *sink;
    CK(hipMalloc(&wb, wp * qh + 64)); CK(hipMalloc(&arena, mp * CH)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(wb, 77, wp * qh)); CK(hipMemset(arena, 90, mp * CH));
    uint8_t *mq = arena + dy0 * mp + dx0 * 3;
    const double mb = 3.0 * wp * qh / 1e6;
#define RUN(NAME, TW, IT, MODE)                                                                                              \
    {                                                                                                                        \
        constexpr int RPW = 64 / (TW / 4);                                                                                   \
        dim3 g((qw + 4 * TW - 1) / (4 * TW), (qh + RPW * IT - 1) / (RPW * IT));                                              \
        float t = time_us(200, [&] { hipLaunchKernelGGL((k_tile<TW, IT, MODE>), g, dim3(256), 0, 0, wb, wp, mq, mp, qw, qh, sink); }); \
        printf("%-44s %4d x %4d wgs  %7.2f us\n", NAME, g.x, g.y, t);                                                        \
    }
    printf("region %d x %d BGR, %.1f MB per pass (2 reads + 1 write)\n", qw, qh, mb);
    RUN("tile 64x4, IT=1, read w + r, write", 64, 1, 0)
    RUN("tile 64x4, IT=2, read w + r, write", 64, 2, 0)
    RUN("tile 64x4, IT=4, read w + r, write", 64, 4, 0)
    RUN("tile 64x4, IT=8, read w + r, write", 64, 8, 0)
    RUN("tile 128x2, IT=2, read w + r, write", 128, 2, 0)
    RUN("tile 256x1, IT=2, read w + r, write", 256, 2, 0)
    RUN("tile 256x1, IT=4, read w + r, write", 256, 4, 0)
    RUN("tile 64x4, IT=2, reads only", 64, 2, 1)
    RUN("tile 64x4, IT=2, write only", 64, 2, 2)
    RUN("tile 64x4, IT=2, read r + write (copy in place)", 64, 2, 3)
    RUN("tile 256x1, IT=2, reads only", 256, 2, 1)
    RUN("tile 256x1, IT=2, write only", 256, 2, 2)
    {
        const int lead = (int)((16 - ((uintptr_t)mq & 15)) & 15), qwb = qw * 3;
        dim3 g1((qwb + 4095) / 4096, qh), g2((qwb + 4095) / 4096, (qh + 1) / 2), g4((qwb + 4095) / 4096, (qh + 3) / 4);
        float t = time_us(200, [&] { hipLaunchKernelGGL((k_bytes<1>), g1, dim3(256), 0, 0, wb, wp, mq, mp, qwb, qh, lead); });
        printf("%-44s %4d x %4d wgs  %7.2f us\n", "16-byte lanes aligned on the mosaic, IT=1", g1.x, g1.y, t);
        t = time_us(200, [&] { hipLaunchKernelGGL((k_bytes<2>), g2, dim3(256), 0, 0, wb, wp, mq, mp, qwb, qh, lead); });
        printf("%-44s %4d x %4d wgs  %7.2f us\n", "16-byte lanes aligned on the mosaic, IT=2", g2.x, g2.y, t);
        t = time_us(200, [&] { hipLaunchKernelGGL((k_bytes<4>), g4, dim3(256), 0, 0, wb, wp, mq, mp, qwb, qh, lead); });
        printf("%-44s %4d x %4d wgs  %7.2f us\n", "16-byte lanes aligned on the mosaic, IT=4", g4.x, g4.y, t);
    }
    float t = time_us(200, [&] { CK(hipMemcpy2DAsync(mq, mp, wb, wp, wp, qh, hipMemcpyDeviceToDevice, 0)); });
    printf("%-44s                  %7.2f us\n", "hipMemcpy2DAsync wbuf -> mosaic region", t);
    t = time_us(200, [&] { hipLaunchKernelGGL((k_tile<64, 1, 2>), dim3(1, 1), dim3(256), 0, 0, wb, wp, mq, mp, qw, qh, sink); });
    printf("%-44s                  %7.2f us\n", "one-workgroup kernel (launch floor)", t);
    return 0;
}

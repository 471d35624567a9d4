// What a synchronous library call pays to wait for the GPU: kernel + small D2H copy into pinned memory, then
// hipStreamSynchronize against a spin on hipStreamQuery / hipEventQuery, and against a kernel that writes its (one int per
// workgroup) results into pinned host memory itself.
// Build: hipcc --offload-arch=gfx950 -O3 tools/sync_probe.hip -o tools/bin/sync_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_small(int *p, int spin) {
    int v = 0;
    for (int i = 0; i < spin; i++) v += __builtin_amdgcn_s_memtime() & 1;
    if (threadIdx.x == 0) p[blockIdx.x] = v;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    int *d, *h;
    CK(hipMalloc(&d, 1 << 20)); CK(hipHostMalloc(&h, 1 << 20));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int spin : {0, 2000}) {          // ~1 us and ~25 us kernels
        for (int bytes : {128, 60000}) {
            for (int mode = 0; mode < 4; mode++) {
                double best = 1e9, sum = 0;
                const int reps = 300;
                for (int r = 0; r < reps + 20; r++) {
                    const double t0 = now_us();
                    // mode 3: the kernel writes its results straight into pinned host memory, no copy command
                    hipLaunchKernelGGL(k_small, dim3(mode == 3 ? (bytes + 3) / 4 / 6 + 1 : 256), dim3(256), 0, st, mode == 3 ? h : d, spin);
                    if (mode != 3) CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st));
                    if (mode == 0 || mode == 3) CK(hipStreamSynchronize(st));
                    else if (mode == 1) { while (hipStreamQuery(st) == hipErrorNotReady) {} }
                    else { CK(hipEventRecord(ev, st)); while (hipEventQuery(ev) == hipErrorNotReady) {} }
                    const double t = now_us() - t0;
                    if (r >= 20) { sum += t; if (t < best) best = t; }
                }
                printf("kernel spin %4d, D2H %5d B, %-22s mean %6.1f us  best %6.1f us\n", spin, bytes,
                       mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "spin hipStreamQuery" : mode == 2 ? "record + spin EventQuery" : "kernel writes host, sync", sum / reps, best);
            }
        }
    }
    return 0;
}

// What clock does ONE wave get when the rest of the chip idles (the serial MSS walks: one wave per stretch)?
// A chain of N dependent v_fma_f32 (one per 4+ cycles... measured against s_memrealtime, 100 MHz) alone, and beside a
// grid that keeps every CU busy.   hipcc --offload-arch=gfx950 -O3 single_wave_clock.hip -o swc && ./swc
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void chain(float *out, int n, unsigned long long *ticks)
{
    float x = threadIdx.x * 1e-9f, a = 1.0000001f, b = 1e-7f;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) x = __builtin_fmaf(x, a, b);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) *ticks = t1 - t0;
}
__global__ void busy(float *out, int n)
{
    float x = threadIdx.x, y = blockIdx.x;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) { x = __builtin_fmaf(x, 1.0000001f, y); y = __builtin_fmaf(y, 0.9999999f, x); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}
int main()
{
    float *d, *d2; unsigned long long *t, h;
    hipMalloc(&d, 4096); hipMalloc(&d2, 4 * 1024 * 256 * 8); hipMalloc(&t, 8);
    const int n = 20000;                       // 1.28 M dependent fmas
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, d, n, t);
        hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        printf("alone:        %.3f ms for %d dependent fma -> %.2f ns each\n", h / 1e5, n * 64, h * 10.0 / (n * 64.0));
    }
    hipStream_t s2; hipStreamCreate(&s2);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, s2, d2, 40000);
        hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, d, n, t);
        hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        hipStreamSynchronize(s2);
        printf("beside a busy grid: %.3f ms -> %.2f ns each\n", h / 1e5, h * 10.0 / (n * 64.0));
    }
    return 0;
}

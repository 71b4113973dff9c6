// Microbenchmark: VALU issue rates on gfx950 for the GRU gate mix, by waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed)
{
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) v[i] = v[i] * 1.0001f + 0.5f;                                  // 1 fma
            if (MODE == 1) v[i] = __builtin_amdgcn_exp2f(v[i]);                             // 1 trans
            if (MODE == 2) v[i] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[i]));  // exp, add, rcp
            if (MODE == 3) { float a = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[i])); v[i] = a * 1.0001f + (v[i] - a) * 0.5f; }  // 2 trans + 4 simple
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, int ops_simple, int ops_trans)
{
    float *d; hipMalloc(&d, 1 << 24);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int iters = 2000;
    for (int wps = 1; wps <= 8; wps *= 2) {          // waves per SIMD: blocks of 256 threads = 1 wave per SIMD
        int grid = p.multiProcessorCount * wps;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 0.25f);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 0.25f);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double cyc = ms * 1e-3 * 2.4e9;              // at 2.4 GHz nominal
        double per_wave_iter = cyc / iters / 16.0;   // cycles per element-iteration as seen by the SIMD (all its waves)
        printf("%-28s waves/SIMD %d: %.3f ms  -> %.1f SIMD-cycles per 16-element group per wave-set, %.2f cycles per wave-instruction\n", name, wps, ms,
               cyc / iters, cyc / iters / (16.0 * (ops_simple + ops_trans) * wps));
    }
    hipFree(d);
}
int main()
{
    run<0>("fma", 1, 0);
    run<1>("exp2", 0, 1);
    run<2>("exp2+add+rcp", 1, 2);
    run<3>("2 trans + 4 simple", 4, 2);
    return 0;
}

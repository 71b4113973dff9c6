// Does v_fma_mix_f32 widen SUBNORMAL fp16 inputs exactly (the split kernels compute h - fp16(h) with it)?
// Every fp16 bit pattern (hi) against fma_mix(hi, -1, h) with h = float(hi) * (1 + 2^-13): expected h - float(hi) bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const float *h, const unsigned *hi, float *lo, float *hi32)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float r = h[i];
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(r) : "v"(hi[i]));
    float r2 = h[i];
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r2) : "v"(hi[i] << 16));
    lo[i] = r; hi32[i] = r2;
}
int main()
{
    const int n = 65536;
    std::vector<float> h(n), want(n);
    std::vector<unsigned> hi(n);
    for (int i = 0; i < n; ++i) {
        _Float16 x; unsigned short b = (unsigned short)i; memcpy(&x, &b, 2);
        hi[i] = b;
        const float f = (float)x;
        h[i] = f * (1.0f + 1.0f / 8192.0f);
        want[i] = h[i] - f;
    }
    float *dh, *dlo, *dl2; unsigned *dhi;
    hipMalloc(&dh, n * 4); hipMalloc(&dlo, n * 4); hipMalloc(&dl2, n * 4); hipMalloc(&dhi, n * 4);
    hipMemcpy(dh, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dhi, hi.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dh, dhi, dlo, dl2);
    std::vector<float> lo(n), l2(n);
    hipMemcpy(lo.data(), dlo, n * 4, hipMemcpyDeviceToHost); hipMemcpy(l2.data(), dl2, n * 4, hipMemcpyDeviceToHost);
    int bad = 0, bad_sub = 0;
    for (int i = 0; i < n; ++i) {
        if ((i & 0x7c00) == 0x7c00) continue;             // inf / nan
        const bool ok = memcmp(&lo[i], &want[i], 4) == 0 && memcmp(&l2[i], &want[i], 4) == 0;
        if (!ok) { ++bad; if ((i & 0x7c00) == 0) ++bad_sub; if (bad < 5) printf("bits %04x: got %g / %g want %g\n", i, lo[i], l2[i], want[i]); }
    }
    printf("fma_mix residual: %d mismatches of 63488 finite fp16 values (%d of them subnormal inputs)\n", bad, bad_sub);
    return bad != 0;
}

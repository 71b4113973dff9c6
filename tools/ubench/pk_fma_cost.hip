// v_pk_*_f32 on gfx950: what do inline constants mean for the high half, and what does an op cost next to an MFMA stream?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
__global__ void probe(float *o)
{
    f32x2 a = { 3.0f + threadIdx.x, 5.0f }, b = { 7.0f, 11.0f }, r0, r1, r2, r3, r4;
    asm volatile("v_pk_add_f32 %0, %1, 1.0" : "=v"(r0) : "v"(a));
    asm volatile("v_pk_add_f32 %0, %1, 1.0 op_sel_hi:[1,0]" : "=v"(r1) : "v"(a));
    asm volatile("v_pk_fma_f32 %0, %1, 0.5, 1.0 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(a));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r3) : "v"(a), "v"(b), "v"(a));
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r4) : "v"(a), "v"(b));
    if (threadIdx.x == 0) { o[0] = r0.x; o[1] = r0.y; o[2] = r1.x; o[3] = r1.y; o[4] = r2.x; o[5] = r2.y; o[6] = r3.x; o[7] = r3.y; o[8] = r4.x; o[9] = r4.y; }
}
// port cost: per gap one MFMA 16x16x32 + NP packed fma + NF plain fma + NE exp, one wave per SIMD
template <int NP, int NF, int NE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) cost(int iters, unsigned long long *cyc, float *sink)
{
    f32x4 acc0 = { 0, 0, 0, 0 }, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
    f32x2 p0 = { 1.0f, 2.0f }, p1 = p0, p2 = p0, p3 = p0, c = { 0.999f, 0.998f }, d = { 0.001f, 0.002f };
    float y0 = 1, y1 = 2, y2 = 3, y3 = 4, x0 = 0.1f, x1 = 0.2f, x2 = 0.3f, x3 = 0.4f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(k == 0 ? acc0 : k == 1 ? acc1 : k == 2 ? acc2 : acc3) : "v"(a), "v"(b));
            if (NP > 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(c), "v"(d));
            if (NF > 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y0) : "v"(c.x), "v"(d.x));
            if (NE > 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x0));
            if (NP > 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(c), "v"(d));
            if (NF > 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y1) : "v"(c.x), "v"(d.x));
            if (NE > 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x1));
            if (NP > 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(c), "v"(d));
            if (NF > 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y2) : "v"(c.x), "v"(d.x));
            if (NP > 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(c), "v"(d));
            if (NF > 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y3) : "v"(c.x), "v"(d.x));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = p0.x + p1.y + p2.x + p3.y + y0 + y1 + y2 + y3 + x0 + x1 + x2 + x3 + acc0[0] + acc1[1] + acc2[2] + acc3[3];
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
    if (r == 1234.5f) sink[0] = r;
}
template <int NP, int NF, int NE> void run(const char *name, unsigned long long *d, float *sink)
{
    const int iters = 20000;
    hipLaunchKernelGGL((cost<NP, NF, NE>), dim3(256), dim3(256), 0, 0, iters, d, sink);
    hipLaunchKernelGGL((cost<NP, NF, NE>), dim3(256), dim3(256), 0, 0, iters, d, sink);
    unsigned long long h[4];
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("%-34s %.2f ticks per gap\n", name, (double)h[0] / (iters * 4.0));
}
int main()
{
    float *o, h[10]; hipMalloc(&o, 64);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, o);
    hipMemcpy(h, o, 40, hipMemcpyDeviceToHost);
    printf("a = (3, 5), b = (7, 11)\n pk_add a, 1.0            -> (%g, %g)\n pk_add a, 1.0 op_sel_hi:[1,0] -> (%g, %g)\n pk_fma a, 0.5, 1.0 op_sel_hi:[1,0,0] -> (%g, %g)\n pk_fma a, b, -a -> (%g, %g)\n pk_mul a, b -> (%g, %g)\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9]);
    unsigned long long *d; float *sink; hipMalloc(&d, 64); hipMalloc(&sink, 4);
    run<0, 0, 0>("mfma only", d, sink);
    run<0, 1, 0>("mfma + 1 fma", d, sink);
    run<0, 2, 0>("mfma + 2 fma", d, sink);
    run<0, 3, 0>("mfma + 3 fma", d, sink);
    run<0, 4, 0>("mfma + 4 fma", d, sink);
    run<1, 0, 0>("mfma + 1 pk_fma", d, sink);
    run<2, 0, 0>("mfma + 2 pk_fma", d, sink);
    run<3, 0, 0>("mfma + 3 pk_fma", d, sink);
    run<4, 0, 0>("mfma + 4 pk_fma", d, sink);
    run<0, 0, 1>("mfma + 1 exp", d, sink);
    run<0, 0, 2>("mfma + 2 exp", d, sink);
    run<0, 2, 1>("mfma + 1 exp + 2 fma", d, sink);
    run<1, 0, 1>("mfma + 1 exp + 1 pk_fma", d, sink);
    run<0, 4, 2>("mfma + 2 exp + 4 fma", d, sink);
    run<2, 0, 2>("mfma + 2 exp + 2 pk_fma", d, sink);
    return 0;
}

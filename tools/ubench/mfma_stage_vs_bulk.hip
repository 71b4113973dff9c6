// Microbenchmark for the two-tile split kernel's question: with ONE wave per SIMD, does vector work hide
// behind that wave's own MFMAs when it is staged between them, and not when it follows a back-to-back group?
// Per group: nine v_mfma_f32_32x32x16_f16 on three independent accumulators (A operand from VGPRs or from AGPRs) and
// twelve transcendentals + seven packed/plain VALU ops (one gate slice of the kernel), either after the nine MFMAs
// ("bulk") or two or three behind each of the first seven MFMAs ("staged").  Inline asm only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA_V(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_A(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[0:3], %1, %0" : "+v"(acc) : "v"(b))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x))
#define FMA(y) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(c1), "v"(c2))

// MODE 0: MFMAs only; 1: VALU only; 2: bulk (9 MFMAs, then the slice); 3: staged; AG: A operand from AGPRs
template <int MODE, bool AG>
__device__ __forceinline__ float body(int iters, float seed)
{
    f32x16 r0 = { 0 }, r1 = { 0 }, r2 = { 0 };
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
    if (AG) asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_write_b32 a3, %0" :: "v"(seed) : "a3");
    float x0 = seed, x1 = seed + 1, y0 = seed, y1 = seed + 1;
    const float c1 = 0.999f, c2 = 0.5f;
    for (int it = 0; it < iters; ++it) {
#define M(acc) do { if (MODE != 1) { if (AG) MFMA_A(acc); else MFMA_V(acc); } } while (0)
#define S0 do { if (MODE == 3) { EXP(x0); EXP(x1); } } while (0)
#define S1 do { if (MODE == 3) { FMA(y0); RCP(x0); RCP(x1); } } while (0)
        M(r0); S0; M(r1); S1; M(r2); S0; M(r0); S1; M(r1); S0; M(r2); S1; M(r0); if (MODE == 3) { FMA(y0); FMA(y1); FMA(y0); FMA(y1); } M(r1); M(r2);
        if (MODE == 1 || MODE == 2) {
            EXP(x0); EXP(x1); FMA(y0); RCP(x0); RCP(x1); EXP(x0); EXP(x1); FMA(y0); RCP(x0); RCP(x1); EXP(x0); EXP(x1); FMA(y0); RCP(x0); RCP(x1);
            FMA(y0); FMA(y1); FMA(y0); FMA(y1);
        }
#undef M
#undef S0
#undef S1
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float r = x0 + x1 + y0 + y1;
    for (int i = 0; i < 16; ++i) r += r0[i] + r1[i] + r2[i];
    return r;
}

template <int MODE, bool AG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) bench(int iters, uint64_t *cyc, float *sink)
{
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    const float r = body<MODE, AG>(iters, threadIdx.x * 1e-3f);
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (r == 12345.678f) sink[0] = r;
}

template <int MODE, bool AG>
static void run(const char *name, uint64_t *d, float *sink)
{
    const int iters = 4000, blocks = 256;
    hipLaunchKernelGGL((bench<MODE, AG>), dim3(blocks), dim3(256), 0, 0, iters, d, sink);
    hipLaunchKernelGGL((bench<MODE, AG>), dim3(blocks), dim3(256), 0, 0, iters, d, sink);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(blocks * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    // s_memtime counts at 100 MHz; report in units of "ticks per group" and relative to the MFMA-only case
    printf("%-34s %9.3f memtime ticks per group of 9 MFMAs\n", name, s / h.size() / iters);
}

int main()
{
    uint64_t *d; float *sink;
    hipMalloc(&d, 256 * 4 * 8); hipMalloc(&sink, 4);
    run<0, false>("MFMAs only (VGPR A)", d, sink);
    run<0, true>("MFMAs only (AGPR A)", d, sink);
    run<1, false>("gate slice only (19 VALU)", d, sink);
    run<2, false>("bulk: 9 MFMAs then slice (VGPR A)", d, sink);
    run<3, false>("staged between MFMAs (VGPR A)", d, sink);
    run<2, true>("bulk (AGPR A)", d, sink);
    run<3, true>("staged (AGPR A)", d, sink);
    return 0;
}

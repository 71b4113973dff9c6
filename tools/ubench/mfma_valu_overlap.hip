// Microbenchmark (inline asm, nothing for the compiler to move): how much VALU / transcendental work
// hides behind v_mfma_f32_32x32x16_f16 on gfx950 -- inside one wave (fillers in the MFMA gaps) and across
// the two waves of a SIMD.  512-thread workgroups (2 waves per SIMD), one per CU; waves 0-3 take role A,
// waves 4-7 role B.  A role is "8 x { [MFMA] + NE v_exp_f32 + NF v_fma_f32 }" per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool M, int NE, int NF>
__device__ __forceinline__ float body(int iters, float seed)
{
    f32x16 acc = { 0 };
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
    float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, y0 = seed, y1 = seed + 1, y2 = seed + 2, y3 = seed + 3, y4 = seed + 4, y5 = seed + 5;
    const float c1 = 0.999f, c2 = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (M) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            if (NE > 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x0));
            if (NF > 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y0) : "v"(c1), "v"(c2));
            if (NE > 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x1));
            if (NF > 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y1) : "v"(c1), "v"(c2));
            if (NE > 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x2));
            if (NF > 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y2) : "v"(c1), "v"(c2));
            if (NE > 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x3));
            if (NF > 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y3) : "v"(c1), "v"(c2));
            if (NF > 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y4) : "v"(c1), "v"(c2));
            if (NF > 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y5) : "v"(c1), "v"(c2));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float r = x0 + x1 + x2 + x3 + y0 + y1 + y2 + y3 + y4 + y5;
    for (int i = 0; i < 16; ++i) r += acc[i];
    return r;
}

struct role { const char *name; int id; };
__device__ float run_role(int id, int iters, float seed)
{
    switch (id) {
    case 1: return body<true, 0, 0>(iters, seed);
    case 2: return body<false, 4, 0>(iters, seed);
    case 3: return body<false, 0, 4>(iters, seed);
    case 4: return body<true, 1, 3>(iters, seed);
    case 5: return body<true, 2, 2>(iters, seed);
    case 6: return body<true, 3, 0>(iters, seed);
    case 7: return body<true, 0, 6>(iters, seed);
    case 8: return body<true, 2, 4>(iters, seed);
    case 9: return body<true, 3, 3>(iters, seed);
    case 10: return body<true, 4, 4>(iters, seed);
    case 11: return body<false, 2, 2>(iters, seed);
    case 12: return body<true, 1, 0>(iters, seed);
    case 13: return body<true, 0, 3>(iters, seed);
    default: return 0.0f;
    }
}

__global__ void __launch_bounds__(512) bench(int roleA, int roleB, int prioA, int prioB, int iters, uint64_t *cyc, float *sink)
{
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? roleA : roleB;
    const int prio = wave < 4 ? prioA : prioB;
    if (prio == 3) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    const float r = run_role(role, iters, threadIdx.x * 1e-3f);
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (r == 12345.678f) sink[0] = r;
}

int main()
{
    const int iters = 2000, blocks = 256;
    uint64_t *d; float *sink;
    (void)hipMalloc(&d, blocks * 8 * 8); (void)hipMalloc(&sink, 4);
    const char *names[] = { "idle", "mfma", "4exp", "4fma", "mfma+1e+3f", "mfma+2e+2f", "mfma+3e", "mfma+6f", "mfma+2e+4f", "mfma+3e+3f", "mfma+4e+4f", "2e+2f", "mfma+1e", "mfma+3f" };
    const int pairs[][4] = { {1,0,0,0}, {2,0,0,0}, {3,0,0,0}, {11,0,0,0}, {12,0,0,0}, {13,0,0,0}, {4,0,0,0}, {5,0,0,0}, {6,0,0,0}, {7,0,0,0}, {8,0,0,0}, {9,0,0,0}, {10,0,0,0},
                             {1,1,0,0}, {2,2,0,0}, {3,3,0,0}, {4,4,0,0}, {5,5,0,0}, {8,8,0,0}, {9,9,0,0},
                             {1,3,0,0}, {3,1,0,0}, {1,3,0,3}, {1,2,0,0}, {2,1,0,0}, {1,11,0,0}, {11,1,0,0}, {2,3,0,0} };
    printf("cycles per GAP (one eighth of an iteration); B's time runs to the end of both roles when A has priority\n");
    for (auto &pr : pairs) {
        bench<<<blocks, 512>>>(pr[0], pr[1], pr[2], pr[3], iters, d, sink);
        (void)hipDeviceSynchronize();
        std::vector<uint64_t> h(blocks * 8);
        (void)hipMemcpy(h.data(), d, blocks * 64, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int i = 0; i < blocks; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += h[i * 8 + w];
        a /= blocks * 4.0 * iters * 8; b /= blocks * 4.0 * iters * 8;
        printf("A=%-12s(prio %d) B=%-12s(prio %d)   A %.1f  B %.1f\n", names[pr[0]], pr[2], names[pr[1]], pr[3], a, b);
    }
    return 0;
}

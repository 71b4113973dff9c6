// Would the recurrent kernel gain from two waves per SIMD with different roles?  (one wave per SIMD issues 150 MFMAs,
// 80 transcendentals, ~245 plain vector operations and ~52 LDS operations per tile-step through one in-order instruction stream:
// 3 860 cycles, matrix pipe 62 % busy.)  Here the same work per SIMD and tile-step is split over two waves of 256 registers:
//   matrix wave: U_hi resident (96 AGPRs), passes U_hi.h_hi and U_hi.h_lo (96 MFMAs 16x16x32), hands its 12 accumulator tiles over in LDS
//   gate wave:   U_lo resident (48 AGPRs), pass U_lo.h_hi (48 MFMAs), adds the partner's tiles, runs 16 gate chains per lane
//                (5 transcendentals + 7 plain operations each), publishes h as fp16 hi/lo, one workgroup barrier per tile-step;
//                the two waves work on DIFFERENT tiles (the matrix wave on tile X's step while the gate wave finishes tile Y's), as the
//                kernel's two row tiles do today inside one wave
// against ONE wave doing all of it (compiler-scheduled, so slower than the hand-interleaved kernel: the two-role figure is to be read
// against the real kernel's 3 860 cycles / 1.75 us per tile-step).  Random operands, all CUs busy: the power limit is part of the answer.
//   usage: two_role [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LOADW(a, p) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&a"(a) : "v"(p) : "memory")
#define MFMA16(acc, W, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(W), "v"(b))

__device__ __forceinline__ float gate(float ar, float ag, float az, float ax, float s)
{
    float er = __builtin_amdgcn_exp2f(ar), e2 = __builtin_amdgcn_exp2f(az);
    er = __builtin_amdgcn_rcpf(er + 1.0f);
    const float g = __builtin_fmaf(er, ag, ax);
    const float A = __builtin_amdgcn_exp2f(g) + 1.0f;
    const float zt = __builtin_fmaf(0.5f, e2, 1.0f);
    const float d = __builtin_amdgcn_rcpf(A * zt);
    return __builtin_fmaf(s, A, -e2) * d;
}

// LDS: h fragments [hi|lo][k-step 4][row half 2][64 lanes] x 16 B = 16 KB; accumulator hand-over [4 pairs][12 tiles][64 lanes] x 16 B = 48 KB
struct lds_t { uint4 frag[2][4][2][64]; f32x4 acc[2][4][12][64]; unsigned hpub[2][4][16][64]; };   // acc: two tile-steps in flight

template <int ROLES>
__global__ void __launch_bounds__(ROLES == 2 ? 512 : 256) __attribute__((amdgpu_waves_per_eu(ROLES == 2 ? 2 : 1, ROLES == 2 ? 2 : 1)))
bench(const uint4 *w, const uint4 *hinit, float *sink, int iters, uint64_t *cyc)
{
    __shared__ lds_t L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pair = wave & 3;
    const bool matrix = ROLES == 2 ? wave < 4 : true, gates = ROLES == 2 ? wave >= 4 : true;
    for (int i = threadIdx.x; i < 2 * 4 * 2 * 64; i += blockDim.x) (&L.frag[0][0][0][0])[i] = hinit[i];
    // one register array for both roles (a wave has ONE role: the matrix wave's U_hi fragments -- 3 gates x 4 k-steps x 2 unit
    // halves -- and the gate wave's U_lo fragments share the registers); the one-wave form needs both
    u32x4 Whi[24], Wlo[ROLES == 2 ? 1 : 12];
#pragma unroll
    for (int i = 0; i < 24; ++i) LOADW(Whi[i], w + ((size_t)pair * 36 + (ROLES == 2 && !matrix ? 24 + i % 12 : i)) * 64 + lane);
    if (ROLES == 1) {
#pragma unroll
        for (int i = 0; i < 12; ++i) LOADW(Wlo[i], w + ((size_t)pair * 36 + 24 + i) * 64 + lane);
    }
    float h[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) h[i] = 0.01f * i;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 c[3][4];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) c[g][q] = f32x4{ 0.01f * g, 0.02f * q, 0.0f, 0.01f };
        if (matrix) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                half8 hf[2], lf[2];
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
                    hf[rh] = __builtin_bit_cast(half8, L.frag[0][k][rh][lane]);
                    lf[rh] = __builtin_bit_cast(half8, L.frag[1][k][rh][lane]);
                }
#pragma unroll
                for (int pass = 0; pass < 2; ++pass)
#pragma unroll
                    for (int g = 0; g < 3; ++g)
#pragma unroll
                        for (int uh = 0; uh < 2; ++uh)
#pragma unroll
                            for (int rh = 0; rh < 2; ++rh) MFMA16(c[g][2 * uh + rh], Whi[8 * g + 2 * k + uh], pass ? lf[rh] : hf[rh]);
            }
            if (ROLES == 2) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int q = 0; q < 4; ++q) L.acc[it & 1][pair][4 * g + q][lane] = c[g][q];
            }
        }
        f32x4 d[3][4];
        if (gates) {
            if (ROLES == 2) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int q = 0; q < 4; ++q) d[g][q] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                half8 hf[2];
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) hf[rh] = __builtin_bit_cast(half8, L.frag[0][k][rh][lane]);
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int uh = 0; uh < 2; ++uh)
#pragma unroll
                        for (int rh = 0; rh < 2; ++rh) {
                            if (ROLES == 2) MFMA16(d[g][2 * uh + rh], Whi[4 * g + k], hf[rh]);
                            else MFMA16(c[g][2 * uh + rh], Wlo[4 * g + k], hf[rh]);
                        }
            }
        }
        if (gates) {
            if (ROLES == 2) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int q = 0; q < 4; ++q) c[g][q] = d[g][q] + L.acc[(it + 1) & 1][pair][4 * g + q][lane];   // the OTHER tile's step, handed over one barrier ago
            }
            unsigned pub[2][4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < 4; ++i) h[4 * q + i] = gate(c[0][q][i], c[1][q][i], c[2][q][i], 0.1f * i, h[4 * q + i]);
                // hi / lo halves as the kernel publishes them: 2 packs, 4 residuals, 2 packs per 4 values
                float r[4] = { h[4 * q] + 1.0f, h[4 * q + 1] + 1.0f, h[4 * q + 2] + 1.0f, h[4 * q + 3] + 1.0f };
                asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pub[0][q][0]) : "v"(r[0]), "v"(r[1]));
                asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pub[0][q][1]) : "v"(r[2]), "v"(r[3]));
                asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(r[0]) : "v"(pub[0][q][0]));
                asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r[1]) : "v"(pub[0][q][0]));
                asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(r[2]) : "v"(pub[0][q][1]));
                asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r[3]) : "v"(pub[0][q][1]));
                asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pub[1][q][0]) : "v"(r[0]), "v"(r[1]));
                asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pub[1][q][1]) : "v"(r[2]), "v"(r[3]));
                *reinterpret_cast<uint2 *>(&L.hpub[0][pair][4 * q][lane]) = make_uint2(pub[0][q][0], pub[0][q][1]);
                *reinterpret_cast<uint2 *>(&L.hpub[1][pair][4 * q][lane]) = make_uint2(pub[1][q][0], pub[1][q][1]);
            }
        }
        __syncthreads();                                    // h_t published: the tile-step barrier
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 16; ++i) r += h[i];
    if (r == 12345.0f) sink[0] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int ROLES>
static void run(const char *name, const uint4 *w, const uint4 *h, float *sink, uint64_t *cyc, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 4, threads = ROLES == 2 ? 512 : 256;
    hipLaunchKernelGGL((bench<ROLES>), dim3(blocks), dim3(threads), 0, 0, w, h, sink, iters, cyc);
    hipDeviceSynchronize();
    float sum = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((bench<ROLES>), dim3(blocks), dim3(threads), 0, 0, w, h, sink, iters, cyc);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        sum += ms;
    }
    std::vector<uint64_t> hc(blocks * 8);
    hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost);
    double cs = 0; int nc = 0;
    for (int i = 0; i < blocks; ++i) { cs += (double)hc[i * 8]; ++nc; }
    const double per_step_us = sum / 5 * 1e3 / ((double)iters * (blocks / 256));
    printf("%-28s %8.3f ms   %6.3f us per tile-step and CU   %7.1f memtime ticks per tile-step\n", name, sum / 5, per_step_us, cs / nc / iters);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    std::vector<_Float16> w((size_t)4 * 36 * 64 * 8), h((size_t)2 * 4 * 2 * 64 * 8);
    srand(1);
    for (auto &v : w) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.2f);
    for (auto &v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
    uint4 *dw, *dh; float *sink; uint64_t *cyc;
    hipMalloc(&dw, w.size() * 2); hipMalloc(&dh, h.size() * 2); hipMalloc(&sink, 4); hipMalloc(&cyc, 1024 * 8 * 8);
    hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dh, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int r = 0; r < 2; ++r) {
        run<1>("one wave per SIMD, all of it", dw, dh, sink, cyc, iters);
        run<2>("matrix wave + gate wave", dw, dh, sink, cyc, iters);
    }
    return 0;
}

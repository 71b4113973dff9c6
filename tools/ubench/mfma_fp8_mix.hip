// VERDICT r02 item 5, the time half: the MFMA stream of gru_split2_kernel's tile-step (one wave per SIMD, 32 units x 32 rows x 3 gates,
// K = 128, weights resident, hidden fragments re-read from LDS, random operands) as shipped -- three passes of v_mfma_f32_16x16x32_f16,
// 144 per tile-step -- against the form with the two low-order passes on block-scaled fp8: 48 fp16 MFMAs + 24
// v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3, one MFMA per 16x16 sub-tile and pass: K = 128 in one instruction).  Bare streams: no gate
// math, no publish -- an upper bound of what the change could buy the kernel (which is bound by the issue port and by power as much as
// by the pipe: DESIGN.md 3.1).   usage: mfma_fp8_mix [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int FP8>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) bench(const uint4 *w, const uint4 *hinit, float *sink, int iters)
{
    __shared__ uint4 lds[2][4][2][64];                    // [hi|lo][k-step of 32][row half][lane] fp16 fragments
    __shared__ uint4 lds8[2][2][2][64];                   // [h_hi|h_lo as fp8][row half][16-byte half of the 32-byte fragment][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 4 * 2 * 64; i += 256) (&lds[0][0][0][0])[i] = hinit[i];
    for (int i = threadIdx.x; i < 2 * 2 * 2 * 64; i += 256) (&lds8[0][0][0][0])[i] = hinit[i];
    half8 Whi[3][4][2], Wlo[3][4][2];                     // fp16 fragments: [gate][k-step][unit half]
    i32x8 W8hi[3][2], W8lo[3][2];                         // fp8 fragments, K = 128: [gate][unit half]
    const uint4 *wp = w + (size_t)wave * 96 * 64 + lane;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int uh = 0; uh < 2; ++uh) {
                Whi[g][k][uh] = __builtin_bit_cast(half8, wp[(size_t)((g * 4 + k) * 2 + uh) * 64]);
                Wlo[g][k][uh] = __builtin_bit_cast(half8, wp[(size_t)(24 + (g * 4 + k) * 2 + uh) * 64]);
            }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int uh = 0; uh < 2; ++uh) {
            const uint4 a = wp[(size_t)(48 + (g * 2 + uh) * 2) * 64], b = wp[(size_t)(48 + (g * 2 + uh) * 2 + 1) * 64];
            const uint4 c = wp[(size_t)(72 + (g * 2 + uh) * 2) * 64], d = wp[(size_t)(72 + (g * 2 + uh) * 2 + 1) * 64];
            W8hi[g][uh] = i32x8{ (int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, (int)b.z, (int)b.w };
            W8lo[g][uh] = i32x8{ (int)c.x, (int)c.y, (int)c.z, (int)c.w, (int)d.x, (int)d.y, (int)d.z, (int)d.w };
        }
    __syncthreads();
    f32x4 c[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) c[g][q] = f32x4{ 0, 0, 0, 0 };
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            half8 hf[2], lf[2];
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                hf[rh] = __builtin_bit_cast(half8, lds[0][k][rh][lane]);
                lf[rh] = __builtin_bit_cast(half8, lds[1][k][rh][lane]);
            }
#pragma unroll
            for (int pass = 0; pass < (FP8 ? 1 : 3); ++pass)
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int uh = 0; uh < 2; ++uh)
#pragma unroll
                        for (int rh = 0; rh < 2; ++rh)
                            c[g][2 * uh + rh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pass == 2 ? Wlo[g][k][uh] : Whi[g][k][uh], pass == 1 ? lf[rh] : hf[rh],
                                                                                       c[g][2 * uh + rh], 0, 0, 0);
        }
        if (FP8) {
            i32x8 h8[2][2];                               // [h_hi | h_lo][row half]: 32 fp8 values per lane
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
                    const uint4 a = lds8[p][rh][0][lane], b = lds8[p][rh][1][lane];
                    h8[p][rh] = i32x8{ (int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, (int)b.z, (int)b.w };
                }
#pragma unroll
            for (int pass = 1; pass < 3; ++pass)
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int uh = 0; uh < 2; ++uh)
#pragma unroll
                        for (int rh = 0; rh < 2; ++rh)          // scales: e8m0 127 = 2^0 in every byte
                            c[g][2 * uh + rh] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pass == 2 ? W8lo[g][uh] : W8hi[g][uh], h8[pass == 1 ? 1 : 0][rh],
                                                                                                 c[g][2 * uh + rh], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
        if ((it & 63) == 63)
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) c[g][q] *= 1e-3f;
    }
    float r = 0;
    for (int g = 0; g < 3; ++g)
        for (int q = 0; q < 4; ++q) r += c[g][q][0] + c[g][q][1] + c[g][q][2] + c[g][q][3];
    if (r == 12345.0f) sink[0] = r;
}

template <int FP8>
static void run(const char *name, const uint4 *w, const uint4 *h, float *sink, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 4;
    hipLaunchKernelGGL((bench<FP8>), dim3(blocks), dim3(256), 0, 0, w, h, sink, iters);
    hipDeviceSynchronize();
    float sum = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((bench<FP8>), dim3(blocks), dim3(256), 0, 0, w, h, sink, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        sum += ms;
    }
    // per tile-step: 32 units x 32 rows x 3 gates x K 128 x 3 passes
    const double flops = 2.0 * 3 * 32 * 32 * 3 * 128 * (double)iters * blocks * 4;
    printf("%-44s %8.3f ms   %7.1f TFLOP/s issued   %6.1f ns per tile-step\n", name, sum / 5, flops / (sum / 5) / 1e9, sum / 5 * 1e6 / ((double)iters * 4));
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    std::vector<uint8_t> w((size_t)4 * 96 * 64 * 16), h((size_t)2 * 4 * 2 * 64 * 16);
    srand(1);
    // fp16 halves with small exponents double as well-formed e4m3 bytes: random bit patterns, NaN codes (0x7f / 0xff) avoided
    for (size_t i = 0; i < w.size(); i += 2) { _Float16 v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.2f); memcpy(&w[i], &v, 2); }
    for (size_t i = 0; i < h.size(); i += 2) { _Float16 v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f); memcpy(&h[i], &v, 2); }
    for (auto &b : w) if ((b & 0x7f) == 0x7f) b ^= 1;
    for (auto &b : h) if ((b & 0x7f) == 0x7f) b ^= 1;
    uint4 *dw, *dh; float *sink;
    hipMalloc(&dw, w.size()); hipMalloc(&dh, h.size()); hipMalloc(&sink, 4);
    hipMemcpy(dw, w.data(), w.size(), hipMemcpyHostToDevice); hipMemcpy(dh, h.data(), h.size(), hipMemcpyHostToDevice);
    for (int r = 0; r < 2; ++r) {
        run<0>("three fp16 passes (144 x 16x16x32)", dw, dh, sink, iters);
        run<1>("fp16 pass + 2 e4m3 passes (48 + 24 x 16x16x128)", dw, dh, sink, iters);
    }
    return 0;
}

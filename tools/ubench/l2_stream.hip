// What one CU's vector-memory path delivers when its waves re-read an L2-resident buffer into registers with 16-byte loads per lane
// (the fragment stream of gru_stream64_kernel / gru_stream64x2_kernel: every wave 192 KB per step out of a 768 KB weight set that all
// CUs share): bytes per clock and CU, with 4 and with 8 waves per CU, 6 or 12 loads in flight per wave, alone and next to LDS traffic of
// the other waves.   usage: l2_stream [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1))) *gp;

template <int INFLIGHT, int LDSMIX>
__global__ void __launch_bounds__(512) bench(const uint4 *buf, unsigned *sink, int iters, int frags_per_wave, unsigned long long *clk)
{
    __shared__ u32x4 lds[2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = u32x4{ (unsigned)i, 1, 2, 3 };
    __syncthreads();
    u32x4 acc = { 0, 0, 0, 0 };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (LDSMIX && (wave & 1)) {
        // the other half of the waves reads LDS the whole time (what a gate-phase wave does)
        for (int it = 0; it < iters * frags_per_wave / 4; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= lds[(lane + 64 * j + it) & 2047];
        }
    } else {
        gp base = (gp)(buf) + (size_t)(wave % 4) * frags_per_wave * 64 + lane;
        for (int it = 0; it < iters; ++it) {
            for (int f = 0; f < frags_per_wave; f += INFLIGHT) {
                u32x4 q[INFLIGHT];
#pragma unroll
                for (int j = 0; j < INFLIGHT; ++j) q[j] = base[(size_t)(f + j) * 64];
#pragma unroll
                for (int j = 0; j < INFLIGHT; ++j) acc ^= q[j];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc[0] == 0x12345678u) sink[threadIdx.x] = acc[1] ^ acc[2] ^ acc[3];
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const int frags = 192;                                     // 192 KB per wave and pass, 768 KB for the four streaming waves
    uint4 *buf; unsigned *sink; unsigned long long *clk;
    hipMalloc(&buf, (size_t)4 * frags * 1024); hipMemset(buf, 1, (size_t)4 * frags * 1024);
    hipMalloc(&sink, 4096); hipMalloc(&clk, 256 * 8 * 8);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    auto run = [&](const char *what, auto kern, int threads, int streaming_waves) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, buf, sink, 2, frags, clk);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, buf, sink, iters, frags, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)cus * streaming_waves * frags * 1024.0 * iters;
        const double clk_mhz = prop.clockRate / 1e3;
        printf("%-64s %8.3f ms  %7.1f GB/s per CU  %6.1f B per clock and CU (at %.0f MHz)\n", what, ms, bytes / ms / 1e6 / cus,
               bytes / cus / (ms * 1e-3 * clk_mhz * 1e6), clk_mhz);
    };
    run("4 waves per CU, 6 loads in flight each", bench<6, 0>, 256, 4);
    run("4 waves per CU, 12 loads in flight each", bench<12, 0>, 256, 4);
    run("4 waves per CU, 24 loads in flight each", bench<24, 0>, 256, 4);
    run("8 waves per CU, 12 loads in flight each", bench<12, 0>, 512, 8);
    run("8 waves per CU: 4 stream (12 in flight), 4 read LDS", bench<12, 1>, 512, 4);
    return 0;
}

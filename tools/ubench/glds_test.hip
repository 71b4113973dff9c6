// Semantics probe for __builtin_amdgcn_global_load_lds (LDS-DMA) inside ONE wave: a ring of 1-KiB pieces written by
// the DMA, retired with a counted s_waitcnt vmcnt(N), read back by the same wave with ds_read_b128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ void __launch_bounds__(256) probe(const uint4 *__restrict__ src, int npieces, unsigned long long *__restrict__ bad, int use_barrier)
{
    __shared__ __attribute__((aligned(1024))) unsigned char ring[4][DEPTH][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4 *mine = src + ((size_t)blockIdx.x * 4 + wave) * npieces * 64 + lane;
    unsigned long long errs = 0;
    // prologue: DEPTH pieces in flight
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        __builtin_amdgcn_global_load_lds(mine + (size_t)i * 64, (__attribute__((address_space(3))) void *)&ring[wave][i][0], 16, 0, 0);
    for (int k = 0; k < npieces; ++k) {
        // piece k is the oldest outstanding one: at most DEPTH-1 younger ones may stay in flight
        if (DEPTH == 1) __builtin_amdgcn_s_waitcnt(0x0f70 | 0);            // vmcnt(0), lgkm/exp untouched
        else if (DEPTH == 2) __builtin_amdgcn_s_waitcnt(0x0f70 | 1);
        else if (DEPTH == 4) __builtin_amdgcn_s_waitcnt(0x0f70 | 3);
        else __builtin_amdgcn_s_waitcnt(0x0f70 | 7);
        if (use_barrier) __builtin_amdgcn_s_barrier();
        // read through inline asm: the compiler then does not know about the DMA -> LDS -> read dependency and cannot
        // add its own vmcnt(0); only the counted wait above orders the read
        u32x4 gotv;
        const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)&ring[wave][k % DEPTH][lane * 16];
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(gotv) : "v"(addr) : "memory");
        const uint4 got = make_uint4(gotv[0], gotv[1], gotv[2], gotv[3]);
        // expected value from the index (an ordinary global load here would make hipcc drain the ring with vmcnt(0))
        const unsigned long long gi = ((size_t)blockIdx.x * 4 + wave) * npieces * 64 + lane + (size_t)k * 64;
        const uint4 want = make_uint4((unsigned)gi * 2654435761u, (unsigned)(gi >> 3), (unsigned)gi ^ 0x5a5a5a5au, (unsigned)(gi * 7));
        if (got.x != want.x || got.y != want.y || got.z != want.z || got.w != want.w) ++errs;
        // the slot is free again (our own ds_read must have returned: wait for it before the DMA overwrites)
        __builtin_amdgcn_s_waitcnt(0xc07f);                                // lgkmcnt(0)
        if (k + DEPTH < npieces)
            __builtin_amdgcn_global_load_lds(mine + (size_t)(k + DEPTH) * 64, (__attribute__((address_space(3))) void *)&ring[wave][k % DEPTH][0], 16, 0, 0);
        else
            asm volatile("" ::: "memory");
    }
    for (int o = 32; o > 0; o >>= 1) errs += __shfl_xor(errs, o);
    if (lane == 0 && errs) atomicAdd(bad, errs);
}

int main()
{
    const int blocks = 1024, npieces = 64;
    const size_t n = (size_t)blocks * 4 * npieces * 64;
    std::vector<uint4> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = make_uint4((unsigned)i * 2654435761u, (unsigned)(i >> 3), (unsigned)i ^ 0x5a5a5a5au, (unsigned)(i * 7));
    uint4 *d; unsigned long long *bad;
    (void)hipMalloc(&d, n * 16); (void)hipMalloc(&bad, 8);
    (void)hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    for (int bar = 0; bar < 2; ++bar) {
        unsigned long long r[4] = { 0, 0, 0, 0 };
        (void)hipMemset(bad, 0, 8); probe<1><<<blocks, 256>>>(d, npieces, bad, bar); (void)hipMemcpy(&r[0], bad, 8, hipMemcpyDeviceToHost);
        (void)hipMemset(bad, 0, 8); probe<2><<<blocks, 256>>>(d, npieces, bad, bar); (void)hipMemcpy(&r[1], bad, 8, hipMemcpyDeviceToHost);
        (void)hipMemset(bad, 0, 8); probe<4><<<blocks, 256>>>(d, npieces, bad, bar); (void)hipMemcpy(&r[2], bad, 8, hipMemcpyDeviceToHost);
        (void)hipMemset(bad, 0, 8); probe<8><<<blocks, 256>>>(d, npieces, bad, bar); (void)hipMemcpy(&r[3], bad, 8, hipMemcpyDeviceToHost);
        printf("barrier=%d: mismatching lanes with ring depth 1/2/4/8: %llu %llu %llu %llu (of %zu)\n", bar, r[0], r[1], r[2], r[3], n);
    }
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}

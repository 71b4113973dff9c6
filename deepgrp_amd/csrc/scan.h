// Device-wide exclusive scan of packed uint64 counters and the launch-grid helper, shared by the FASTA ingest, the MSS scan and the
// segment extraction (each translation unit gets its own copy of these small kernels: internal linkage).
#pragma once
#include "dgrp_common.h"

namespace {

static inline int grid_for(int64_t work_items, int block, int max_blocks = 256 * 8)
{
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

// ------------------------------------------------------------------------------------------
// Device-wide exclusive scan of uint64 (two packed 32-bit counters never overflow into each
// other for n < 2^31).  Three launches: tile sums, single-workgroup scan of the sums, apply.
// ------------------------------------------------------------------------------------------
#define SCAN_TILE 2048   // elements per workgroup (256 threads x 8)

__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t *total, uint64_t *lds)
{
    // 256 threads; returns the exclusive prefix of v within the workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint64_t y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    uint64_t base = 0;
    for (int w = 0; w < wave; ++w) base += lds[w];
    if (total) *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + x - v;
}

__global__ void __launch_bounds__(256) scan_tilesum_kernel(const uint64_t *__restrict__ in, int64_t n, uint64_t *__restrict__ tilesum)
{
    __shared__ uint64_t lds[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    uint64_t s = 0;
    for (int j = 0; j < 8; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i < n) s += in[i];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tilesum[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ void __launch_bounds__(256) scan_sums_kernel(uint64_t *__restrict__ tilesum, int64_t ntiles, uint64_t *__restrict__ grand)
{
    __shared__ uint64_t lds[4];
    uint64_t carry = 0;
    for (int64_t base = 0; base < ntiles; base += 256) {
        const int64_t i = base + threadIdx.x;
        const uint64_t v = i < ntiles ? tilesum[i] : 0;
        uint64_t tot;
        const uint64_t ex = block_exclusive_scan(v, &tot, lds);
        if (i < ntiles) tilesum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && grand) *grand = carry;
}

__global__ void __launch_bounds__(256) scan_apply_kernel(const uint64_t *__restrict__ in, int64_t n,
                                                         const uint64_t *__restrict__ tilesum, uint64_t *__restrict__ out)
{
    __shared__ uint64_t lds[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * 8;
    uint64_t v[8], s = 0;
    for (int j = 0; j < 8; ++j) { v[j] = base + j < n ? in[base + j] : 0; s += v[j]; }
    uint64_t ex = block_exclusive_scan(s, nullptr, lds) + tilesum[blockIdx.x];
    for (int j = 0; j < 8; ++j) {
        if (base + j < n) out[base + j] = ex;
        ex += v[j];
    }
}

// in/out may alias; tiles: workspace of ceil(n / SCAN_TILE) uint64; grand: optional device uint64 total
static int device_exclusive_scan(const uint64_t *in, uint64_t *out, int64_t n, uint64_t *tiles, uint64_t *grand,
                                 hipStream_t stream)
{
    if (n <= 0) {
        if (grand) DGRP_HIP(hipMemsetAsync(grand, 0, sizeof(uint64_t), stream));
        return DGRP_OK;
    }
    const int64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(scan_tilesum_kernel, dim3((unsigned)ntiles), dim3(256), 0, stream, in, n, tiles);
    DGRP_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, stream, tiles, ntiles, grand);
    DGRP_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)ntiles), dim3(256), 0, stream, in, n, tiles, out);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

}   // namespace

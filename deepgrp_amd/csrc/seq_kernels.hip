// Sequence-side kernels: byte -> class index / one-hot (A2), window materialisation (A3),
// standalone overlap max-merge (A6).  All HBM-bound byte/word streaming: coalesced 16-byte
// accesses, LDS only where it turns a scattered write pattern into full lines.
#include "dgrp_common.h"

#include <stdlib.h>

// ------------------------------------------------------------------------------------------
// A2  deepgrp/sequence.pyx:11-17 (table) and :33-35 (loop)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t class_of(uint32_t c)
{
    uint32_t l = c | 0x20u;                       // only 'A'/'a' map onto 'a' etc.
    return l == 'a' ? 0u : l == 'c' ? 1u : l == 'g' ? 2u : l == 't' ? 3u : 4u;
}

__device__ __forceinline__ uint32_t class_of4(uint32_t w)
{
    return class_of(w & 0xff) | (class_of((w >> 8) & 0xff) << 8) | (class_of((w >> 16) & 0xff) << 16) |
           (class_of(w >> 24) << 24);
}

// 16 bytes per thread per iteration; the unaligned head/tail is done bytewise.
__global__ void __launch_bounds__(256) encode_kernel(const uint8_t *__restrict__ seq, int64_t n,
                                                     uint8_t *__restrict__ idx)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    // both pointers come from the same offset into 256-byte aligned allocations in practice, but
    // do not rely on it: vectorise only when both are 16-byte aligned
    const bool aligned = ((((uintptr_t)seq) | ((uintptr_t)idx)) & 15) == 0;
    int64_t nvec = aligned ? n / 16 : 0;
    const uint4 *s4 = reinterpret_cast<const uint4 *>(seq);
    uint4 *o4 = reinterpret_cast<uint4 *>(idx);
    for (int64_t i = tid; i < nvec; i += nthreads) {
        uint4 v = s4[i];
        uint4 r;
        r.x = class_of4(v.x); r.y = class_of4(v.y); r.z = class_of4(v.z); r.w = class_of4(v.w);
        o4[i] = r;
    }
    for (int64_t i = nvec * 16 + tid; i < n; i += nthreads) idx[i] = (uint8_t)class_of(seq[i]);
}

// int8 [5, n] C-order: row c holds (class == c).  Each thread turns 16 input bytes into five
// 16-byte row stores.
__global__ void __launch_bounds__(256) onehot_kernel(const uint8_t *__restrict__ seq, int64_t n,
                                                     int8_t *__restrict__ out)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    const bool aligned = ((((uintptr_t)seq) | ((uintptr_t)out)) & 15) == 0 && (n % 16) == 0;
    int64_t nvec = aligned ? n / 16 : 0;
    const uint4 *s4 = reinterpret_cast<const uint4 *>(seq);
    for (int64_t i = tid; i < nvec; i += nthreads) {
        uint4 v = s4[i];
        uint32_t k[4] = { class_of4(v.x), class_of4(v.y), class_of4(v.z), class_of4(v.w) };
#pragma unroll
        for (uint32_t c = 0; c < 5; ++c) {
            uint32_t r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t x = k[j] ^ (c * 0x01010101u);        // zero byte <=> class == c
                // exact per-byte zero test (no cross-byte borrow)
                uint32_t nz = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
                r[j] = (nz ^ 0x80808080u) >> 7;
            }
            reinterpret_cast<uint4 *>(out + (int64_t)c * n)[i] = make_uint4(r[0], r[1], r[2], r[3]);
        }
    }
    for (int64_t i = nvec * 16 + tid; i < n; i += nthreads) {
        uint32_t k = class_of(seq[i]);
        for (uint32_t c = 0; c < 5; ++c) out[(int64_t)c * n + i] = (int8_t)(k == c);
    }
}

// ------------------------------------------------------------------------------------------
// A3  deepgrp/prediction.py:30-32 : window w is rows [w*s, w*s+T) of the transposed one-hot,
// cast to float.  One workgroup builds WB whole windows in LDS (each (w,t) writes its five
// values) and streams the image out with 16-byte stores, so HBM sees only full lines.
// ------------------------------------------------------------------------------------------
template <typename elem_t>
__global__ void __launch_bounds__(256) windows_kernel(const uint8_t *__restrict__ idx, int64_t T, int64_t s,
                                                      int64_t w0, int64_t nw, int WB, elem_t *__restrict__ out,
                                                      elem_t one)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    elem_t *img = reinterpret_cast<elem_t *>(smem);
    const int64_t wb = (int64_t)blockIdx.x * WB;                // first window of this group (relative)
    const int nwl = (int)min((int64_t)WB, nw - wb);
    const int64_t cells = (int64_t)nwl * T;
    // a thread builds 8 consecutive (window, position) cells = 40 values = 80 / 160 contiguous bytes
    // and stores them with 16-byte LDS writes (stride 80 / 160 B between lanes: conflict-free)
    constexpr int VEC = 16 / (int)sizeof(elem_t), NV = 40 / VEC;
    const int64_t ngroups = cells / 8;
    for (int64_t g = threadIdx.x; g < ngroups; g += blockDim.x) {
        const int64_t c0 = g * 8;
        int64_t wl = c0 / T, t = c0 - wl * T;
        uint32_t b[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            b[k] = idx[(w0 + wb + wl) * s + t];
            if (++t == T) { t = 0; ++wl; }
        }
        elem_t v[40];
#pragma unroll
        for (int e = 0; e < 40; ++e) v[e] = (b[e / 5] == (uint32_t)(e % 5)) ? one : (elem_t)0;
        uint4 *dst = reinterpret_cast<uint4 *>(img + c0 * 5);
#pragma unroll
        for (int q = 0; q < NV; ++q) dst[q] = *reinterpret_cast<const uint4 *>(&v[q * VEC]);
    }
    for (int64_t i = ngroups * 8 + threadIdx.x; i < cells; i += blockDim.x) {      // < 8 leftover cells
        const int64_t wl = i / T, t = i - wl * T;
        const uint32_t b = idx[(w0 + wb + wl) * s + t];
        elem_t *p = img + i * 5;
#pragma unroll
        for (uint32_t c = 0; c < 5; ++c) p[c] = (b == c) ? one : (elem_t)0;
    }
    __syncthreads();
    const int64_t bytes = cells * 5 * (int64_t)sizeof(elem_t);
    unsigned char *dst = reinterpret_cast<unsigned char *>(out + wb * T * 5);
    // group start is 16-byte aligned when WB*T*5*sizeof(elem) % 16 == 0 (the host picks WB so)
    const int64_t nvec = ((((uintptr_t)dst) & 15) == 0) ? bytes / 16 : 0;
    for (int64_t i = threadIdx.x; i < nvec; i += blockDim.x)
        reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(smem)[i];
    for (int64_t i = nvec * 16 + threadIdx.x; i < bytes; i += blockDim.x) dst[i] = smem[i];
}

// ------------------------------------------------------------------------------------------
// A6  deepgrp/maxcalc.c:10-24, gather form: every output row takes the max over the (<= dim0 /
// stride + 1) windows covering it, so each input element is read once, each output element
// read and written once, and there is no write conflict between windows.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) get_max_kernel(float *__restrict__ out, int64_t out_rows,
                                                      const float *__restrict__ in, int64_t dim0, int64_t dim1,
                                                      int64_t stride, int64_t batch)
{
    const int64_t total_rows = min(out_rows, (batch - 1) * stride + dim0);
    const int64_t total = total_rows * dim1;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += nthreads) {
        const int64_t p = f / dim1, c = f - p * dim1;
        int64_t bhi = p / stride;                              // last window starting at or before p
        if (bhi > batch - 1) bhi = batch - 1;
        float v = out[f];
        for (int64_t b = bhi; b >= 0; --b) {
            const int64_t t = p - b * stride;
            if (t >= dim0) break;
            const float x = in[(b * dim0 + t) * dim1 + c];
            v = x > v ? x : v;                                 // MAX(out, in) = out > in ? out : in
        }
        out[f] = v;
    }
}

static inline int grid_for(int64_t work_items, int block, int max_blocks = 256 * 8)
{
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

DGRP_EXPORT int dgrp_encode(const uint8_t *d_seq, int64_t n, uint8_t *d_idx, void *stream)
{
    DGRP_REQUIRE(n >= 0 && (n == 0 || (d_seq && d_idx)), "dgrp_encode: bad arguments");
    if (n == 0) return DGRP_OK;
    hipLaunchKernelGGL(encode_kernel, dim3(grid_for((n + 15) / 16, 256)), dim3(256), 0, (hipStream_t)stream,
                       d_seq, n, d_idx);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

DGRP_EXPORT int dgrp_onehot(const uint8_t *d_seq, int64_t n, int8_t *d_onehot, void *stream)
{
    DGRP_REQUIRE(n >= 0 && (n == 0 || (d_seq && d_onehot)), "dgrp_onehot: bad arguments");
    if (n == 0) return DGRP_OK;
    hipLaunchKernelGGL(onehot_kernel, dim3(grid_for((n + 15) / 16, 256)), dim3(256), 0, (hipStream_t)stream,
                       d_seq, n, d_onehot);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

DGRP_EXPORT int64_t dgrp_window_count(int64_t n, int64_t T, int64_t s)
{
    if (T <= 0 || s <= 0 || n - T <= 0) return 0;
    return (n - T + s - 1) / s;
}

DGRP_EXPORT int dgrp_windows_onehot(const uint8_t *d_idx, int64_t n, int64_t T, int64_t s, int64_t w0,
                                    int64_t nw, int elem, void *d_out, void *stream)
{
    DGRP_REQUIRE(elem == 2 || elem == 4, "dgrp_windows_onehot: elem must be 2 (fp16) or 4 (fp32)");
    DGRP_REQUIRE(T > 0 && s > 0 && w0 >= 0 && nw >= 0, "dgrp_windows_onehot: bad T/s/w0/nw");
    if (nw == 0) return DGRP_OK;
    DGRP_REQUIRE(d_idx && d_out, "dgrp_windows_onehot: NULL pointer");
    DGRP_REQUIRE((w0 + nw - 1) * s + T <= n, "dgrp_windows_onehot: window %lld runs past n=%lld",
                 (long long)(w0 + nw - 1), (long long)n);
    // windows per group: a multiple of 8 keeps every group start 16-byte aligned for both element
    // sizes (8*T*5*2 = 80 T)
    int WB = 8;
    // ~16 KiB tiles: enough workgroups per CU to keep the stores streaming (measured 4.9 TB/s fp16, 6.3 TB/s fp32)
    const int64_t lds_target = getenv("DGRP_WIN_LDS_KB") ? atoi(getenv("DGRP_WIN_LDS_KB")) * 1024 : 16 * 1024;
    while ((int64_t)WB * 2 * T * 5 * elem <= lds_target && WB < 64) WB *= 2;
    DGRP_REQUIRE((int64_t)WB * T * 5 * elem <= 160 * 1024, "dgrp_windows_onehot: T=%lld too large", (long long)T);
    const size_t lds = (size_t)WB * T * 5 * elem;
    const int64_t groups = (nw + WB - 1) / WB;
    DGRP_REQUIRE(groups < (1ll << 31), "dgrp_windows_onehot: too many windows in one call");
    if (elem == 2) {
        DGRP_HIP(hipFuncSetAttribute((const void *)windows_kernel<_Float16>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(windows_kernel<_Float16>, dim3((unsigned)groups), dim3(256), lds, (hipStream_t)stream,
                           d_idx, T, s, w0, nw, WB, (_Float16 *)d_out, (_Float16)1.0f);
    } else {
        DGRP_HIP(hipFuncSetAttribute((const void *)windows_kernel<float>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(windows_kernel<float>, dim3((unsigned)groups), dim3(256), lds, (hipStream_t)stream,
                           d_idx, T, s, w0, nw, WB, (float *)d_out, 1.0f);
    }
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

DGRP_EXPORT int dgrp_get_max(float *d_output, int64_t out_rows, const float *d_inputs, int64_t dim0,
                             int64_t dim1, int64_t stride, int64_t batchsize, void *stream)
{
    DGRP_REQUIRE(dim0 > 0 && dim1 > 0 && stride > 0 && batchsize >= 0 && out_rows >= 0, "dgrp_get_max: bad shape");
    if (batchsize == 0 || out_rows == 0) return DGRP_OK;
    DGRP_REQUIRE(d_output && d_inputs, "dgrp_get_max: NULL pointer");
    const int64_t rows = min(out_rows, (batchsize - 1) * stride + dim0);
    hipLaunchKernelGGL(get_max_kernel, dim3(grid_for(rows * dim1, 256, 256 * 16)), dim3(256), 0,
                       (hipStream_t)stream, d_output, out_rows, d_inputs, dim0, dim1, stride, batchsize);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

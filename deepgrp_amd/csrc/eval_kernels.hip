// N2: the evaluation helpers next to the prediction path, on per-base label arrays that are already in
// HBM -- deepgrp.prediction.confusion_matrix (deepgrp/prediction.py:204-222) and filter_segments (:244-260).
// Both are single-pass, HBM-bound byte scans (2 B/bp and 1-2 B/bp).
#include "dgrp_common.h"

// cnf[t][p] += 1 for every base: a 16 x 16 (beyond 16 classes: 64 x 64) histogram per workgroup in LDS (wave-private copies would not pay: at
// most a few distinct cells are hot, ds_add_u32 serialises those whatever the layout), flushed with one 64-bit
// atomic per non-zero cell.
__global__ void __launch_bounds__(256) confusion_kernel(const int8_t *__restrict__ truth, const int8_t *__restrict__ pred,
                                                        int64_t n, int ncls, unsigned long long *__restrict__ cnf,
                                                        int *__restrict__ bad)
{
    __shared__ unsigned hist[DGRP_MAXC * DGRP_MAXC];
    const int pitch = ncls <= 16 ? 16 : DGRP_MAXC;
    for (int i = threadIdx.x; i < pitch * pitch; i += 256) hist[i] = 0u;
    __syncthreads();
    const int64_t per = 16 * 256;                                     // bases per workgroup iteration (16 B per lane)
    for (int64_t base = (int64_t)blockIdx.x * per; base < n; base += (int64_t)gridDim.x * per) {
        const int64_t i0 = base + (int64_t)threadIdx.x * 16;
        if (i0 + 16 <= n) {
            const uint4 tv = *reinterpret_cast<const uint4 *>(truth + i0), pv = *reinterpret_cast<const uint4 *>(pred + i0);
            const uint32_t tw[4] = { tv.x, tv.y, tv.z, tv.w }, pw[4] = { pv.x, pv.y, pv.z, pv.w };
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int t = (int8_t)(tw[k >> 2] >> (8 * (k & 3))), q = (int8_t)(pw[k >> 2] >> (8 * (k & 3)));
                if ((unsigned)t < (unsigned)ncls && (unsigned)q < (unsigned)ncls) atomicAdd(&hist[t * pitch + q], 1u);
                else *bad = 1;
            }
        } else {
            for (int64_t i = i0; i < n && i < i0 + 16; ++i) {
                const int t = truth[i], q = pred[i];
                if ((unsigned)t < (unsigned)ncls && (unsigned)q < (unsigned)ncls) atomicAdd(&hist[t * pitch + q], 1u);
                else *bad = 1;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < pitch * pitch; i += 256) {
        const unsigned v = hist[i];
        const int t = i / pitch, q = i % pitch;
        if (v != 0u && t < ncls && q < ncls) atomicAdd(&cnf[t * ncls + q], (unsigned long long)v);
    }
}

DGRP_EXPORT int dgrp_confusion_matrix(const int8_t *d_true, const int8_t *d_pred, int64_t n, int ncls, int64_t *d_cnf,
                                      int *d_bad, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    DGRP_REQUIRE(n >= 0 && ncls >= 1 && ncls <= DGRP_MAXC && d_cnf && d_bad, "dgrp_confusion_matrix: bad arguments (1 <= classes <= 64)");
    DGRP_HIP(hipMemsetAsync(d_cnf, 0, sizeof(int64_t) * ncls * ncls, stream));
    DGRP_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), stream));
    if (n == 0) return DGRP_OK;
    DGRP_REQUIRE(d_true && d_pred, "dgrp_confusion_matrix: NULL pointer");
    DGRP_REQUIRE(((uintptr_t)d_true & 15) == 0 && ((uintptr_t)d_pred & 15) == 0, "dgrp_confusion_matrix: label arrays must be 16-byte aligned");
    const int64_t groups = (n + 4095) / 4096;
    const unsigned grid = (unsigned)(groups < 2048 ? groups : 2048);
    hipLaunchKernelGGL(confusion_kernel, dim3(grid), dim3(256), 0, stream, d_true, d_pred, n, ncls,
                       reinterpret_cast<unsigned long long *>(d_cnf), d_bad);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

// filter_segments: runs of one positive label shorter than min_len become 0.  A thread that sits on the first base
// of a positive run walks at most min_len bases forward; only runs that are about to be cleared are ever written,
// so the in-place form is race-free in effect: a neighbour that sees a half-cleared short run can only decide
// "start of a short run" for bases that are being cleared anyway, and long runs are never touched.
__global__ void __launch_bounds__(256) filter_segments_kernel(const int8_t *in, int8_t *out,
                                                              int64_t n, int64_t min_len)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int8_t v = in[i];
    if (in != out) {
        // out of place: every base decides for itself from the untouched input
        if (v <= 0) { out[i] = v; return; }
        int64_t lo = i, hi = i + 1;
        while (lo > 0 && i - lo + 1 < min_len && in[lo - 1] == v) --lo;
        while (hi < n && hi - lo < min_len && in[hi] == v) ++hi;
        out[i] = hi - lo < min_len ? (int8_t)0 : v;
        return;
    }
    if (v <= 0 || (i > 0 && in[i - 1] == v)) return;
    int64_t j = i + 1;
    while (j < n && j - i < min_len && in[j] == v) ++j;
    if (j - i < min_len)
        for (int64_t k = i; k < j; ++k) out[k] = 0;
}

DGRP_EXPORT int dgrp_filter_segments(const int8_t *d_labels, int8_t *d_out, int64_t n, int64_t min_len, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    DGRP_REQUIRE(n >= 0, "dgrp_filter_segments: negative length");
    if (n == 0) return DGRP_OK;
    DGRP_REQUIRE(d_labels && d_out, "dgrp_filter_segments: NULL pointer");
    DGRP_REQUIRE(n < (1ll << 39), "dgrp_filter_segments: too long");
    hipLaunchKernelGGL(filter_segments_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_labels, d_out, n, min_len);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

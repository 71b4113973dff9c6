// Post-processing kernels: score transform (A7), softmax path (A8) and run-length segment extraction (A11); the maximal scoring
// segments (A9/A10) live in mss_kernels.hip, FASTA ingest and TSV text in fasta_kernels.hip.  Compiled with
// -ffp-contract=off: the float32/float64 expressions below must round exactly where the
// reference's numpy / C code rounds.
#include "dgrp_common.h"
#include "scan.h"
#include <vector>

#include <math.h>

// ------------------------------------------------------------------------------------------
// numpy's float32 log and exp (the routines np.log / np.exp run on a float32 array on any
// AVX2/AVX512F host: numpy/core/src/umath/loops_exponent_log.dispatch.c.src).  They are not
// correctly rounded, but they are plain fma chains, so they can be reproduced bit for bit --
// which is what makes the scores of deepgrp/prediction.py:55 reproducible at all.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float np_logf(float xin)
{
    // callers pass finite positive normal numbers (m / (1 - m) with m in [1e-6, 0.99])
    const uint32_t bits = __float_as_uint(xin);
    float k = (float)((int)(bits >> 23) - 126);
    float m = __uint_as_float((bits & 0x007fffffu) | 0x3f000000u);          // [0.5, 1)
    if (m <= 0.70710678118654752440f) { m = m + m; k = k - 1.0f; }
    const float x = m - 1.0f;
    float den = fmaf(5.875095403124574342950e-03f, x, 1.546476374983906719538e-01f);
    den = fmaf(den, x, 9.864942958519418960339e-01f);
    den = fmaf(den, x, 2.453006071784736363091e+00f);
    den = fmaf(den, x, 2.612677543073109236779e+00f);
    den = fmaf(den, x, 1.0f);
    float num = fmaf(2.589979117907922693523e-02f, x, 3.808837741388407920751e-01f);
    num = fmaf(num, x, 1.480000633576506585156e+00f);
    num = fmaf(num, x, 2.112677543073053063722e+00f);
    num = fmaf(num, x, 9.999999999999998702752e-01f);
    num = fmaf(num, x, 0.0f);
    return fmaf(k, 0.693147180559945309417232121458176568f, __fdiv_rn(num, den));
}

__device__ __forceinline__ float np_expf(float x)
{
    if (x != x) return x;
    if (x >= 88.72283905206835f) return INFINITY;
    if (x <= -103.97208f) return 0.0f;
    const float q = rintf(x * 1.44269504088896340736f);
    float r = fmaf(q, -6.93145752e-1f, x);
    r = fmaf(q, -1.42860677e-6f, r);
    float num = fmaf(5.082762527590693718096e-04f, r, 6.757896990527504603057e-03f);
    num = fmaf(num, r, 5.114512081637298353406e-02f);
    num = fmaf(num, r, 2.473615434895520810817e-01f);
    num = fmaf(num, r, 7.257664613233124478488e-01f);
    num = fmaf(num, r, 9.999999999980870924916e-01f);
    float den = fmaf(2.159509375685829852307e-02f, r, -2.742335390411667452936e-01f);
    den = fmaf(den, r, 1.0f);
    return ldexpf(__fdiv_rn(num, den), (int)q);
}

// ------------------------------------------------------------------------------------------
// A7  deepgrp/prediction.py:51-57, all in float32 like numpy, widened at the end
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) scores_kernel(const float *__restrict__ probs, int64_t n, int C,
                                                     double *__restrict__ scores, int8_t *__restrict__ cls)
{
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) {
        const float *p = probs + i * C;
        int a = 0;
        float mx = p[0];
        for (int c = 1; c < C; ++c) {
            const float v = p[c];
            if (v > mx) { mx = v; a = c; }                    // argmax keeps the first maximum
        }
        float m = mx + 1e-6f;                                 // probs.max(axis=1) + 1e-6
        if (m > 0.99f) m = 0.99f;                             // mins[mins > 0.99] = 0.99
        const float t = np_logf(__fdiv_rn(m, 1.0f - m));      // np.log(mins / (1 - mins))
        const float sc = a > 0 ? t : -10.0f * t;              // np.where(cls > 0, t, -10 * t)
        scores[i] = (double)sc;                               // .astype(float)
        cls[i] = (int8_t)a;
    }
}

DGRP_EXPORT int dgrp_scores(const float *d_probs, int64_t n, int C, double *d_scores, int8_t *d_cls, void *stream)
{
    DGRP_REQUIRE(n >= 0 && C >= 1 && C <= DGRP_MAXC, "dgrp_scores: bad n/C");
    if (n == 0) return DGRP_OK;
    DGRP_REQUIRE(d_probs && d_scores && d_cls, "dgrp_scores: NULL pointer");
    hipLaunchKernelGGL(scores_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, d_probs, n,
                       C, d_scores, d_cls);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

// ------------------------------------------------------------------------------------------
// A8  deepgrp/prediction.py:62-65 + deepgrp/__main__.py:83
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) blockmax_kernel(const float *__restrict__ a, int64_t total, float *__restrict__ part)
{
    __shared__ float red[4];
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    float m = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += nthreads) m = fmaxf(m, a[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ void __launch_bounds__(256) softmax_labels_kernel(const float *__restrict__ a, int64_t n, int C,
                                                             const float *__restrict__ part, int nparts,
                                                             float *__restrict__ sm, int8_t *__restrict__ labels)
{
    float gmax = -INFINITY;
    for (int i = 0; i < nparts; ++i) gmax = fmaxf(gmax, part[i]);
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) {
        // (the exponentials twice -- a deterministic routine -- instead of an array of them: any number of classes in registers)
        float sum = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float ec = np_expf(a[i * C + c] - gmax);
            sum = c == 0 ? ec : sum + ec;                     // numpy folds a short row left to right
        }
        int best = 0;
        float bv = __fdiv_rn(np_expf(a[i * C] - gmax), sum);
        if (sm) sm[i * C] = bv;
        for (int c = 1; c < C; ++c) {
            const float v = __fdiv_rn(np_expf(a[i * C + c] - gmax), sum);
            if (sm) sm[i * C + c] = v;
            if (v > bv) { bv = v; best = c; }
        }
        labels[i] = (int8_t)best;
    }
}

DGRP_EXPORT int dgrp_softmax_labels(const float *d_probs, int64_t n, int C, float *d_softmax, int8_t *d_labels,
                                    void *d_work, int64_t work_bytes, void *stream)
{
    DGRP_REQUIRE(n >= 0 && C >= 1 && C <= DGRP_MAXC, "dgrp_softmax_labels: bad n/C");
    if (n == 0) return DGRP_OK;
    const int nparts = 1024;
    DGRP_REQUIRE(d_probs && d_labels && d_work && work_bytes >= (int64_t)(nparts * sizeof(float)),
                 "dgrp_softmax_labels: need %d bytes of workspace", (int)(nparts * sizeof(float)));
    float *part = (float *)d_work;
    hipLaunchKernelGGL(blockmax_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, d_probs, n * C, part);
    DGRP_LAUNCH_CHECK();
    hipLaunchKernelGGL(softmax_labels_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream,
                       d_probs, n, C, part, nparts, d_softmax, d_labels);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

// ------------------------------------------------------------------------------------------
// A11  deepgrp/sequence.pyx:38-53, :79-85 + the label > 0 filter of deepgrp/__main__.py:290.
// get_segments never scans across the last element (length = size - 1), so runs live in
// [0, n-1) and the last element is always a segment of its own.  A run's start and end are
// flagged independently; the k-th start pairs with the k-th end.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool seg_is_start(const int8_t *lab, int64_t i, int64_t n)
{
    const int8_t v = lab[i];
    return v != 0 && (i == 0 || i == n - 1 || lab[i - 1] != v);
}
__device__ __forceinline__ bool seg_is_end(const int8_t *lab, int64_t i, int64_t n)   // i = last element of a segment
{
    const int8_t v = lab[i];
    return v != 0 && (i >= n - 2 || lab[i + 1] != v);
}

// marks (batched records only): bit0 first base of a record, bit1 its last base, bit2 its second-to-last base, bit3 padding
__device__ __forceinline__ uint64_t seg_flags_batch(const int8_t *lab, const uint8_t *marks, int64_t i)
{
    const uint8_t m = marks[i];
    const int8_t v = lab[i];
    if ((m & 8) || v == 0) return 0;
    // sequence.pyx:43-53 per record: the last base is always its own segment
    const bool st = (m & 1) || (m & 2) || lab[i - 1] != v;
    const bool en = (m & 2) || (m & 4) || lab[i + 1] != v;
    return ((uint64_t)st << 32) + (uint64_t)en;
}

// start flag in the upper word, end flag in the lower one; one record (marks == NULL) or many side by side
template <bool BATCH>
__device__ __forceinline__ uint64_t seg_flags(const int8_t *lab, const uint8_t *marks, int64_t i, int64_t n)
{
    if (BATCH) return seg_flags_batch(lab, marks, i);
    return ((uint64_t)seg_is_start(lab, i, n) << 32) + (uint64_t)seg_is_end(lab, i, n);
}

struct seg_records {                  // BATCH: where the records start, what to add to positions, their tags
    const int64_t *start;
    const int64_t *startpos;
    const int32_t *contig;
    int64_t nrec;
};

template <bool BATCH>
__global__ void __launch_bounds__(256) seg_count_kernel(const int8_t *__restrict__ lab, const uint8_t *__restrict__ marks, int64_t n,
                                                        uint64_t *__restrict__ tilecnt)
{
    __shared__ uint64_t lds[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    uint64_t c = 0;
    for (int j = 0; j < 8; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i < n) c += seg_flags<BATCH>(lab, marks, i, n);
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tilecnt[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

template <bool BATCH>
__global__ void __launch_bounds__(256) seg_emit_kernel(const int8_t *__restrict__ lab, const uint8_t *__restrict__ marks, int64_t n,
                                                       int64_t offset, int32_t contig, seg_records recs,
                                                       const uint64_t *__restrict__ tileoff, dgrp_segment *__restrict__ rec, int64_t cap)
{
    __shared__ uint64_t lds[4];
    // consecutive elements per thread so that the order of flags is the order of positions
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * 8;
    uint64_t f[8], s = 0;
    for (int j = 0; j < 8; ++j) {
        const int64_t i = base + j;
        f[j] = i < n ? seg_flags<BATCH>(lab, marks, i, n) : 0;
        s += f[j];
    }
    uint64_t ex = block_exclusive_scan(s, nullptr, lds) + tileoff[blockIdx.x];
    for (int j = 0; j < 8; ++j) {
        const int64_t i = base + j;
        if (f[j]) {
            int64_t off = offset;
            int32_t tag = contig;
            if (BATCH) {
                // the record of position i: last r with start[r] <= i
                int64_t lo = 0, hi = recs.nrec;
                while (hi - lo > 1) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (recs.start[mid] <= i) lo = mid; else hi = mid;
                }
                off = recs.startpos[lo] - recs.start[lo];
                tag = recs.contig[lo];
            }
            if (f[j] >> 32) {
                const int64_t k = (int64_t)(ex >> 32);
                if (k < cap) { rec[k].start = i + off; rec[k].label = lab[i]; rec[k].contig = tag; }
            }
            if (f[j] & 1) {
                const int64_t k = (int64_t)(ex & 0xffffffffull);
                if (k < cap) rec[k].end = i + 1 + off;
            }
        }
        ex += f[j];
    }
}

__global__ void seg_total_kernel(const uint64_t *__restrict__ grand, int64_t *__restrict__ count)
{
    *count = (int64_t)(*grand >> 32);
}

// ---- batched records: padding fix-up of the score transform (marks: see seg_flags_batch) ------------------------
__global__ void __launch_bounds__(64) rec_marks_kernel(const int64_t *__restrict__ start, const int64_t *__restrict__ len,
                                                       int64_t nrec, uint8_t *__restrict__ marks, double *__restrict__ scores,
                                                       int8_t *__restrict__ cls)
{
    const int64_t r = blockIdx.x;
    if (r >= nrec) return;
    const int64_t a = start[r], n = len[r], b = start[r + 1];
    if (threadIdx.x == 0) {
        // (atomicOr on the containing word would be needed if two marks of one record could share a byte: they cannot)
        uint8_t m0 = 1;
        if (n == 1) m0 |= 2;
        if (n == 2) m0 |= 4;
        marks[a] = m0;
        if (n >= 2) marks[a + n - 1] = (uint8_t)(2 | (n == 1 ? 1 : 0));
        if (n >= 3) marks[a + n - 2] = 4;
    }
    for (int64_t i = a + n + threadIdx.x; i < b; i += 64) {
        marks[i] = 8;
        if (scores) scores[i] = 0.0;
        if (cls) cls[i] = 0;
    }
}

// internal (api.hip: dgrp_predict_batch): marks + zeroed padding, then the segments of all records in position order
int dgrp_batch_marks(const int64_t *d_start, const int64_t *d_len, int64_t nrec, int64_t total_n, uint8_t *d_marks,
                     double *d_scores, int8_t *d_cls, hipStream_t stream)
{
    DGRP_HIP(hipMemsetAsync(d_marks, 0, (size_t)total_n, stream));
    hipLaunchKernelGGL(rec_marks_kernel, dim3((unsigned)nrec), dim3(64), 0, stream, d_start, d_len, nrec, d_marks, d_scores, d_cls);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

int dgrp_batch_segments(const int8_t *d_labels, const uint8_t *d_marks, int64_t total_n, const int64_t *d_start, int64_t nrec,
                        const int64_t *d_startpos, const int32_t *d_contig, dgrp_segment *d_records, int64_t cap,
                        int64_t *d_count, void *d_work, int64_t work_bytes, hipStream_t stream)
{
    if (work_bytes < dgrp_segments_workspace_bytes(total_n)) {
        dgrp_set_error("dgrp_batch_segments: workspace too small");
        return DGRP_ENOMEM;
    }
    const int64_t ntiles = (total_n + SCAN_TILE - 1) / SCAN_TILE;
    uint64_t *tiles = (uint64_t *)d_work;
    uint64_t *grand = tiles + ntiles + 1;
    hipLaunchKernelGGL(seg_count_kernel<true>, dim3((unsigned)ntiles), dim3(256), 0, stream, d_labels, d_marks, total_n, tiles);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, stream, tiles, ntiles, grand);
    hipLaunchKernelGGL(seg_emit_kernel<true>, dim3((unsigned)ntiles), dim3(256), 0, stream, d_labels, d_marks, total_n, (int64_t)0,
                       (int32_t)0, seg_records{ d_start, d_startpos, d_contig, nrec }, tiles, d_records, cap);
    hipLaunchKernelGGL(seg_total_kernel, dim3(1), dim3(1), 0, stream, grand, d_count);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

DGRP_EXPORT int64_t dgrp_segments_workspace_bytes(int64_t n)
{
    if (n < 0) return 0;
    return dgrp_align_up(((n + SCAN_TILE - 1) / SCAN_TILE + 2) * 8, 256) + 256;
}

DGRP_EXPORT int dgrp_segments(const int8_t *d_labels, int64_t n, int64_t offset, int32_t contig, dgrp_segment *d_records,
                              int64_t cap, int64_t *d_count, void *d_work, int64_t work_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    DGRP_REQUIRE(n >= 0 && n < (1ll << 31) && cap >= 0 && d_count, "dgrp_segments: bad arguments");
    if (n == 0) {
        DGRP_HIP(hipMemsetAsync(d_count, 0, sizeof(int64_t), stream));
        return DGRP_OK;
    }
    DGRP_REQUIRE(d_labels && d_work && (cap == 0 || d_records), "dgrp_segments: NULL pointer");
    if (work_bytes < dgrp_segments_workspace_bytes(n)) {
        dgrp_set_error("dgrp_segments: workspace too small");
        return DGRP_ENOMEM;
    }
    const int64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    uint64_t *tiles = (uint64_t *)d_work;
    uint64_t *grand = tiles + ntiles + 1;
    hipLaunchKernelGGL(seg_count_kernel<false>, dim3((unsigned)ntiles), dim3(256), 0, stream, d_labels, (const uint8_t *)nullptr, n, tiles);
    DGRP_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, stream, tiles, ntiles, grand);
    DGRP_LAUNCH_CHECK();
    hipLaunchKernelGGL(seg_emit_kernel<false>, dim3((unsigned)ntiles), dim3(256), 0, stream, d_labels, (const uint8_t *)nullptr, n, offset,
                       contig, seg_records{ nullptr, nullptr, nullptr, 0 }, tiles, d_records, cap);
    DGRP_LAUNCH_CHECK();
    hipLaunchKernelGGL(seg_total_kernel, dim3(1), dim3(1), 0, stream, grand, d_count);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

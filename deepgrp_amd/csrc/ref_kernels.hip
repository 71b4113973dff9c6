// Accuracy yardstick on the device: the model of deepgrp/model.py:293-336 evaluated in plain fp32 -- fp32 weights,
// fp32 state, libm-grade expf/tanhf, no MFMA, no fp16 anywhere -- for a handful of windows.  It exists so that a user
// can measure, on THEIR weights and THEIR sequence, how far the fp16-operand fused kernel (gru_kernel.hip) is from
// full precision (`python -m deepgrp_amd verify`), and so that the tests have a third, independent statement of the
// forward pass that runs at sizes the CPU checker of the test suite does not.  Not a fallback: nothing on the prediction path calls it.
//
//   ref_rnn_kernel   one workgroup per 8 (window, strand) pairs: thread j owns unit j of each, h_{t-1} in LDS, U read
//                    coalesced from L2 (k-major rows), outputs h_t for every step  -> seq [nw][2][T][u], last [nw][2][u]
//   ref_head_kernel  one workgroup per window: Average, [additive attention], Dense, Softmax -> probs [nw][T][C]
#include "dgrp_model.h"
#include <mutex>

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// CELL 0: Keras GRU, reset_after=True, gate columns z|r|h, bias [2][3u] (input row, recurrent row)
// CELL 1: Keras LSTM, gate columns i|f|c|o, bias [4u]
// A workgroup carries RP (window, strand) pairs through all T steps; thread j owns unit j of every pair, so each
// element of U it loads is used RP times (the kernel is bound by L1/L2 reads of U otherwise).
#ifndef DGRP_REF_RP
#define DGRP_REF_RP 8
#endif
constexpr int RP = DGRP_REF_RP;

// NSL = unit slots per thread: thread j owns units j, j + 256, ... (one slot up to 256 units -- the yardstick of the fused kernels --,
// more for the models beyond the fused kernels' sizes, which run on these kernels: api.hip, "fp32 path")
template <int CELL, int NSL>
__global__ void __launch_bounds__(256) ref_rnn_kernel(const uint8_t *__restrict__ idx, int64_t s, int64_t w0, int64_t npairs,
                                                      int T, int u, const float *__restrict__ kernel,
                                                      const float *__restrict__ rec, const float *__restrict__ bias,
                                                      float *__restrict__ seq, float *__restrict__ last)
{
    constexpr int G = CELL ? 4 : 3;
    extern __shared__ float sh[];                            // h_{t-1}[k][pair], 256 NSL x RP floats
    const int64_t p0 = (int64_t)blockIdx.x * RP;
    const int gu = G * u;
    const uint8_t *x[RP];
    bool live[RP];
#pragma unroll
    for (int p = 0; p < RP; ++p) {
        live[p] = p0 + p < npairs;
        const int64_t pr = live[p] ? p0 + p : p0;
        x[p] = idx + (w0 + (pr >> 1)) * s;
    }
    float h[NSL][RP], c[NSL][RP];
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int p = 0; p < RP; ++p) { h[sl][p] = 0.0f; c[sl][p] = 0.0f; }
    for (int i = threadIdx.x; i < 256 * NSL * RP; i += blockDim.x) sh[i] = 0.0f;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        int base[RP];
#pragma unroll
        for (int p = 0; p < RP; ++p) {
            // deepgrp/model.py:266-279: the second pass reads the window backwards through the complement table
            const int dir = (int)((p0 + p) & 1);
            int b = dir ? x[p][T - 1 - t] : x[p][t];
            if (dir) b = b < 4 ? 3 - b : 4;
            base[p] = b;
        }
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
            const int j = threadIdx.x + 256 * sl;
            if (j >= u) continue;
            float ax[G][RP], ah[G][RP];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float bx = bias[g * u + j], bh = CELL == 0 ? bias[gu + g * u + j] : 0.0f;
#pragma unroll
                for (int p = 0; p < RP; ++p) {
                    ax[g][p] = kernel[(size_t)base[p] * gu + g * u + j] + bx;
                    ah[g][p] = bh;
                }
            }
            for (int k = 0; k < u; ++k) {
                const float *row = rec + (size_t)k * gu + j;
                float wv[G], hk[RP];
#pragma unroll
                for (int g = 0; g < G; ++g) wv[g] = row[g * u];
#pragma unroll
                for (int p = 0; p < RP; ++p) hk[p] = sh[k * RP + p];
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int p = 0; p < RP; ++p) ah[g][p] += hk[p] * wv[g];
            }
#pragma unroll
            for (int p = 0; p < RP; ++p) {
                if (CELL == 0) {
                    const float z = sigmoidf_(ax[0][p] + ah[0][p]);
                    const float r = sigmoidf_(ax[1][p] + ah[1][p]);
                    const float hh = tanhf(ax[2][p] + r * ah[2][p]);
                    h[sl][p] = z * h[sl][p] + (1.0f - z) * hh;
                } else {
                    const float ig = sigmoidf_(ax[0][p] + ah[0][p]);
                    const float fg = sigmoidf_(ax[1][p] + ah[1][p]);
                    const float og = sigmoidf_(ax[G - 1][p] + ah[G - 1][p]);
                    c[sl][p] = fg * c[sl][p] + ig * tanhf(ax[2][p] + ah[2][p]);
                    h[sl][p] = og * tanhf(c[sl][p]);
                }
            }
        }
        __syncthreads();                                     // every thread has read h_{t-1}
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
            const int j = threadIdx.x + 256 * sl;
            if (j >= u) continue;
#pragma unroll
            for (int p = 0; p < RP; ++p) {
                sh[j * RP + p] = h[sl][p];
                if (live[p]) seq[((size_t)(p0 + p) * T + t) * u + j] = h[sl][p];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
        const int j = threadIdx.x + 256 * sl;
        if (j >= u) continue;
#pragma unroll
        for (int p = 0; p < RP; ++p)
            if (live[p]) last[(size_t)(p0 + p) * u + j] = h[sl][p];
    }
}

__device__ __forceinline__ float block_reduce(float v, bool is_max, float *scratch)
{
    for (int o = 32; o > 0; o >>= 1) {
        const float y = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, y) : v + y;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = scratch[0];
    for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, scratch[i]) : r + scratch[i];
    return r;
}

__global__ void __launch_bounds__(256) ref_head_kernel(int T, int u, int C, int attention, const float *__restrict__ seq,
                                                       const float *__restrict__ last, const float *__restrict__ scale,
                                                       const float *__restrict__ ffk, const float *__restrict__ ffb,
                                                       float *__restrict__ e, float *__restrict__ probs)
{
    extern __shared__ float qc[];                            // q[u], ctx[u]
    __shared__ float ctxlogit[DGRP_MAXC], scratch[4];
    float *const q = qc, *const ctx = qc + u;
    const int64_t w = blockIdx.x;
    const float *fwd = seq + (size_t)(w * 2) * T * u, *rev = fwd + (size_t)T * u;
    const int tid = threadIdx.x;
    for (int c = tid; c < DGRP_MAXC; c += 256) ctxlogit[c] = 0.0f;
    if (attention) {
        // AdditiveAttention with the averaged final states as the single query (deepgrp/model.py:309-319)
        for (int k = tid; k < u; k += 256) q[k] = 0.5f * (last[(size_t)(w * 2) * u + k] + last[(size_t)(w * 2 + 1) * u + k]);
        __syncthreads();
        float *ew = e + (size_t)w * T;
        float mx = -INFINITY;
        for (int t = tid; t < T; t += 256) {
            float acc = 0.0f;
            for (int k = 0; k < u; ++k)
                acc += scale[k] * tanhf(q[k] + 0.5f * (fwd[(size_t)t * u + k] + rev[(size_t)t * u + k]));
            ew[t] = acc;
            mx = fmaxf(mx, acc);
        }
        mx = block_reduce(mx, true, scratch);
        float den = 0.0f;
        for (int t = tid; t < T; t += 256) {
            const float a = expf(ew[t] - mx);
            ew[t] = a;
            den += a;
        }
        den = block_reduce(den, false, scratch);
        __syncthreads();                                     // ew[] complete for every thread
        for (int k = tid; k < u; k += 256) {
            float acc = 0.0f;
            for (int t = 0; t < T; ++t)
                acc += (ew[t] / den) * 0.5f * (fwd[(size_t)t * u + k] + rev[(size_t)t * u + k]);
            ctx[k] = acc;
        }
        __syncthreads();
        if (tid < C) {
            float acc = 0.0f;
            for (int k = 0; k < u; ++k) acc += ctx[k] * ffk[(size_t)k * C + tid];     // rows 0..u-1: the context half
            ctxlogit[tid] = acc;
        }
    }
    __syncthreads();
    const float *wavg = ffk + (attention ? (size_t)u * C : 0);
    for (int t = tid; t < T; t += 256) {
        float lg[DGRP_MAXC];
        for (int c = 0; c < C; ++c) lg[c] = ffb[c] + ctxlogit[c];
        for (int k = 0; k < u; ++k) {
            const float a = 0.5f * (fwd[(size_t)t * u + k] + rev[(size_t)t * u + k]);
            for (int c = 0; c < C; ++c) lg[c] += a * wavg[(size_t)k * C + c];
        }
        float mx = lg[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, lg[c]);
        float den = 0.0f;
        for (int c = 0; c < C; ++c) { lg[c] = expf(lg[c] - mx); den += lg[c]; }
        for (int c = 0; c < C; ++c) probs[((size_t)w * T + t) * C + c] = lg[c] / den;
    }
}

struct ref_layout { int64_t seq, last, e, bytes; };
ref_layout ref_carve(const dgrp_model *m, int64_t nw)
{
    ref_layout l;
    int64_t off = 0;
    l.seq = off;  off += dgrp_align_up(nw * 2 * (int64_t)m->T * m->u * 4, 256);
    l.last = off; off += dgrp_align_up(nw * 2 * (int64_t)m->u * 4, 256);
    l.e = off;    off += dgrp_align_up(nw * (int64_t)m->T * 4, 256);
    l.bytes = off;
    return l;
}

}  // namespace

DGRP_EXPORT int64_t dgrp_forward_reference_workspace_bytes(const dgrp_model *m, int64_t nw)
{
    if (!m || nw < 0) return 0;
    return ref_carve(m, nw).bytes;
}

DGRP_EXPORT int dgrp_forward_windows_reference(const dgrp_model *m, const uint8_t *d_idx, int64_t n, int64_t s, int64_t w0,
                                               int64_t nw, float *d_probs, void *d_work, int64_t work_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    DGRP_REQUIRE(m && s >= 1 && w0 >= 0 && nw >= 0 && n >= 0, "dgrp_forward_windows_reference: bad arguments");
    if (nw == 0) return DGRP_OK;
    DGRP_REQUIRE(d_idx && d_probs && d_work, "dgrp_forward_windows_reference: NULL pointer");
    DGRP_REQUIRE((w0 + nw - 1) * s + m->T <= n, "dgrp_forward_windows_reference: window %lld runs past n=%lld",
                 (long long)(w0 + nw - 1), (long long)n);
    DGRP_REQUIRE(nw < (1ll << 30), "dgrp_forward_windows_reference: too many windows in one call");
    DGRP_REQUIRE(m->d_raw, "dgrp_forward_windows_reference: model carries no fp32 tensors");
    const ref_layout l = ref_carve(m, nw);
    if (work_bytes < l.bytes) {
        dgrp_set_error("dgrp_forward_windows_reference: workspace %lld < %lld bytes", (long long)work_bytes, (long long)l.bytes);
        return DGRP_ENOMEM;
    }
    float *seq = (float *)((char *)d_work + l.seq), *last = (float *)((char *)d_work + l.last);
    float *e = (float *)((char *)d_work + l.e);
    const float *raw = m->d_raw;
    const float *kernel = raw + m->raw_kernel, *rec = raw + m->raw_rec, *bias = raw + m->raw_bias;
    const float *ffk = raw + m->raw_ffk, *ffb = raw + m->raw_ffb, *scale = raw + m->raw_scale;
    const int64_t npairs = 2 * nw;
    const int nsl = (m->u + 255) / 256;
    DGRP_REQUIRE(nsl <= 8, "dgrp_forward_windows_reference: units=%d (up to 2048)", m->u);
    const dim3 grid((unsigned)((npairs + RP - 1) / RP)), block((unsigned)(nsl > 1 ? 256 : (m->u + 63) / 64 * 64));
    const size_t lds = (size_t)256 * (nsl <= 1 ? 1 : nsl <= 2 ? 2 : nsl <= 4 ? 4 : 8) * RP * sizeof(float);
#define REF_GO(CELLv, NSLv) do {                                                                                              \
        static std::once_flag once_; static hipError_t err_ = hipSuccess;                                                     \
        std::call_once(once_, [] { err_ = hipFuncSetAttribute((const void *)ref_rnn_kernel<CELLv, NSLv>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); }); \
        DGRP_HIP(err_);                                                                                                        \
        hipLaunchKernelGGL((ref_rnn_kernel<CELLv, NSLv>), grid, block, lds, stream, d_idx, s, w0, npairs, m->T, m->u, kernel, rec, bias, seq, last); } while (0)
    if (m->cell == 0) { if (nsl <= 1) REF_GO(0, 1); else if (nsl <= 2) REF_GO(0, 2); else if (nsl <= 4) REF_GO(0, 4); else REF_GO(0, 8); }
    else { if (nsl <= 1) REF_GO(1, 1); else if (nsl <= 2) REF_GO(1, 2); else if (nsl <= 4) REF_GO(1, 4); else REF_GO(1, 8); }
#undef REF_GO
    DGRP_LAUNCH_CHECK();
    hipLaunchKernelGGL(ref_head_kernel, dim3((unsigned)nw), dim3(256), (size_t)2 * m->u * sizeof(float), stream, m->T, m->u, m->C, m->attention, seq, last,
                       scale, ffk, ffb, e, d_probs);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

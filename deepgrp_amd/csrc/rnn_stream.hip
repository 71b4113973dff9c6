// A4 (+A5/A6 fused), split operands for the models the specialised split kernels (gru_kernel.hip: gru_split_kernel,
// gru_split2.hip) do not cover: GRU with 129-256 units (deepgrp/model.py:225-229; BASELINE configs[4] is 256 units with
// attention) and the LSTM cell (model.py:219-223).  With it EVERY model has an fp32-grade fused forward, which is what the
// reference computes (TensorFlow float32): the fp16-operand kernels stay as the explicit `--fast` mode.
//
// Same decomposition as the other recurrent kernels -- a workgroup owns 16 windows = 32 recurrent rows (windows and their
// reverse complements), wave w owns units [32w, 32w+32) of all gates, tile computed transposed (weights = A operand) -- and the
// same three-pass split: U = U_hi + U_lo, h_{t-1} = h_hi + h_lo as fp16 pairs, U.h ~ U_hi.h_hi + U_hi.h_lo + U_lo.h_hi with fp32
// accumulation.  Nothing but the input projection and the Dense fragments is resident: the recurrent fragments of both halves
// (256 units: 96 KB per wave and step) STREAM from L2 through a register ring in consumption order (k-step major, hi then lo,
// gates in pack order).  At 256 units the matrix pipe is the bound (288 MFMAs per wave-step, two waves per SIMD: ~80 % busy), not the
// stream: keeping fragments resident or a deeper ring changes nothing -- about 0.4 x the speed of the fp16-operand
// kernel of the same model, which issues a third of the MFMAs.  Correctness first: no staging of vector work into MFMA gaps.
// (r03, SQ counters at 256 units: waves 54 % of their time in s_waitcnt, matrix pipe ~50 % busy, 768 KB of fragments per CU and step
// through a vector-memory path of 64 B per clock = 12 k cycles next to 9.2 k cycles of MFMAs: the CU's L2 -> register path is what it
// waits for.  U_hi of r and of the candidate resident in AGPRs -- a third less to stream -- does not fit this shape: two waves per SIMD
// leave 256 registers each, 128 resident + 80 of accumulators and state spill 180 registers at 8 waves; it needs four waves of 64
// units with the 512-register budget: DESIGN.md 6.)
// GRU: two-reciprocal gate chain of gru_shared.h (beyond 128 units the one-reciprocal form is not offered);
// LSTM: c = f c + i tanh(z_c), h = o tanh(c) exactly as lstm_fused_kernel evaluates them (accumulators in the exp2 domain).
#include "gru_shared.h"
#include <mutex>

template <int CELL, int NW, int MODE>
__global__ void __launch_bounds__(64 * NW, 2) rnn_split_stream_kernel(const gru_params pin)
{
    gru_params p = pin;
    const int64_t bid = wg_record<MODE>(pin, p);
    constexpr int G = CELL ? 4 : 3, UP = 32 * NW, KS = UP / 16, HS = UP + 8, NF = 2 * G, NFRAG = KS * NF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, C = p.C;
    const uint4 *mypack = p.pack + (size_t)wave * p.nfrag * 64 + lane;
    const uint4 *mystream = p.stream + (size_t)wave * NFRAG * 64 + lane;
    // resident: the input k-step of every gate (one-hot rows + biases, hi|lo), the candidate's input projection (GRU), Dense
    half8 Bin[G], Bxh, Bd_hi, Bd_lo;
#pragma unroll
    for (int g = 0; g < G; ++g) Bin[g] = __builtin_bit_cast(half8, mypack[(size_t)(g * (KS + 1) + KS) * 64]);
    Bxh = __builtin_bit_cast(half8, mypack[(size_t)(G * (KS + 1)) * 64]);                      // (LSTM: this is Dense hi, unused as Bxh)
    Bd_hi = __builtin_bit_cast(half8, mypack[(size_t)(G * (KS + 1) + (CELL ? 0 : 1)) * 64]);
    Bd_lo = __builtin_bit_cast(half8, mypack[(size_t)(G * (KS + 1) + (CELL ? 1 : 2)) * 64]);

    _Float16 *const lbuf = reinterpret_cast<_Float16 *>(smem + p.lo_tile_off);          // [2][32][HS] lo tiles
    for (int i = tid; i < 32 * HS; i += 64 * NW) lbuf[i] = (_Float16)0.0f;
    const wg_ctx ctx = wg_setup<NW, MODE>(p, smem, bid);                                  // ends with a barrier
    float *const dpart = ctx.dpart;

    const int r = lane & 31, wi_a = r & 15, dir = r >> 4, khalf = lane >> 5;
    const uint8_t *myseq = ctx.seqs + wi_a * p.Tp;
    float h[16], c[CELL ? 16 : 1];
#pragma unroll
    for (int i = 0; i < 16; ++i) h[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < (CELL ? 16 : 1); ++i) c[i] = 0.0f;
    _Float16 *hcur = ctx.hbuf, *hnxt = ctx.hbuf + 32 * HS, *lcur = lbuf, *lnxt = lbuf + 32 * HS;
    const f32x16 zero16 = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const f32x4 zero4 = { 0, 0, 0, 0 };
    const int cls = lane & 15;
    const float fbias = cls < C ? p.ffb[cls] : 0.0f;
    auto finish_step = [&](int t) {
        for (int reg = wave; reg < 4; reg += NW) {
            const int wi = 4 * (lane >> 4) + reg;
            finish_register<NW, MODE>(p, ctx, t, reg, fbias, ctx.rowoff[wi], ctx.row0s[wi]);
        }
    };
    const int doff = (lane & 15) * HS + 32 * wave + 8 * (lane >> 4);
    auto dense_issue = [&](const _Float16 *hb, const _Float16 *lb, int tt) -> f32x4 {
        const half8 a0 = *reinterpret_cast<const half8 *>(hb + doff), a1 = *reinterpret_cast<const half8 *>(hb + doff + 16 * HS);
        const half8 l0 = *reinterpret_cast<const half8 *>(lb + doff), l1 = *reinterpret_cast<const half8 *>(lb + doff + 16 * HS);
        if (MODE == 2 && (lane & 15) < ctx.nvalid)
            split_avg_store(p, ctx.wg_w, tt, UP, wave, a0, a1, l0, l1);
        f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, Bd_hi, zero4, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, Bd_hi, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, Bd_lo, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, Bd_lo, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(l0, Bd_hi, d, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(l1, Bd_hi, d, 0, 0, 0);
    };
    auto dense_store = [&](int t, const f32x4 &d) {
        float *dw = dpart + ((size_t)(t & 1) * 4 * NW + wave) * 64 + lane;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) dw[reg * NW * 64] = d[reg];
    };

    for (int t = 0; t < T; ++t) {
        uint32_t b = myseq[dir ? T - 1 - t : t];
        if (dir) b = b < 4 ? 3 - b : 4;                      // complement table [3,2,1,0,4], model.py:233-237
        const uint32_t one = 0x3C00u << ((b & 1) * 16);
        const uint32_t sel = b >> 1;
        const uint4 xu = make_uint4(sel == 0 ? one : 0u, sel == 1 ? one : 0u, (sel == 2 ? one : 0u) | 0x3C000000u, 0u);
        const half8 xa = __builtin_bit_cast(half8, xu);
        const _Float16 *arow = hcur + r * HS + 8 * khalf, *lrow = lcur + r * HS + 8 * khalf;

        // the ring holds one k-step of fragments (NF = 2 G): slot j always carries fragment j of a k-step, so the k loop can stay
        // ROLLED (fully unrolled, 16 k-steps of loads hoisted ahead of their use cost more registers than the file has)
        uint4 q[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) q[i] = mystream[(size_t)i * 64];
        f32x16 acc[G], ax = zero16;
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bin[g], xa, zero16, 0, 0, 0);
        if (CELL == 0) ax = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bxh, xa, zero16, 0, 0, 0);     // the candidate's input projection
#pragma unroll 1
        for (int k = 0; k < KS; ++k) {
            const half8 hf = *reinterpret_cast<const half8 *>(arow + 16 * k);
            const half8 lf = *reinterpret_cast<const half8 *>(lrow + 16 * k);
            const int kn = k + 1 < KS ? k + 1 : k;                 // (the last k-step re-requests its own fragments: harmless)
            const uint4 *nxt = mystream + (size_t)kn * NF * 64;
            half8 wh[G], wl[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                wh[g] = __builtin_bit_cast(half8, q[g]);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[g], hf, acc[g], 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[g], lf, acc[g], 0, 0, 0);
                q[g] = nxt[(size_t)g * 64];                       // slot free: request the next k-step's hi fragment
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                wl[g] = __builtin_bit_cast(half8, q[G + g]);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[g], hf, acc[g], 0, 0, 0);
                q[G + g] = nxt[(size_t)(G + g) * 64];
            }
        }
        f32x4 dpl = zero4;
        if (t > 0) dpl = dense_issue(hcur, lcur, t - 1);
        if (t > 1) finish_step(t - 2);
        if (t > 0) dense_store(t - 1, dpl);
        if (CELL == 0) {
            // pack order of the GRU gates: z, r, h (api.hip)
#pragma unroll
            for (int i = 0; i < 16; ++i) h[i] = split_gate_chain<false>(acc[1][i], acc[2][i], acc[0][i], ax[i], h[i]);
        } else {
            // i | f | c | o; c = f*c + i*tanh(z_c); h = o*tanh(c)
#pragma unroll
            for (int i = 0; i < 16; ++i) split_lstm_cell(acc[0][i], acc[1][i], acc[2][i], acc[G - 1][i], c[CELL ? i : 0], h[i]);
        }
        // publish h_t as an fp16 pair: hi = fp16(h), lo = fp16(h - hi)
        _Float16 *wrow = hnxt + (lane & 31) * HS + 32 * wave + 4 * khalf;
        _Float16 *wlow = lnxt + (lane & 31) * HS + 32 * wave + 4 * khalf;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float h4[4] = { h[4 * qd], h[4 * qd + 1], h[4 * qd + 2], h[4 * qd + 3] };
            uint2 hv, lv;
            split_hi_lo4(h4, hv, lv);
            *reinterpret_cast<uint2 *>(wrow + 8 * qd) = hv;
            *reinterpret_cast<uint2 *>(wlow + 8 * qd) = lv;
        }
        __syncthreads();
        _Float16 *tmp = hcur; hcur = hnxt; hnxt = tmp;
        tmp = lcur; lcur = lnxt; lnxt = tmp;
    }
    {
        const f32x4 dpl = dense_issue(hcur, lcur, T - 1);
        if (T > 1) finish_step(T - 2);
        dense_store(T - 1, dpl);
        __syncthreads();
        finish_step(T - 1);
    }
    if (MODE == 0 && p.ospan > 0) flush_image<NW>(p, ctx);
}

template <int CELL, int NW>
static int launch_stream(const gru_params &p, int64_t groups, size_t lds, hipStream_t stream)
{
    static std::once_flag configured;
    static hipError_t cfg_err = hipSuccess;
    std::call_once(configured, [] {
        auto set = [](const void *f) { const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (e != hipSuccess) cfg_err = e; };
        set((const void *)rnn_split_stream_kernel<CELL, NW, 0>);
        set((const void *)rnn_split_stream_kernel<CELL, NW, 1>);
        if (CELL == 0) set((const void *)rnn_split_stream_kernel<CELL, NW, CELL ? 1 : 2>);
    });
    DGRP_HIP(cfg_err);
    if (p.mode == 0)
        hipLaunchKernelGGL((rnn_split_stream_kernel<CELL, NW, 0>), dim3((unsigned)groups), dim3(64 * NW), lds, stream, p);
    else if (p.mode == 1 || CELL == 1)
        hipLaunchKernelGGL((rnn_split_stream_kernel<CELL, NW, 1>), dim3((unsigned)groups), dim3(64 * NW), lds, stream, p);
    else
        hipLaunchKernelGGL((rnn_split_stream_kernel<CELL, NW, CELL ? 1 : 2>), dim3((unsigned)groups), dim3(64 * NW), lds, stream, p);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

// GRU with 5..8 waves (129-256 units), LSTM with 1..8 waves (up to 256 units)
int dgrp_stream_launch(const gru_params &p, int cell, int NW, int64_t groups, size_t lds, hipStream_t stream)
{
    if (cell == 0) {
        switch (NW) {
        case 5: return launch_stream<0, 5>(p, groups, lds, stream);
        case 6: return launch_stream<0, 6>(p, groups, lds, stream);
        case 7: return launch_stream<0, 7>(p, groups, lds, stream);
        case 8: return launch_stream<0, 8>(p, groups, lds, stream);
        default: break;
        }
    } else {
        switch (NW) {
        case 1: return launch_stream<1, 1>(p, groups, lds, stream);
        case 2: return launch_stream<1, 2>(p, groups, lds, stream);
        case 3: return launch_stream<1, 3>(p, groups, lds, stream);
        case 4: return launch_stream<1, 4>(p, groups, lds, stream);
        case 5: return launch_stream<1, 5>(p, groups, lds, stream);
        case 6: return launch_stream<1, 6>(p, groups, lds, stream);
        case 7: return launch_stream<1, 7>(p, groups, lds, stream);
        case 8: return launch_stream<1, 8>(p, groups, lds, stream);
        default: break;
        }
    }
    dgrp_set_error("no streamed split-operand kernel for cell %d with %d units", cell, 32 * NW);
    return DGRP_EINVAL;
}

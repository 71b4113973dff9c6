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

// ---- GRU, 129-256 units: waves of 64 units with the 512-register budget, U_hi of r and of the candidate RESIDENT -----------------
// What bounds rnn_split_stream_kernel at 256 units is the CU's vector-memory path, not the matrix pipe: every wave re-reads its 96 KB
// of fragments from L2 every step -- 768 KB per CU and step through a path that moves 64 B per clock: 12 k cycles, next to 9.2 k cycles
// of MFMAs for the SIMD's two waves (r03 SQ counters: waves 54 % of their time in s_waitcnt, matrix pipe ~50 % busy).  With two waves
// per SIMD a wave has 256 registers: no room for resident fragments next to 80 registers of accumulators and state.  Here a workgroup
// is ceil(UP / 64) waves, ONE per SIMD, each owning 64 units (two 32-unit halves) and all 512 registers: the hi fragments of the r and
// the candidate gate -- a third of all fragments, 16 KS registers, 256 at 256 units -- live in AGPRs (inline-asm MFMAs name them there;
// loaded once), U_hi of z and the three U_lo stream (512 KB per CU and step instead of 768), one k-step in flight.  The input
// projection is a table row the accumulators start from (LDS, as in gru_split2_kernel), the candidate's input projection is read from
// the table when the gate chain needs it.  Same tile, same three passes, same gate chain and rounding as rnn_split_stream_kernel.
// The stream layout is that kernel's: [32-unit slice v][k-step][hi z, hi r, hi h, lo z, lo r, lo h][64].
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define SMFMA_A(acc, Wf, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(Wf), "v"(b))
#define SMFMA_V(acc, Wf, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(Wf), "v"(b))
#define SLOAD2A(a, pa, b, pb)                                                                                           \
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(0)"            \
                 : "=&a"(a), "=&a"(b) : "v"(pa), "v"(pb) : "memory")

template <int NW, int MODE>               // NW = 32-unit slices of the model (5..8); waves = (NW + 1) / 2
__global__ void __launch_bounds__(64 * ((NW + 1) / 2)) __attribute__((amdgpu_waves_per_eu(1, 1))) gru_stream64_kernel(const gru_params pin)
{
    gru_params p = pin;
    const int64_t bid = wg_record<MODE>(pin, p);
    constexpr int NW64 = (NW + 1) / 2, UP = 32 * NW, KS = UP / 16, HS = UP + 8, NF = 6, NFRAG = KS * NF;
    constexpr int XT_PITCH = 4 * UP * 4 + 32;                    // table row of one base: 4 kinds x UP units fp32 + 32 B (bank spread)
    constexpr int NT = 64 * NW64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, C = p.C;
    const bool two = 2 * wave + 1 < NW;                           // this wave's second 32-unit half exists (odd NW: not in the last wave)
    const uint4 *st0 = p.stream + (size_t)(2 * wave) * NFRAG * 64 + lane;
    const uint4 *st1 = p.stream + (size_t)(two ? 2 * wave + 1 : 2 * wave) * NFRAG * 64 + lane;
    // resident: U_hi of r (slot 1), both halves: 8 KS AGPRs.  Everything else streams through a ring of D k-steps that runs on across
    // the time steps (the weights do not depend on t: k-step k of step t + 1 is requested while step t is still in its gate phase), so
    // that no step starts by waiting for its first fragments.
    constexpr int D = 2;                                          // k-steps in flight; KS is even
    static_assert(KS % D == 0, "the ring's slot of a k-step must not depend on the time step");
    u32x4 Wr[KS][2];
#pragma unroll
    for (int k = 0; k < KS; ++k) SLOAD2A(Wr[k][0], st0 + (size_t)(k * NF + 1) * 64, Wr[k][1], st1 + (size_t)(k * NF + 1) * 64);
    // Dense fragments of the wave's two slices (the 32-unit pack of api.hip)
    half8 Bd_hi[2], Bd_lo[2];
#pragma unroll
    for (int uh = 0; uh < 2; ++uh) {
        const uint4 *mypack = p.pack + (size_t)(uh == 1 && two ? 2 * wave + 1 : 2 * wave) * p.nfrag * 64 + lane;
        Bd_hi[uh] = __builtin_bit_cast(half8, mypack[(size_t)(3 * (KS + 1) + 1) * 64]);
        Bd_lo[uh] = __builtin_bit_cast(half8, mypack[(size_t)(3 * (KS + 1) + 2) * 64]);
    }
    // the input-projection table: [5 bases][4 kinds r, g (recurrent bias only), z, x][UP units] fp32 -> LDS rows of XT_PITCH bytes
    for (int i = tid; i < 5 * 4 * UP; i += NT)
        *reinterpret_cast<float *>(smem + p.xtab_off + (i / (4 * UP)) * XT_PITCH + (i % (4 * UP)) * 4) = p.xtab[i];

    _Float16 *const lbuf = reinterpret_cast<_Float16 *>(smem + p.lo_tile_off);          // [2][32][HS] lo tiles
    for (int i = tid; i < 32 * HS; i += NT) lbuf[i] = (_Float16)0.0f;
    // carve, staged sequences, placement: wg_setup's job for 64 NW64 threads and a tile of 32 NW units
    wg_ctx ctx;
    {
        ctx.hbuf = reinterpret_cast<_Float16 *>(smem);
        ctx.dpart = reinterpret_cast<float *>(smem + gru_lds_hbuf(UP, 8));
        ctx.seqs = smem + gru_lds_hbuf(UP, 8) + gru_lds_dpart(NW64);
        ctx.row0s = reinterpret_cast<int64_t *>(ctx.seqs + gru_lds_seq(p.Tp));
        ctx.rowoff = reinterpret_cast<int *>(ctx.row0s + DGRP_WG_WINDOWS);
        ctx.obuf = reinterpret_cast<unsigned *>(ctx.rowoff + DGRP_WG_WINDOWS);
        ctx.wg_w = p.w0 + bid * DGRP_WG_WINDOWS;
        ctx.nvalid = (int)min((int64_t)DGRP_WG_WINDOWS, p.w0 + p.nw - ctx.wg_w);
        for (int i = tid; i < DGRP_WG_WINDOWS * T; i += NT) {
            const int wi = i / T, t = i - wi * T;
            ctx.seqs[wi * p.Tp + t] = wi < ctx.nvalid ? p.idx[(ctx.wg_w + wi) * p.s + t] : (uint8_t)4;
        }
        for (int i = tid; i < 32 * HS; i += NT) ctx.hbuf[i] = (_Float16)0.0f;          // h_{-1} = 0
        ctx.lo = 0;
        if (MODE == 0) {
            const int64_t a = dgrp_place_row(p.place, ctx.wg_w, p.s), b = dgrp_place_row(p.place, ctx.wg_w + ctx.nvalid - 1, p.s);
            ctx.lo = a < b ? a : b;
            for (int i = tid; i < p.ospan * C; i += NT) ctx.obuf[i] = 0u;
        }
        if (tid < DGRP_WG_WINDOWS) {
            int64_t r0 = -1;
            int off = -1;
            if (tid < ctx.nvalid) {
                r0 = MODE == 0 ? dgrp_place_row(p.place, ctx.wg_w + tid, p.s) : (ctx.wg_w + tid - p.w0 + p.avgw) * (int64_t)T;
                if (MODE == 0 && r0 >= ctx.lo && r0 - ctx.lo + T <= p.ospan) off = (int)(r0 - ctx.lo);
            }
            ctx.row0s[tid] = r0;
            ctx.rowoff[tid] = off;
        }
        __syncthreads();
    }
    float *const dpart = ctx.dpart;

    const int r = lane & 31, wi_a = r & 15, dir = r >> 4, khalf = lane >> 5;
    const uint8_t *myseq = ctx.seqs + wi_a * p.Tp;
    float h[2][16];
#pragma unroll
    for (int uh = 0; uh < 2; ++uh)
#pragma unroll
        for (int i = 0; i < 16; ++i) h[uh][i] = 0.0f;
    _Float16 *hcur = ctx.hbuf, *hnxt = ctx.hbuf + 32 * HS, *lcur = lbuf, *lnxt = lbuf + 32 * HS;
    const f32x4 zero4 = { 0, 0, 0, 0 };
    const int cls = lane & 15;
    const float fbias = cls < C ? p.ffb[cls] : 0.0f;
    auto finish_step = [&](int t) {
        for (int reg = wave; reg < 4; reg += NW64) {
            const int wi = 4 * (lane >> 4) + reg;
            finish_register<NW64, MODE>(p, ctx, t, reg, fbias, ctx.rowoff[wi], ctx.row0s[wi]);
        }
    };
    auto dense_issue = [&](const _Float16 *hb, const _Float16 *lb, int tt) -> f32x4 {
        f32x4 d = zero4;
#pragma unroll
        for (int uh = 0; uh < 2; ++uh) {
            if (uh == 1 && !two) break;
            const int doff = (lane & 15) * HS + 64 * wave + 32 * uh + 8 * (lane >> 4);
            const half8 a0 = *reinterpret_cast<const half8 *>(hb + doff), a1 = *reinterpret_cast<const half8 *>(hb + doff + 16 * HS);
            const half8 l0 = *reinterpret_cast<const half8 *>(lb + doff), l1 = *reinterpret_cast<const half8 *>(lb + doff + 16 * HS);
            if (MODE == 2 && (lane & 15) < ctx.nvalid)
                split_avg_store(p, ctx.wg_w, tt, UP, 2 * wave + uh, a0, a1, l0, l1);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, Bd_hi[uh], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, Bd_hi[uh], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, Bd_lo[uh], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, Bd_lo[uh], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(l0, Bd_hi[uh], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(l1, Bd_hi[uh], d, 0, 0, 0);
        }
        return d;
    };
    auto dense_store = [&](int t, const f32x4 &d) {
        float *dw = dpart + ((size_t)(t & 1) * 4 * NW64 + wave) * 64 + lane;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) dw[reg * NW64 * 64] = d[reg];
    };
    auto ldsf4 = [&](unsigned off) -> f32x4 { return *reinterpret_cast<const f32x4 *>(smem + off); };
    // table offset of this lane's units: 64 wave + 32 uh + 8 qd + 4 khalf .. + 3
    const unsigned tab_lane = (unsigned)p.xtab_off + (unsigned)(64 * wave + 4 * khalf) * 4;

    uint4 q[D][2][5];
#pragma unroll
    for (int k = 0; k < D; ++k)
#pragma unroll
        for (int uh = 0; uh < 2; ++uh) {
            const uint4 *src = (uh ? st1 : st0) + (size_t)k * NF * 64;
            q[k][uh][0] = src[0]; q[k][uh][1] = src[(size_t)3 * 64]; q[k][uh][2] = src[(size_t)4 * 64]; q[k][uh][3] = src[(size_t)5 * 64];
            q[k][uh][4] = src[(size_t)2 * 64];
        }
    // accumulators start as the table rows of the step's base (kinds 0 r, 1 g, 2 z) -- requested at the END of the step before, in front
    // of its barrier: the table does not depend on the hidden state, and read behind the barrier the 24 LDS reads were latency every
    // wave of the workgroup waited out together
    f32x16 ar[2], ag[2], az[2];
    unsigned tab = 0;
    auto acc_start = [&](int tn) {
        uint32_t b = myseq[dir ? T - 1 - tn : tn];
        if (dir) b = b < 4 ? 3 - b : 4;                      // complement table [3,2,1,0,4], model.py:233-237
        tab = tab_lane + b * XT_PITCH;
#pragma unroll
        for (int uh = 0; uh < 2; ++uh)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const f32x4 vr = ldsf4(tab + 0 * UP * 4 + (32 * uh + 8 * qd) * 4), vg = ldsf4(tab + 1 * UP * 4 + (32 * uh + 8 * qd) * 4);
                const f32x4 vz = ldsf4(tab + 2 * UP * 4 + (32 * uh + 8 * qd) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) { ar[uh][4 * qd + i] = vr[i]; ag[uh][4 * qd + i] = vg[i]; az[uh][4 * qd + i] = vz[i]; }
            }
    };
    acc_start(0);
    for (int t = 0; t < T; ++t) {
        const unsigned tab_t = tab;                            // this step's table row (the candidate's input projection is read below)
        const _Float16 *arow = hcur + r * HS + 8 * khalf, *lrow = lcur + r * HS + 8 * khalf;
        // streamed per k-step and half: U_hi of z (slot 0) and the three U_lo (slots 3 z, 4 r, 5 h)
        // (the stream pointers are laundered every step: left to itself the compiler hoists the fragment addresses of the unrolled
        // loop out of the time loop and spills 200 registers)
        const uint4 *sp[2] = { st0, st1 };
        asm volatile("" : "+v"(sp[0]), "+v"(sp[1]));
        // the hidden tile's fragments one k-step ahead of their MFMAs (read right in front of them every k-step waited out an LDS latency)
        half8 hfn = *reinterpret_cast<const half8 *>(arow), lfn = *reinterpret_cast<const half8 *>(lrow);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const half8 hf = hfn, lf = lfn;
            if (k + 1 < KS) {
                hfn = *reinterpret_cast<const half8 *>(arow + 16 * (k + 1));
                lfn = *reinterpret_cast<const half8 *>(lrow + 16 * (k + 1));
            }
#pragma unroll
            for (int uh = 0; uh < 2; ++uh) {
                if (uh == 1 && !two) break;
                // slots of a k-step: 0 hi z, 1 lo z, 2 lo r, 3 lo h, 4 hi h; the fragments of k-step (k + D) mod KS go where these were
                uint4 (&qk)[5] = q[k % D][uh];
                const uint4 *nxt = sp[uh] + (size_t)((k + D) % KS) * NF * 64;
                const u32x4 zh = __builtin_bit_cast(u32x4, qk[0]), zl = __builtin_bit_cast(u32x4, qk[1]);
                const u32x4 rl = __builtin_bit_cast(u32x4, qk[2]), gl = __builtin_bit_cast(u32x4, qk[3]), gh = __builtin_bit_cast(u32x4, qk[4]);
                SMFMA_V(az[uh], zh, hf); SMFMA_A(ar[uh], Wr[k][uh], hf); SMFMA_V(ag[uh], gh, hf);
                SMFMA_V(az[uh], zh, lf); SMFMA_A(ar[uh], Wr[k][uh], lf); SMFMA_V(ag[uh], gh, lf);
                qk[0] = nxt[0]; qk[4] = nxt[(size_t)2 * 64];
                SMFMA_V(az[uh], zl, hf); SMFMA_V(ar[uh], rl, hf); SMFMA_V(ag[uh], gl, hf);
                qk[1] = nxt[(size_t)3 * 64]; qk[2] = nxt[(size_t)4 * 64]; qk[3] = nxt[(size_t)5 * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(az[0]), "+v"(ar[0]), "+v"(ag[0]), "+v"(az[1]), "+v"(ar[1]), "+v"(ag[1]));   // asm MFMA results -> compiler-scheduled readers
        f32x4 dpl = zero4;
        if (t > 0) dpl = dense_issue(hcur, lcur, t - 1);
        if (t > 1) finish_step(t - 2);
        if (t > 0) dense_store(t - 1, dpl);
        // pack order of the GRU gates: z, r, h (api.hip); the candidate's input projection (kind 3) from the table, 4 units at a time
#pragma unroll
        for (int uh = 0; uh < 2; ++uh) {
            if (uh == 1 && !two) break;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const f32x4 vx = ldsf4(tab_t + 3 * UP * 4 + (32 * uh + 8 * qd) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    h[uh][4 * qd + i] = split_gate_chain<false>(ar[uh][4 * qd + i], ag[uh][4 * qd + i], az[uh][4 * qd + i], vx[i], h[uh][4 * qd + i]);
            }
            // publish h_t as an fp16 pair: hi = fp16(h), lo = fp16(h - hi)
            _Float16 *wrow = hnxt + (lane & 31) * HS + 64 * wave + 32 * uh + 4 * khalf;
            _Float16 *wlow = lnxt + (lane & 31) * HS + 64 * wave + 32 * uh + 4 * khalf;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const float h4[4] = { h[uh][4 * qd], h[uh][4 * qd + 1], h[uh][4 * qd + 2], h[uh][4 * qd + 3] };
                uint2 hv, lv;
                split_hi_lo4(h4, hv, lv);
                *reinterpret_cast<uint2 *>(wrow + 8 * qd) = hv;
                *reinterpret_cast<uint2 *>(wlow + 8 * qd) = lv;
            }
        }
        if (t + 1 < T) acc_start(t + 1);
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
    if (MODE == 0 && p.ospan > 0) flush_image<NW64>(p, ctx);
}

template <int NW>
static int launch_stream64(const gru_params &p, int64_t groups, size_t lds, hipStream_t stream)
{
    static std::once_flag configured;
    static hipError_t cfg_err = hipSuccess;
    std::call_once(configured, [] {
        auto set = [](const void *f) { const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (e != hipSuccess) cfg_err = e; };
        set((const void *)gru_stream64_kernel<NW, 0>);
        set((const void *)gru_stream64_kernel<NW, 1>);
        set((const void *)gru_stream64_kernel<NW, 2>);
    });
    DGRP_HIP(cfg_err);
    constexpr unsigned NT = 64 * ((NW + 1) / 2);
    if (p.mode == 0) hipLaunchKernelGGL((gru_stream64_kernel<NW, 0>), dim3((unsigned)groups), dim3(NT), lds, stream, p);
    else if (p.mode == 1) hipLaunchKernelGGL((gru_stream64_kernel<NW, 1>), dim3((unsigned)groups), dim3(NT), lds, stream, p);
    else hipLaunchKernelGGL((gru_stream64_kernel<NW, 2>), dim3((unsigned)groups), dim3(NT), lds, stream, p);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

// GRU with 129-256 units on waves of 64 units; `lds` = the carve of dgrp_stream64_carve
int dgrp_stream64_launch(const gru_params &p, int NW, int64_t groups, size_t lds, hipStream_t stream)
{
    switch (NW) {
    case 5: return launch_stream64<5>(p, groups, lds, stream);
    case 6: return launch_stream64<6>(p, groups, lds, stream);
    case 7: return launch_stream64<7>(p, groups, lds, stream);
    case 8: return launch_stream64<8>(p, groups, lds, stream);
    default:
        dgrp_set_error("dgrp_stream64_launch: %d slices of 32 units (5..8)", NW);
        return DGRP_EINVAL;
    }
}

// LDS carve: hi tiles, partial logits of (NW + 1) / 2 waves, sequences, placement, image (as much as `budget` allows), lo tiles, table.
// Sets p.ospan, p.lo_tile_off, p.xtab_off; returns the bytes.
size_t dgrp_stream64_carve(int NW, gru_params &p, int mode, int64_t s, int64_t budget)
{
    const int UP = 32 * NW, NW64 = (NW + 1) / 2;
    const int tiles = gru_lds_hbuf(UP, 8), xtab = 5 * (4 * UP * 4 + 32);
    const int fixed = tiles + gru_lds_dpart(NW64) + gru_lds_seq(p.Tp) + gru_lds_meta();
    p.ospan = 0;
    if (mode == 0) {
        const int64_t want = (DGRP_WG_WINDOWS - 1) * s + p.T;
        const int64_t cap = (budget - fixed - tiles - xtab) / (p.C * 4);
        p.ospan = (int)(want < cap ? want : cap);
        if (p.ospan < p.T) p.ospan = 0;
    }
    p.lo_tile_off = (int)dgrp_align_up(fixed + (int64_t)p.ospan * p.C * 4, 16);
    p.xtab_off = p.lo_tile_off + tiles;
    return (size_t)p.xtab_off + xtab;
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

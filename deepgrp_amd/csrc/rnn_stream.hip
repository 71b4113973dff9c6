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
// (r03: the GRU with 129-256 units moved to gru_stream64_kernel below -- waves of 64 units with the 512-register budget and a third of
// the fragments resident; this kernel now serves the LSTM cell only.)
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
// A workgroup is ceil(UP / 64) waves, ONE per SIMD, each owning 64 units (two 32-unit halves) of a 16-window tile and all 512
// registers.  What decides the time of a step at this size is (r03 measurements):
//   * the CU's vector-memory path: tools/ubench/l2_stream.hip -- four waves re-reading an L2-resident buffer with 16-byte loads get
//     117-135 GB/s per CU (49-56 B per clock), 92 GB/s while the CU's other waves read LDS; L1-resident data is no faster.  Streaming
//     ALL recurrent fragments (768 KB per tile-step at 256 units) therefore costs 6-8 us per tile-step whatever the ring depth -- more
//     than the step's 288 MFMAs (4.4 us).  Hence: U_hi of r and of the candidate live in AGPRs (16 KS registers, 256 at 256 units;
//     inline-asm MFMAs name them there; loaded once); U_hi of z and the three U_lo stream (512 KB) through a ring of D blocks that
//     runs on across the time steps, every slot requested again right behind its last MFMA, from a uniform base + one lane offset
//     in the GLOBAL address space (a laundered generic pointer made them FLAT loads, which count on both wait counters: every
//     k-step then waited for every load in flight).
//   * the phases of a step following one another (MFMAs 9.2 k cycles, gate math 4.5 k, waiting): the two 32-unit halves of a wave are
//     therefore worked one after the other -- all k-steps of half 0, then all k-steps of half 1 WITH the gate chains of half 0, link
//     by link, in the gaps between those MFMAs (pinned with sched_barrier), half 0's new state published meanwhile (ping-pong tiles);
//     only half 1's gate math is left behind the MFMAs.
//   (Measured and dropped: two row tiles per workgroup as two wave groups half a step apart, two waves per SIMD with 256 registers
//   each -- the overlap is there, no-load time 12.4 k cycles per tile-step against 15 k, but nothing can be resident and the stream
//   alone takes longer than that: 21.8 Mbp/s against 23.1 for this form on the cfg5 shape without attention.  The ring's depth does
//   not matter (2, 4: the same time) and loads from an L1-resident 6 KB cost what loads from L2 cost; WITHOUT the loads the step is
//   19 % shorter: what a streamed kilobyte costs is its way into the SIMD's registers, not its latency.)
// The state of a unit IS its published fp16 pair (h = hi + lo, exact in float, 2^-22 of h away from the float the gate chain
// produced): nothing but accumulators, ring and the current chain's temporaries lives in VGPRs.  The input projection is a table row
// the accumulators start from (LDS, as in gru_split2_kernel).  Same three passes and gate chain as rnn_split_stream_kernel; the
// stream layout is that kernel's: [32-unit slice v][k-step][hi z, hi r, hi h, lo z, lo r, lo h][64].
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define SMFMA_A(acc, Wf, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(Wf), "v"(b))
#define SMFMA_V(acc, Wf, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(Wf), "v"(b))
#define SLOAD2A(a, pa, b, pb)                                                                                           \
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(0)"            \
                 : "=&a"(a), "=&a"(b) : "v"(pa), "v"(pb) : "memory")
#ifndef DGRP_S64_RING
#define DGRP_S64_RING 2
#endif
#ifndef DGRP_S64_DIAG
#define DGRP_S64_DIAG 0
#endif

template <int NW, int MODE>               // NW = 32-unit slices of the model (5..8); waves = (NW + 1) / 2
__global__ void __launch_bounds__(64 * ((NW + 1) / 2)) __attribute__((amdgpu_waves_per_eu(1, 1))) gru_stream64_kernel(const gru_params pin)
{
    gru_params p = pin;
    const int64_t bid = wg_record<MODE>(pin, p);
    constexpr int NW64 = (NW + 1) / 2, UP = 32 * NW, KS = UP / 16, HS = UP + 8, NF = 6, NFRAG = KS * NF;
    constexpr int XT_PITCH = 4 * UP * 4 + 32;                    // table row of one base: 4 kinds x UP units fp32 + 32 B (bank spread)
    constexpr int NT = 64 * NW64;
    constexpr int TILE = 32 * HS;                                // halves of one hidden tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, C = p.C;
    const bool two = 2 * wave + 1 < NW;                           // this wave's second 32-unit half exists (odd NW: not in the last wave)
    // the stream of the wave's two slices: uniform bases (scalar registers) + one lane offset for every load
    const char *const sb0 = reinterpret_cast<const char *>(p.stream + (size_t)(2 * wave) * NFRAG * 64);
    const char *const sb1 = reinterpret_cast<const char *>(p.stream + (size_t)(two ? 2 * wave + 1 : 2 * wave) * NFRAG * 64);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto frag = [&](const char *b, int f) -> uint4 {
        return __builtin_bit_cast(uint4, *reinterpret_cast<const u32x4 __attribute__((address_space(1))) *>(
                                             (const char __attribute__((address_space(1))) *)b + (size_t)lane16 + (size_t)f * 1024));
    };
    // resident: U_hi of r (slot 1) and of the candidate (slot 2), both halves
    u32x4 Wr[KS][2], Wg[KS][2];
    {
        const uint4 *r0 = reinterpret_cast<const uint4 *>(sb0) + lane, *r1 = reinterpret_cast<const uint4 *>(sb1) + lane;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            SLOAD2A(Wr[k][0], r0 + (size_t)(k * NF + 1) * 64, Wr[k][1], r1 + (size_t)(k * NF + 1) * 64);
            SLOAD2A(Wg[k][0], r0 + (size_t)(k * NF + 2) * 64, Wg[k][1], r1 + (size_t)(k * NF + 2) * 64);
        }
    }
    // the input-projection table [5 bases][4 kinds r, g (recurrent bias only), z, x][UP units] fp32 -> LDS rows of XT_PITCH bytes, and
    // behind it the Dense fragments (hi, lo of the wave's two slices; read when a step's logits are due)
    for (int i = tid; i < 5 * 4 * UP; i += NT)
        *reinterpret_cast<float *>(smem + p.xtab_off + (i / (4 * UP)) * XT_PITCH + (i % (4 * UP)) * 4) = p.xtab[i];
    {
        uint4 *bd = reinterpret_cast<uint4 *>(smem + p.xtab_off + 5 * XT_PITCH) + (size_t)wave * 4 * 64 + lane;
#pragma unroll
        for (int uh = 0; uh < 2; ++uh) {
            const uint4 *mypack = p.pack + (size_t)(uh == 1 && two ? 2 * wave + 1 : 2 * wave) * p.nfrag * 64 + lane;
            bd[(2 * uh) * 64] = mypack[(size_t)(3 * (KS + 1) + 1) * 64];
            bd[(2 * uh + 1) * 64] = mypack[(size_t)(3 * (KS + 1) + 2) * 64];
        }
    }
    _Float16 *const hbuf = reinterpret_cast<_Float16 *>(smem);                            // [2][32][HS] hi tiles
    _Float16 *const lbuf = reinterpret_cast<_Float16 *>(smem + p.lo_tile_off);          // [2][32][HS] lo tiles
    // carve, staged sequences, placement: wg_setup's job for 64 NW64 threads and a tile of 32 NW units
    wg_ctx ctx;
    {
        ctx.hbuf = hbuf;
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
        for (int i = tid; i < TILE; i += NT) { hbuf[i] = (_Float16)0.0f; lbuf[i] = (_Float16)0.0f; }          // h_{-1} = 0
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
    const f32x4 zero4 = { 0, 0, 0, 0 };
    const int cls = lane & 15;
    const float fbias = cls < C ? p.ffb[cls] : 0.0f;
    auto finish_step = [&](int t) {
        for (int reg = wave; reg < 4; reg += NW64) {
            const int wi = 4 * (lane >> 4) + reg;
            finish_register<NW64, MODE>(p, ctx, t, reg, fbias, ctx.rowoff[wi], ctx.row0s[wi]);
        }
    };
    // this wave's share of a step's logits (and of avg[t] for the attention pass) from the tile columns it has just written (one wave's
    // LDS operations execute in order).  The lane index is laundered: what is addressed from it here is recomputed every step instead
    // of living across the MFMA phases.
    auto dense_issue = [&](const _Float16 *hb, const _Float16 *lb, int tt) -> f32x4 {
        f32x4 d = zero4;
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int uh = 0; uh < 2; ++uh) {
            if (uh == 1 && !two) break;
            const uint4 *bdl = reinterpret_cast<const uint4 *>(smem + p.xtab_off + 5 * XT_PITCH) + (size_t)wave * 4 * 64 + ln;
            const half8 Bd_hi = __builtin_bit_cast(half8, bdl[(2 * uh) * 64]), Bd_lo = __builtin_bit_cast(half8, bdl[(2 * uh + 1) * 64]);
            const int doff = (ln & 15) * HS + 64 * wave + 32 * uh + 8 * (ln >> 4);
            const half8 a0 = *reinterpret_cast<const half8 *>(hb + doff), a1 = *reinterpret_cast<const half8 *>(hb + doff + 16 * HS);
            const half8 l0 = *reinterpret_cast<const half8 *>(lb + doff), l1 = *reinterpret_cast<const half8 *>(lb + doff + 16 * HS);
            if (MODE == 2 && (ln & 15) < ctx.nvalid)
                split_avg_store_lane(p, ctx.wg_w, tt, UP, 2 * wave + uh, ln, a0, a1, l0, l1);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, Bd_hi, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, Bd_hi, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, Bd_lo, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, Bd_lo, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(l0, Bd_hi, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(l1, Bd_hi, d, 0, 0, 0);
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

    // The ring: blocks b = uh KS + k in the order the MFMAs take them (all k-steps of half 0, then of half 1), cyclic across time steps;
    // per block the four streamed fragments: 0 hi z, 1 lo z, 2 lo r, 3 lo h (stream slots 0, 3, 4, 5).
    constexpr int D = DGRP_S64_RING, NB = 2 * KS;
    static_assert(NB % D == 0, "the ring's slot of a block must not depend on the time step");
    uint4 q[D][4];
#pragma unroll
    for (int b = 0; b < D; ++b) {
        const char *sb = b < KS ? sb0 : sb1;
        const int f0 = (b % KS) * NF;
        q[b][0] = frag(sb, f0); q[b][1] = frag(sb, f0 + 3); q[b][2] = frag(sb, f0 + 4); q[b][3] = frag(sb, f0 + 5);
    }
    // accumulators start as the table rows of the step's base (kinds 0 r, 1 g, 2 z), requested at the end of the step before
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

    // one block of the MFMA stream: the nine MFMAs of (half UH, k-step k) with the ring's reloads; HOOK(i) = what goes behind MFMA i
#if DGRP_S64_DIAG == 2
#define S64_LD(slot, f) do { } while (0)
#else
#define S64_LD(slot, f) qk[slot] = frag(nsb, nf0 + (f))
#endif
#define S64_BLOCK(UH, HOOK)                                                                                              \
    {                                                                                                                   \
        const int b_ = (UH) * KS + k, nb_ = (b_ + D) % NB;                                                              \
        uint4 (&qk)[4] = q[b_ % D];                                                                                     \
        const char *nsb = nb_ < KS ? sp0 : sp1;                                                                         \
        const int nf0 = (nb_ % KS) * NF;                                                                                \
        const u32x4 zh = __builtin_bit_cast(u32x4, qk[0]), zl = __builtin_bit_cast(u32x4, qk[1]);                         \
        const u32x4 rl = __builtin_bit_cast(u32x4, qk[2]), gl = __builtin_bit_cast(u32x4, qk[3]);                         \
        SMFMA_V(az[UH], zh, hf); HOOK(0);                                                                                \
        SMFMA_A(ar[UH], Wr[k][UH], hf); HOOK(1);                                                                         \
        SMFMA_V(az[UH], zh, lf); S64_LD(0, 0); HOOK(2);                                                                  \
        SMFMA_A(ag[UH], Wg[k][UH], hf); HOOK(3);                                                                         \
        SMFMA_A(ar[UH], Wr[k][UH], lf); HOOK(4);                                                                         \
        SMFMA_V(az[UH], zl, hf); S64_LD(1, 3); HOOK(5);                                                                  \
        SMFMA_A(ag[UH], Wg[k][UH], lf); HOOK(6);                                                                         \
        SMFMA_V(ar[UH], rl, hf); S64_LD(2, 4); HOOK(7);                                                                  \
        SMFMA_V(ag[UH], gl, hf); S64_LD(3, 5); HOOK(8);                                                                  \
    }
#define S64_NOHOOK(i) do { } while (0)
    // the gate chain of half 0's element e_ (12 links) spread over the nine gaps of a block of half 1
#define S64_G(op) split_gate_op<false, op>(gt, ar[0][e_], ag[0][e_], az[0][e_], vx[e_ & 3], hp[e_ & 3])
#define S64_GATEHOOK(i)                                                                                                  \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if constexpr ((i) == 0) { S64_G(0); S64_G(1); }                                                                 \
        else if constexpr ((i) == 1) { S64_G(2); }                                                                      \
        else if constexpr ((i) == 2) { S64_G(3); }                                                                      \
        else if constexpr ((i) == 3) { S64_G(4); }                                                                      \
        else if constexpr ((i) == 4) { S64_G(5); }                                                                      \
        else if constexpr ((i) == 5) { S64_G(6); S64_G(7); }                                                            \
        else if constexpr ((i) == 6) { S64_G(8); S64_G(9); }                                                            \
        else if constexpr ((i) == 7) { S64_G(10); }                                                                     \
        else { S64_G(11); }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)

    // state of four units of a row = their published pair, and back
    auto pair_load = [&](const _Float16 *hrow, const _Float16 *lrow, float (&hp)[4]) {
        const uint2 ph = *reinterpret_cast<const uint2 *>(hrow), pl = *reinterpret_cast<const uint2 *>(lrow);
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(hp[0]) : "v"(ph.x), "v"(pl.x));
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(hp[1]) : "v"(ph.x), "v"(pl.x));
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(hp[2]) : "v"(ph.y), "v"(pl.y));
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(hp[3]) : "v"(ph.y), "v"(pl.y));
    };
    auto pair_store = [&](_Float16 *hrow, _Float16 *lrow, const float (&h4)[4]) {
        uint2 hv, lv;
        split_hi_lo4(h4, hv, lv);
        *reinterpret_cast<uint2 *>(hrow) = hv;
        *reinterpret_cast<uint2 *>(lrow) = lv;
    };

    for (int t = 0; t < T; ++t) {
        const unsigned tab_t = tab;                            // this step's table row (the candidate's input projection is read below)
        const _Float16 *hcur = hbuf + (t & 1) * TILE, *lcur = lbuf + (t & 1) * TILE;
        _Float16 *hnxt = hbuf + ((t + 1) & 1) * TILE, *lnxt = lbuf + ((t + 1) & 1) * TILE;
        const _Float16 *arow = hcur + r * HS + 8 * khalf, *lrow = lcur + r * HS + 8 * khalf;
        // (the stream bases are laundered every step: left alone the compiler hoists a hundred fragment addresses out of the time loop)
        const char *sp0 = sb0, *sp1 = sb1;
        asm volatile("" : "+s"(sp0), "+s"(sp1));
        // own columns of the tiles: row (lane & 31), units 64 wave + 32 uh + 8 qd + 4 khalf .. + 3
        const int own = (lane & 31) * HS + 64 * wave + 4 * khalf;
        // ---- half 0: all k-steps ----
        half8 hfn = *reinterpret_cast<const half8 *>(arow), lfn = *reinterpret_cast<const half8 *>(lrow);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const half8 hf = hfn, lf = lfn;
            hfn = *reinterpret_cast<const half8 *>(arow + 16 * ((k + 1) % KS));
            lfn = *reinterpret_cast<const half8 *>(lrow + 16 * ((k + 1) % KS));
            S64_BLOCK(0, S64_NOHOOK)
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- half 1: all k-steps, with half 0's gate chains in the gaps (element e of the lane's 16 at k-step e KS / 16) ----
        {
            float hp[4], h4[4];
            f32x4 vx = zero4;
            split_gate_tmp gt;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const half8 hf = hfn, lf = lfn;
                if (k + 1 < KS) {
                    hfn = *reinterpret_cast<const half8 *>(arow + 16 * (k + 1));
                    lfn = *reinterpret_cast<const half8 *>(lrow + 16 * (k + 1));
                }
                const int e0 = (16 * k + KS - 1) / KS, e1 = (16 * (k + 1) + KS - 1) / KS;            // elements [e0, e1) belong to this k-step
                if (e0 < e1) {
                    const int e_ = e0;
                    if ((e_ & 3) == 0) {
                        pair_load(hcur + own + 8 * (e_ >> 2), lcur + own + 8 * (e_ >> 2), hp);
                        vx = ldsf4(tab_t + 3 * UP * 4 + (8 * (e_ >> 2)) * 4);
                    }
                    S64_BLOCK(1, S64_GATEHOOK)
                    h4[e_ & 3] = hp[e_ & 3];
                    if ((e_ & 3) == 3) pair_store(hnxt + own + 8 * (e_ >> 2), lnxt + own + 8 * (e_ >> 2), h4);
                } else {
                    S64_BLOCK(1, S64_NOHOOK)
                }
                // (models below 256 units have fewer k-steps than elements: the rest of this k-step's share in one go)
#pragma unroll
                for (int e_ = e0 + 1; e_ < e1; ++e_) {
                    if ((e_ & 3) == 0) {
                        pair_load(hcur + own + 8 * (e_ >> 2), lcur + own + 8 * (e_ >> 2), hp);
                        vx = ldsf4(tab_t + 3 * UP * 4 + (8 * (e_ >> 2)) * 4);
                    }
                    h4[e_ & 3] = split_gate_chain<false>(ar[0][e_], ag[0][e_], az[0][e_], vx[e_ & 3], hp[e_ & 3]);
                    if ((e_ & 3) == 3) pair_store(hnxt + own + 8 * (e_ >> 2), lnxt + own + 8 * (e_ >> 2), h4);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(az[1]), "+v"(ar[1]), "+v"(ag[1]));   // asm MFMA results -> compiler-scheduled readers
        // ---- half 1's gate math (all that is left behind the MFMAs) ----
        if (t > 0) finish_step(t - 1);
        if (two) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                float hp[4], h4[4];
                pair_load(hcur + own + 32 + 8 * qd, lcur + own + 32 + 8 * qd, hp);
                const f32x4 vx = ldsf4(tab_t + 3 * UP * 4 + (32 + 8 * qd) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    h4[i] = split_gate_chain<false>(ar[1][4 * qd + i], ag[1][4 * qd + i], az[1][4 * qd + i], vx[i], hp[i]);
                pair_store(hnxt + own + 32 + 8 * qd, lnxt + own + 32 + 8 * qd, h4);
            }
        }
        dense_store(t, dense_issue(hnxt, lnxt, t));
        if (t + 1 < T) acc_start(t + 1);
        __syncthreads();
    }
    finish_step(T - 1);
    if (MODE == 0 && p.ospan > 0) flush_image<NW64>(p, ctx);
#undef S64_BLOCK
#undef S64_GATEHOOK
#undef S64_NOHOOK
#undef S64_G
#undef S64_LD
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

// LDS carve: hi tiles (ping-pong), partial logits of (NW + 1) / 2 waves, sequences, placement, image (as much as `budget` allows),
// lo tiles, table, Dense fragments.  Sets p.ospan, p.lo_tile_off, p.xtab_off; returns the bytes.
size_t dgrp_stream64_carve(int NW, gru_params &p, int mode, int64_t s, int64_t budget)
{
    const int UP = 32 * NW, NW64 = (NW + 1) / 2;
    const int tiles = gru_lds_hbuf(UP, 8), xtab = 5 * (4 * UP * 4 + 32) + NW64 * 4 * 1024;
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

// A4 (+A5/A6 fused): the model forward of deepgrp/model.py:293-336 as one CDNA4 kernel.
//
//   x [T,5] one-hot --+--> GRU --> fwd[t] --+
//                     |                     +--> avg[t] = (fwd[t] + rc[t]) / 2 --> Dense --> Softmax
//   reverse-complement --> GRU --> rc[t]  --+      (same layer object => shared weights; rc is NOT
//                                                   re-reversed: model.py:309-312, :321-323)
//
// Work decomposition (UP = u rounded up to 32, NW = UP/32 waves per workgroup):
//   * a workgroup owns 16 consecutive windows = 32 recurrent rows (rows 0-15 the windows, rows
//     16-31 their reverse complements), i.e. one 32-row MFMA tile, for all T steps;
//   * wave w owns hidden units [32w, 32w+32) of ALL THREE gates, so z, r and the candidate of one
//     (row, unit) land in the same lane/register of three accumulators and the gate math needs no
//     cross-lane traffic;
//   * the wave's slice of the recurrent kernel U (K = UP rows x 96 columns) lives in VGPRs for the
//     whole kernel as v_mfma_f32_32x32x16_f16 fragments (112 VGPRs at u = 128) -- nothing but the
//     8 KB hidden-state tile moves per step, through LDS.  The tile is computed TRANSPOSED (weights
//     are the A operand, h the B operand: D[unit][row]), so a lane holds 4 x 4 consecutive units of one
//     row and publishes them with packed converts and four 8-byte stores.  Beyond 128 units the z slice
//     streams from L2 and the fragments of h are re-read per chain (see the loop);
//   * the input projection is one extra 16-deep k-step: one-hot(base) with a constant-1 column
//     against kernel rows and biases, each split into fp16 hi + lo parts (exact products, fp32
//     accumulate => ~fp32-exact projection for free);
//   * the candidate's  x.W_h + b  lands on top of  r * (h.U_h + b_rec)  by feeding r*g as the C
//     operand of that MFMA -- no separate accumulator;
//   * Dense(C) is four v_mfma_f32_16x16x32_f16 per wave per step on the wave's own 32 units of the
//     hidden tile (fwd rows accumulate onto rc rows = the Average), issued ONE STEP LATE behind the next
//     step's r chain; partial sums meet in LDS, every wave finishes a quarter of the 16 x C tile
//     (softmax over the class lanes by DPP) another step later and max-merges into a per-workgroup
//     LDS image of the [rows, C] output (get_max, deepgrp/maxcalc.c:10-24), flushed once at the end
//     with full-line atomic max (probabilities are >= 0, so unsigned-int max == float max).
//
// Roofline: 12 u^2 T flop per window on the MFMA pipe; what binds is the SIMD's vector issue port
// (5-6 transcendentals and ~8 plain operations per (row, unit, step)): a wave issues
// in order and waits for the matrix pipe at every back-to-back MFMA, so the gate math of one chain
// is placed in the gaps of the next chain in program order; two workgroups per CU share each SIMD.
//
// Kernels in this file:
//   gru_fused_kernel<NW,MODE,ONERCP>  fp16 MFMA operands (dgrp_model_set_precision 0; the only fused GRU kernel beyond 128 units)
//   gru_split_kernel<NW,MODE>         split operands (U and h as fp16 hi+lo pairs, three MFMA passes): the default up to 128
//                                     units where two LDS carves do not fit or below 97 units
//   gru_split2_kernel<MODE>           split operands, 97-128 units: one wave per SIMD, two row tiles per wave, U_hi and U_lo
//                                     resident (512-register budget)
//   lstm_fused_kernel<NW,MODE>        rnn = "LSTM", fp16 operands
//   attention_wave_kernel<UP,CM> (up to 64 units) / attention_row_kernel<AT,EPL,TT> (65-256 units)   second pass of attention models (after a MODE 2 pre-pass)
// MODE 0: forward + max-merge into [n, C]; 1: probabilities [nw, T, C]; 2: attention pre-pass (avg[t] spill + partial logits)
#include "gru_shared.h"
#include <mutex>
template <int NW, int MODE, bool ONERCP>
__global__ void __launch_bounds__(64 * NW, 2) gru_fused_kernel(const gru_params pin)
{
    gru_params p = pin;
    const int64_t bid = wg_record<MODE>(pin, p);
    constexpr bool PIPE = DGRP_PIPE;
    constexpr int UP = 32 * NW, KS = UP / 16, HS = UP + 8;   // HS: padded row pitch (halves) -> conflict-free b128 reads
    // u > 128: the three gate slices no longer fit 256 VGPRs; the z gate's fragments (needed last in a
    // step) are then re-read from L2 every step (16 KB per wave-step, a few % of L2 bandwidth)
    constexpr bool ZSTREAM = NW > 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

#ifdef DGRP_STAMP
    const uint64_t stamp_entry = __builtin_amdgcn_s_memtime();
    const uint64_t stamp_rentry = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, C = p.C;
    // ---- resident B fragments ------------------------------------------------------------
    const uint4 *mypack = p.pack + (size_t)wave * p.nfrag * 64 + lane;
    half8 Bz[KS + 1], Br[KS + 1], Bg[KS + 1], Bxh, Bd_hi, Bd_lo;
#pragma unroll
    for (int k = 0; k <= KS; ++k) {
        uint4 a = mypack[(size_t)(k) * 64], b = mypack[(size_t)(KS + 1 + k) * 64], c = mypack[(size_t)(2 * (KS + 1) + k) * 64];
        if (!ZSTREAM || k == KS) Bz[k] = __builtin_bit_cast(half8, a);
        Br[k] = __builtin_bit_cast(half8, b);
        Bg[k] = __builtin_bit_cast(half8, c);
    }
    Bxh = __builtin_bit_cast(half8, mypack[(size_t)(3 * (KS + 1)) * 64]);
    Bd_hi = __builtin_bit_cast(half8, mypack[(size_t)(3 * (KS + 1) + 1) * 64]);
    Bd_lo = __builtin_bit_cast(half8, mypack[(size_t)(3 * (KS + 1) + 2) * 64]);

    // ---- stage sequences, placement rows, zero state ---------------------------------------
    const wg_ctx ctx = wg_setup<NW, MODE>(p, smem, bid);
    _Float16 *const hbuf = ctx.hbuf;
    float *const dpart = ctx.dpart;
    const uint8_t *const seqs = ctx.seqs;
    const int64_t wg_w = ctx.wg_w;
    const int nvalid = ctx.nvalid;

    const int r = lane & 31;            // recurrent row of this lane's A fragment
    const int wi_a = r & 15;            // its window
    const int dir = r >> 4;             // 0: window as is, 1: reverse complement
    const int khalf = lane >> 5;        // which 8 of a 16-deep k-step this lane feeds
    const uint8_t *myseq = seqs + wi_a * p.Tp;

    f32x2 h[8];                         // fp32 master state, pairs of adjacent units (packed VALU operands)
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = ONERCP ? f32x2{ -1.0f, -1.0f } : f32x2{ 0.0f, 0.0f };   // ONERCP keeps h - 1

    _Float16 *hcur = hbuf, *hnxt = hbuf + 32 * HS;
    const f32x16 zero16 = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const f32x4 zero4 = { 0, 0, 0, 0 };
    const int cls = lane & 15;
    const float fbias = cls < C ? p.ffb[cls] : 0.0f;

    // Softmax + merge of step t's partial logits.  The 16x16 logit tile (window = 4*(lane>>4) + reg,
    // class = lane & 15) is split by accumulator register over the waves, one value per lane.
    // placement of the wave's first register (reg = wave) never changes: keep it out of the step loop
    const int pwi = 4 * (lane >> 4) + (wave & 3);
    const int p_off = ctx.rowoff[pwi];
    const int64_t p_row0 = ctx.row0s[pwi];
    auto finish_reg = [&](int t, int reg) {
        const int wi = 4 * (lane >> 4) + reg;
        finish_register<NW, MODE>(p, ctx, t, reg, fbias, reg == wave ? p_off : ctx.rowoff[wi], reg == wave ? p_row0 : ctx.row0s[wi]);
    };
    auto finish_step = [&](int t) {
        for (int reg = wave; reg < 4; reg += NW) finish_reg(t, reg);
    };
    // The wave's first register again, cut into single-instruction stages that the step loop drops into the
    // gaps of the r chain (a wave issues in order and waits for the matrix pipe at every back-to-back MFMA, so
    // what sits between two MFMAs in program order is nearly free).  Used up to 128 units; beyond, the
    // z-streaming variant has no registers to spare.
    constexpr bool STAGED = NW <= 4;
    const bool pr_on = cls < C && pwi < nvalid;
    float fs_d[NW], fs_lg = 0.0f, fs_m = 0.0f, fs_e = 0.0f, fs_s = 0.0f;
    constexpr int FS_STAGES = 12;
    auto finish_stage = [&](int st, int t) {
        switch (st) {
        case 0: {
            const float *dp = dpart + ((size_t)(t & 1) * 4 + (wave & 3)) * NW * 64 + lane;
#pragma unroll
            for (int w = 0; w < NW; ++w) fs_d[w] = dp[w * 64];
        } break;
        case 1: {
            float sum = fs_d[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) sum += fs_d[w];
            fs_lg = cls < C ? sum + fbias : -INFINITY;
        } break;
        case 2: if (MODE != 2) fs_m = row_max_ror<8>(fs_lg); break;
        case 3: if (MODE != 2) fs_m = row_max_ror<4>(fs_m); break;
        case 4: if (MODE != 2) fs_m = row_max_ror<2>(fs_m); break;
        case 5: if (MODE != 2) fs_m = row_max_ror<1>(fs_m); break;
        case 6: if (MODE != 2) fs_e = __builtin_amdgcn_exp2f(1.4426950408889634f * (fs_lg - fs_m)); break;
        case 7: if (MODE != 2) fs_s = fs_e + row_ror<8>(fs_e); break;
        case 8: if (MODE != 2) fs_s += row_ror<4>(fs_s); break;
        case 9: if (MODE != 2) fs_s += row_ror<2>(fs_s); break;
        case 10: if (MODE != 2) fs_s += row_ror<1>(fs_s); break;
        default: if (MODE != 2) fs_e *= __builtin_amdgcn_rcpf(fs_s); break;
        }
    };
    auto finish_commit = [&](int t) {
        const float val = MODE == 2 ? fs_lg : fs_e;
        if (pr_on) emit_value<MODE>(p, ctx, p_off, p_row0, t, cls, val);
        for (int reg = wave + NW; reg < 4; reg += NW) finish_reg(t, reg);      // NW < 4: the wave's other registers
    };
    // Dense on this wave's 32 units of the hidden tile `hb`: rows r (window) and r+16 (its rc) accumulate
    // (= the Average; the attention pre-pass also stores it, step tt); issue only -- the partial logits are stored by dense_store once the MFMAs are done.
    const int doff = (lane & 15) * HS + 32 * wave + 8 * (lane >> 4);
    auto dense_issue = [&](const _Float16 *hb, int tt) -> f32x4 {
        const half8 a0 = *reinterpret_cast<const half8 *>(hb + doff);
        const half8 a1 = *reinterpret_cast<const half8 *>(hb + doff + 16 * HS);
        if (MODE == 2 && (lane & 15) < nvalid) {
            // attention: keep avg[t] (fp16: it is an fp16 MFMA operand everywhere else too) for the second kernel
            const half8 av = (a0 + a1) * (_Float16)0.5f;
            *reinterpret_cast<half8 *>(reinterpret_cast<_Float16 *>(p.avg) + ((wg_w + (lane & 15) - p.w0 + p.avgw) * (int64_t)T + tt) * UP + 32 * wave + 8 * (lane >> 4)) = av;
        }
        f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, Bd_hi, zero4, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, Bd_hi, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, Bd_lo, d, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, Bd_lo, d, 0, 0, 0);
    };
    auto dense_store = [&](int t, const f32x4 &d) {
        float *dw = dpart + ((size_t)(t & 1) * 4 * NW + wave) * 64 + lane;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) dw[reg * NW * 64] = d[reg];
    };

#ifdef DGRP_STAMP
    uint32_t stamp_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    uint64_t stamp_prev = __builtin_amdgcn_s_memtime();
    const uint64_t stamp_t0 = stamp_prev, stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // Step t, in program order (the matrix pipe runs behind the wave's instruction stream, so whatever is
    // issued after a group of MFMAs executes in their shadow):
    //   loads of h_{t-1} | r chain + Dense(t-1) MFMAs | softmax/merge of step t-2 | g chain + sigmoid(r) |
    //   z chain + r*g, candidate input projection, tanh | store Dense(t-1) partials | sigmoid(z), blend,
    //   publish h_t | barrier
    for (int t = 0; t < T; ++t) {
        STAMP(0);
        // ---- A operand of the input k-step: one-hot(base) | 1 ------------------------------
        uint32_t b = myseq[dir ? T - 1 - t : t];
        if (dir) b = b < 4 ? 3 - b : 4;                      // complement table [3,2,1,0,4], model.py:233-237
        const uint32_t one = 0x3C00u << ((b & 1) * 16);
        const uint32_t sel = b >> 1;
        const uint4 xu = make_uint4(sel == 0 ? one : 0u, sel == 1 ? one : 0u, (sel == 2 ? one : 0u) | 0x3C000000u, 0u);
        const half8 xa = __builtin_bit_cast(half8, xu);

        const _Float16 *arow = hcur + r * HS + 8 * khalf;
        // u <= 128: the KS fragments of h_{t-1} are read once and serve all three chains.  Beyond, they alone
        // would take 64+ VGPRs next to two resident gate slices: every chain re-reads them from LDS instead
        // (3 x 16 KB per wave-step, well inside the LDS rate next to 51 MFMAs).
        half8 af[ZSTREAM ? 1 : KS];
        if (!ZSTREAM) {
#pragma unroll
            for (int k = 0; k < KS; ++k) af[k] = *reinterpret_cast<const half8 *>(arow + 16 * k);
        }
        auto hfrag = [&](int k) -> half8 { return ZSTREAM ? *reinterpret_cast<const half8 *>(arow + 16 * k) : af[ZSTREAM ? 0 : k]; };
        // the streamed z fragments: ZPF of them in flight, the first ones requested before the g chain (same box, u=256:
        // 8 deep 157 ms, 4 deep 116, 2 deep 105, 1 deep 104 per 5 Mbp -- spilled registers cost more than exposed latency)
#ifndef DGRP_ZPF
#define DGRP_ZPF (NW >= 7 ? 2 : 4)
#endif
        constexpr int ZPF = DGRP_ZPF;                        // 224+ units spill: every register counts more than latency
        uint4 zq[ZSTREAM ? ZPF : 1];
        if (ZSTREAM) {
#pragma unroll
            for (int i = 0; i < ZPF; ++i) zq[i] = mypack[(size_t)i * 64];
        }
      if (ZSTREAM) {
        // Beyond 128 units: the r and the (streamed) z chain run interleaved k-step by k-step, so one z fragment is
        // consumed per TWO MFMAs and a fragment requested ZPF k-steps ahead has twice the time to arrive from L2; the
        // hidden tile's fragments are read once for both.  Then the g chain with both sigmoids in its gaps.
        f32x16 ar = __builtin_amdgcn_mfma_f32_32x32x16_f16(Br[KS], xa, zero16, 0, 0, 0);
        f32x16 az = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bz[KS], xa, zero16, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const half8 hf = hfrag(k);
            const half8 bz = __builtin_bit_cast(half8, zq[ZSTREAM ? k % ZPF : 0]);
            if (k + ZPF < KS) zq[ZSTREAM ? k % ZPF : 0] = mypack[(size_t)(k + ZPF) * 64];
            ar = __builtin_amdgcn_mfma_f32_32x32x16_f16(Br[k], hf, ar, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_32x32x16_f16(bz, hf, az, 0, 0, 0);
        }
        f32x4 dpl = zero4;
        if (t > 0) dpl = dense_issue(hcur, t - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (t > 1) finish_step(t - 2);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bg[KS], xa, zero16, 0, 0, 0);
        f32x2 rr[8], zz[8];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bg[k], hfrag(k), ag, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 2 * (k * 8 / KS); i < 2 * ((k + 1) * 8 / KS); i += 2) {
                rr[i / 2] = rcp1p_exp2_pair(ar[i], ar[i + 1]);
                zz[i / 2] = rcp1p_exp2_pair(az[i], az[i + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 16; i += 2) {                                          // r * (h.U_h + b_rec_h)
            const f32x2 pr = f32x2{ ag[i], ag[i + 1] } * rr[i / 2];
            ag[i] = pr.x; ag[i + 1] = pr.y;
        }
        ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bxh, xa, ag, 0, 0, 0);            // + x.W_h + b_in_h
        if (t > 0) dense_store(t - 1, dpl);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x2 hh = 1.0f - 2.0f * rcp1p_exp2_pair(ag[2 * i], ag[2 * i + 1]);
            h[i] = hh + zz[i] * (h[i] - hh);                                         // z*h + (1-z)*hh
        }
      } else {
        f32x16 ar = __builtin_amdgcn_mfma_f32_32x32x16_f16(Br[KS], xa, zero16, 0, 0, 0);
        if (STAGED) finish_stage(0, t);                      // (steps 0 and 1 run the stages on stale data, uncommitted)
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            ar = __builtin_amdgcn_mfma_f32_32x32x16_f16(Br[k], hfrag(k), ar, 0, 0, 0);
            if (STAGED) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int st = 1 + k * (FS_STAGES - 1) / KS; st < 1 + (k + 1) * (FS_STAGES - 1) / KS; ++st) finish_stage(st, t);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        f32x4 dpl = zero4;
        if (t > 0) dpl = dense_issue(hcur, t - 1);
        if (PIPE) __builtin_amdgcn_sched_barrier(0);
        STAMP(1);
        if (t > 1) {
            if (STAGED) finish_commit(t - 2);
            else finish_step(t - 2);
        }
        if (PIPE) __builtin_amdgcn_sched_barrier(0);
        STAMP(2);
        f32x16 ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bg[KS], xa, zero16, 0, 0, 0);
        f32x2 rr[8];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bg[k], hfrag(k), ag, 0, 0, 0);
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 2 * (k * 8 / KS); i < 2 * ((k + 1) * 8 / KS); i += 2) {
                rr[i / 2] = rcp1p_exp2_pair(ar[i], ar[i + 1]);
            }
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(3);
        f32x16 az = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bz[KS], xa, zero16, 0, 0, 0);
        // z chain: its first half hides r * g, then the candidate's input projection is issued and the
        // second half hides the tanh
        constexpr int KH = KS / 2;
        f32x2 hh[8];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const half8 bz = ZSTREAM ? __builtin_bit_cast(half8, zq[ZSTREAM ? k % ZPF : 0]) : Bz[ZSTREAM ? KS : k];
            if (ZSTREAM && k + ZPF < KS) zq[ZSTREAM ? k % ZPF : 0] = mypack[(size_t)(k + ZPF) * 64];
            az = __builtin_amdgcn_mfma_f32_32x32x16_f16(bz, hfrag(k), az, 0, 0, 0);
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
            if (k < KH) {
#pragma unroll
                for (int i = 2 * (k * 8 / KH); i < 2 * ((k + 1) * 8 / KH); i += 2) {      // r * (h.U_h + b_rec_h)
                    const f32x2 pr = f32x2{ ag[i], ag[i + 1] } * rr[i / 2];
                    ag[i] = pr.x; ag[i + 1] = pr.y;
                }
                if (k == KH - 1) ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bxh, xa, ag, 0, 0, 0);    // + x.W_h + b_in_h
            } else {
#pragma unroll
                for (int i = 2 * ((k - KH) * 8 / (KS - KH)); i < 2 * ((k - KH + 1) * 8 / (KS - KH)); i += 2) {
                    if (ONERCP) hh[i / 2] = f32x2{ __builtin_amdgcn_exp2f(ag[i]), __builtin_amdgcn_exp2f(ag[i + 1]) } + 1.0f;   // A = 1 + 2^ag
                    else hh[i / 2] = 1.0f - 2.0f * rcp1p_exp2_pair(ag[i], ag[i + 1]);
                }
            }
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
        }
        if (t > 0) dense_store(t - 1, dpl);
        STAMP(4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (ONERCP) {
                // h' - 1 = [ (h - 1)(1 + Eg) - 2 Ez ] / [ (1 + Eg)(1 + Ez) ],  Eg = 2^ag (tanh), Ez = 2^az (sigmoid):
                // one reciprocal for both gates; the packed z bias carries +1, so the exponential below is 2 Ez.
                // The model constructor proved (1 + Eg)(1 + Ez) finite (api.hip).
                const f32x2 e2 = { __builtin_amdgcn_exp2f(az[2 * i]), __builtin_amdgcn_exp2f(az[2 * i + 1]) };
                const f32x2 d = hh[i] * (0.5f * e2 + 1.0f);
                const f32x2 n = h[i] * hh[i] - e2;
                h[i] = n * f32x2{ __builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y) };
            } else {
                const f32x2 z = rcp1p_exp2_pair(az[2 * i], az[2 * i + 1]);
                h[i] = hh[i] + z * (h[i] - hh[i]);                             // z*h + (1-z)*hh
            }
        }
      }
        // ---- publish h_t (fp16) for the next step's B operand: a lane holds 4 x 4 consecutive units of
        // one row (transposed tile), i.e. four 8-byte stores
        _Float16 *wrow = hnxt + (lane & 31) * HS + 32 * wave + 4 * khalf;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 h0 = ONERCP ? h[2 * q] + 1.0f : h[2 * q], h1 = ONERCP ? h[2 * q + 1] + 1.0f : h[2 * q + 1];
            const half4 hv = { (_Float16)h0.x, (_Float16)h0.y, (_Float16)h1.x, (_Float16)h1.y };
            *reinterpret_cast<half4 *>(wrow + 8 * q) = hv;
        }
        STAMP(5);
        __syncthreads();
        STAMP(6);
        _Float16 *tmp = hcur; hcur = hnxt; hnxt = tmp;
    }
#ifdef DGRP_STAMP
    if (p.stamps && lane == 0) {
        uint64_t *o = p.stamps + ((size_t)blockIdx.x * NW + wave) * 16;
        for (int k = 0; k < 8; ++k) o[k] = stamp_acc[k];
        o[8] = __builtin_amdgcn_s_memtime() - stamp_t0;
        o[9] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
        o[10] = stamp_t0 - stamp_entry;
        o[11] = __builtin_amdgcn_s_getreg(6148 | (11 << 11)) ;   // HW_ID low 12 bits
    }
#endif
    // drain: Dense of the last step, the two outstanding softmax/merge steps
    {
        const f32x4 dpl = dense_issue(hcur, T - 1);
        if (T > 1) finish_step(T - 2);
        dense_store(T - 1, dpl);
        __syncthreads();
        finish_step(T - 1);
    }

    if (MODE == 0 && p.ospan > 0) flush_image<NW>(p, ctx);
#ifdef DGRP_STAMP
    if (p.stamps && lane == 0) {
        uint64_t *o = p.stamps + ((size_t)blockIdx.x * NW + wave) * 16;
        o[11] = __builtin_amdgcn_s_memtime() - stamp_entry;
        o[12] = stamp_rentry;
        o[13] = __builtin_amdgcn_s_memrealtime();
        o[14] = __builtin_amdgcn_s_getreg((15 << 11) | 4);     // HW_ID[15:0]
    }
#endif
}

// ---- split-operand variant (dgrp_model_set_precision(m, 1)) -------------------------------------------------------
// Same decomposition, but both MFMA operands of the recurrent contraction carry fp32-grade precision as fp16 pairs:
// U = U_hi + U_lo (packed once), h_{t-1} = h_hi + h_lo (two LDS tiles), and  U.h ~ U_hi.h_hi + U_hi.h_lo + U_lo.h_hi
// (the dropped U_lo.h_lo term is below 2^-22 of the product).  Three MFMAs per k-step instead of one: the matrix pipe,
// half idle in the fast kernel, becomes the bound.  U_hi stays resident in VGPRs; the U_lo fragments stream from L2
// every step through a small register ring in consumption order (k-step major, gates r, g, z); the hidden tile's two
// fragments are read from LDS once per k-step.  Dense takes the lo tile too.  Written for correctness first: no staging of the softmax into
// MFMA gaps, two reciprocals.  GRU up to 128 units (MODE 2 = the attention pre-pass).
template <int NW, int MODE, bool ONERCP>
__global__ void __launch_bounds__(64 * NW, 2) gru_split_kernel(const gru_params pin)
{
    gru_params p = pin;
    const int64_t bid = wg_record<MODE>(pin, p);
#ifndef DGRP_SPLIT_PF
#define DGRP_SPLIT_PF 6
#endif
#ifndef DGRP_SPLIT_PIN
#define DGRP_SPLIT_PIN 1
#endif
    constexpr int UP = 32 * NW, KS = UP / 16, HS = UP + 8, NLO = 3 * KS, PF = DGRP_SPLIT_PF < NLO ? DGRP_SPLIT_PF : NLO;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, C = p.C;
    const uint4 *mypack = p.pack + (size_t)wave * p.nfrag * 64 + lane;
    const uint4 *mylo = p.pack_lo + (size_t)wave * NLO * 64 + lane;
    half8 Bz[KS + 1], Br[KS + 1], Bg[KS + 1], Bxh, Bd_hi, Bd_lo;
#pragma unroll
    for (int k = 0; k <= KS; ++k) {
        Bz[k] = __builtin_bit_cast(half8, mypack[(size_t)(k) * 64]);
        Br[k] = __builtin_bit_cast(half8, mypack[(size_t)(KS + 1 + k) * 64]);
        Bg[k] = __builtin_bit_cast(half8, mypack[(size_t)(2 * (KS + 1) + k) * 64]);
    }
    Bxh = __builtin_bit_cast(half8, mypack[(size_t)(3 * (KS + 1)) * 64]);
    Bd_hi = __builtin_bit_cast(half8, mypack[(size_t)(3 * (KS + 1) + 1) * 64]);
    Bd_lo = __builtin_bit_cast(half8, mypack[(size_t)(3 * (KS + 1) + 2) * 64]);

    _Float16 *const lbuf = reinterpret_cast<_Float16 *>(smem + p.lo_tile_off);          // [2][32][HS] lo tiles
    for (int i = tid; i < 32 * HS; i += 64 * NW) lbuf[i] = (_Float16)0.0f;
    const wg_ctx ctx = wg_setup<NW, MODE>(p, smem, bid);                                  // ends with a barrier
    float *const dpart = ctx.dpart;

    const int r = lane & 31, wi_a = r & 15, dir = r >> 4, khalf = lane >> 5;
    const uint8_t *myseq = ctx.seqs + wi_a * p.Tp;
    float h[16];                                              // gate state: h, or h - 1 (ONERCP), see split_gate_op
#pragma unroll
    for (int i = 0; i < 16; ++i) h[i] = ONERCP ? -1.0f : 0.0f;
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
        if (dir) b = b < 4 ? 3 - b : 4;
        const uint32_t one = 0x3C00u << ((b & 1) * 16);
        const uint32_t sel = b >> 1;
        const uint4 xu = make_uint4(sel == 0 ? one : 0u, sel == 1 ? one : 0u, (sel == 2 ? one : 0u) | 0x3C000000u, 0u);
        const half8 xa = __builtin_bit_cast(half8, xu);
        const _Float16 *arow = hcur + r * HS + 8 * khalf, *lrow = lcur + r * HS + 8 * khalf;

        // All three contractions advance together, k-step by k-step: the two fragments of the hidden tile are read once
        // per k-step and serve nine MFMAs on three independent accumulators (no dependent back-to-back issue), one
        // lo weight fragment per gate arrives from the ring.  The lo stream is packed in this order: f = 3 k + gate,
        // gates r, g, z.
        uint4 q[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) q[i] = mylo[(size_t)i * 64];
        f32x16 ar = __builtin_amdgcn_mfma_f32_32x32x16_f16(Br[KS], xa, zero16, 0, 0, 0);
        f32x16 ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bg[KS], xa, zero16, 0, 0, 0);
        f32x16 az = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bz[KS], xa, zero16, 0, 0, 0);
        const f32x16 ax = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bxh, xa, zero16, 0, 0, 0);   // the candidate's input projection
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const half8 hf = *reinterpret_cast<const half8 *>(arow + 16 * k);
            const half8 lf = *reinterpret_cast<const half8 *>(lrow + 16 * k);
            half8 wl[3];
#pragma unroll
            for (int gi = 0; gi < 3; ++gi) {
                const int f = 3 * k + gi;
                wl[gi] = __builtin_bit_cast(half8, q[f % PF]);
                if (f + PF < NLO) q[f % PF] = mylo[(size_t)(f + PF) * 64];
            }
            // Keep the refills HERE for the attention pre-pass: left alone the scheduler sinks each load to two MFMAs in front of its
            // use (the ring then hides nothing: 2 500 of a wave-step's 6 700 cycles waiting on vmcnt).  defaults.toml shape +4.7 %;
            // without the avg[t] stores in the step (MODE 0 / 1) the sunk form is 1 % faster at 64 units and stays.
            if (DGRP_SPLIT_PIN && (MODE == 2 || DGRP_SPLIT_PIN == 2)) __builtin_amdgcn_sched_barrier(0);
            ar = __builtin_amdgcn_mfma_f32_32x32x16_f16(Br[k], hf, ar, 0, 0, 0);
            ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bg[k], hf, ag, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bz[k], hf, az, 0, 0, 0);
            ar = __builtin_amdgcn_mfma_f32_32x32x16_f16(Br[k], lf, ar, 0, 0, 0);
            ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bg[k], lf, ag, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bz[k], lf, az, 0, 0, 0);
            ar = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[0], hf, ar, 0, 0, 0);
            ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[1], hf, ag, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[2], hf, az, 0, 0, 0);
        }
        f32x4 dpl = zero4;
        if (t > 0) dpl = dense_issue(hcur, lcur, t - 1);
        if (t > 1) finish_step(t - 2);
        if (t > 0) dense_store(t - 1, dpl);
#pragma unroll
        for (int i = 0; i < 16; ++i) h[i] = split_gate_chain<ONERCP>(ar[i], ag[i], az[i], ax[i], h[i]);
        // publish h_t as an fp16 pair: hi = fp16(h), lo = fp16(h - hi)
        _Float16 *wrow = hnxt + (lane & 31) * HS + 32 * wave + 4 * khalf;
        _Float16 *wlow = lnxt + (lane & 31) * HS + 32 * wave + 4 * khalf;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float h4[4] = { split_state_h<ONERCP>(h[4 * qd]), split_state_h<ONERCP>(h[4 * qd + 1]), split_state_h<ONERCP>(h[4 * qd + 2]),
                                  split_state_h<ONERCP>(h[4 * qd + 3]) };
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

// ------------------------------------------------------------------------------------------
// rnn = "LSTM" (deepgrp/model.py:219-223): same decomposition as the GRU kernel -- a workgroup owns 16
// windows (32 rows), wave w owns units [32w, 32w+32) of all FOUR gates (i|f|c|o), fp32 cell and hidden
// state in registers.  Four gate slices do not fit 256 VGPRs beyond 64 units, so from NW = 3 on the f
// and o fragments are re-read from L2 every step (the i and c fragments stay resident).
// Fragment order per wave: gate g, k-step k -> g*(KS+1)+k (k = KS: input projection + bias, hi|lo),
// then dense hi, dense lo.  Scales folded in: -log2 e for i, f, o; 2 log2 e for c.
// ------------------------------------------------------------------------------------------
template <int NW, int MODE>
__global__ void __launch_bounds__(64 * NW, 2) lstm_fused_kernel(const gru_params pin)
{
    gru_params p = pin;
    const int64_t bid = wg_record<MODE>(pin, p);
    constexpr int UP = 32 * NW, KS = UP / 16, HS = UP + 8;
    constexpr bool STREAM = NW > 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, C = p.C;

    const uint4 *mypack = p.pack + (size_t)wave * p.nfrag * 64 + lane;
    half8 Bi[KS + 1], Bf[KS + 1], Bc[KS + 1], Bo[KS + 1], Bd_hi, Bd_lo;
#pragma unroll
    for (int k = 0; k <= KS; ++k) {
        Bi[k] = __builtin_bit_cast(half8, mypack[(size_t)(0 * (KS + 1) + k) * 64]);
        Bc[k] = __builtin_bit_cast(half8, mypack[(size_t)(2 * (KS + 1) + k) * 64]);
        if (!STREAM || k == KS) {
            Bf[k] = __builtin_bit_cast(half8, mypack[(size_t)(1 * (KS + 1) + k) * 64]);
            Bo[k] = __builtin_bit_cast(half8, mypack[(size_t)(3 * (KS + 1) + k) * 64]);
        }
    }
    Bd_hi = __builtin_bit_cast(half8, mypack[(size_t)(4 * (KS + 1)) * 64]);
    Bd_lo = __builtin_bit_cast(half8, mypack[(size_t)(4 * (KS + 1) + 1) * 64]);

    const wg_ctx ctx = wg_setup<NW, MODE>(p, smem, bid);
    _Float16 *const hbuf = ctx.hbuf;
    float *const dpart = ctx.dpart;
    const uint8_t *const seqs = ctx.seqs;

    const int r = lane & 31, wi_a = r & 15, dir = r >> 4, khalf = lane >> 5;
    const uint8_t *myseq = seqs + wi_a * p.Tp;
    float h[16], c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { h[i] = 0.0f; c[i] = 0.0f; }
    _Float16 *hcur = hbuf, *hnxt = hbuf + 32 * HS;
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

    for (int t = 0; t < T; ++t) {
        if (t > 0) finish_step(t - 1);
        uint32_t b = myseq[dir ? T - 1 - t : t];
        if (dir) b = b < 4 ? 3 - b : 4;
        const uint32_t one = 0x3C00u << ((b & 1) * 16);
        const uint32_t sel = b >> 1;
        const uint4 xu = make_uint4(sel == 0 ? one : 0u, sel == 1 ? one : 0u, (sel == 2 ? one : 0u) | 0x3C000000u, 0u);
        const half8 xa = __builtin_bit_cast(half8, xu);
        f32x16 ai = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa, Bi[KS], zero16, 0, 0, 0);
        f32x16 af = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa, Bf[KS], zero16, 0, 0, 0);
        f32x16 ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa, Bc[KS], zero16, 0, 0, 0);
        f32x16 ao = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa, Bo[KS], zero16, 0, 0, 0);
        const _Float16 *arow = hcur + r * HS + 8 * khalf;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const half8 a = *reinterpret_cast<const half8 *>(arow + 16 * k);
            const half8 bf = STREAM ? __builtin_bit_cast(half8, mypack[(size_t)(1 * (KS + 1) + k) * 64]) : Bf[k];
            const half8 bo = STREAM ? __builtin_bit_cast(half8, mypack[(size_t)(3 * (KS + 1) + k) * 64]) : Bo[k];
            ai = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, Bi[k], ai, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, Bc[k], ac, 0, 0, 0);
            af = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bf, af, 0, 0, 0);
            ao = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bo, ao, 0, 0, 0);
        }
        // c = f*c + i*tanh(z_c) ; h = o*tanh(c)     (accumulators are in the exp2 domain)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float ig = sigmoid_from_scaled(ai[i]), fg = sigmoid_from_scaled(af[i]);
            c[i] = fg * c[i] + ig * tanh_from_scaled(ac[i]);
            h[i] = sigmoid_from_scaled(ao[i]) * fast_tanh(c[i]);
        }
        _Float16 *wcol = hnxt + 32 * wave + (lane & 31);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * khalf;
            wcol[row * HS] = (_Float16)h[i];
        }
        {
            const _Float16 *drow = hnxt + (lane & 15) * HS + 32 * wave + 8 * (lane >> 4);
            const half8 a0 = *reinterpret_cast<const half8 *>(drow);
            const half8 a1 = *reinterpret_cast<const half8 *>(drow + 16 * HS);
            f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, Bd_hi, zero4, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, Bd_hi, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, Bd_lo, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, Bd_lo, d, 0, 0, 0);
            float *dw = dpart + ((size_t)(t & 1) * 4 * NW + wave) * 64 + lane;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) dw[reg * NW * 64] = d[reg];
        }
        __syncthreads();
        _Float16 *tmp = hcur; hcur = hnxt; hnxt = tmp;
    }
    finish_step(T - 1);
    if (MODE == 0 && p.ospan > 0) flush_image<NW>(p, ctx);
}

// ------------------------------------------------------------------------------------------
// AdditiveAttention(use_scale=True) + Dense + Softmax for the attention model
// (model.py:309-319, :325-329).  One workgroup per window over the avg[t] tile kept by mode 2.
//   q = avg-of-final-states = avg[T-1];  e[t] = sum_k scale[k] tanh(q[k] + avg[t,k]);
//   a = softmax_t(e);  ctx = sum_t a[t] avg[t];  logits[t] = ctx.W_top + (avg[t].W_bot + b)
// ------------------------------------------------------------------------------------------
struct att_params {
    const void *avg;     // [nw, T, UP] fp16, or fp32 behind a split-operand pre-pass (the kernels' AT)
    const float *pl;     // [nw, T, C]  avg[t].W_bot + b  from the GRU kernel
    const float *scale;  // [UP]
    const float *wtop;   // [UP, 16]
    float *out;
    int64_t n, s, w0, nw;
    dgrp_placement place;
    int T, C, UP, merge;
    int ospan;           // wave kernel, merge: rows of the workgroup's LDS output image
    // batched records: window wl belongs to record r with recs[r].win_first <= wl; placement, first output row and
    // row limit are the record's own
    const gru_rec *recs;
    int64_t nrec;
};

// first merged row of (global) window wl and the row limit of its record
__device__ __forceinline__ int64_t att_window_row(const att_params &p, int64_t wl, int64_t *limit)
{
    if (!p.recs) {
        *limit = p.n;
        return dgrp_place_row(p.place, p.w0 + wl, p.s);
    }
    int64_t lo = 0, hi = p.nrec;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (p.recs[mid].win_first <= wl) lo = mid; else hi = mid;
    }
    const gru_rec rc = p.recs[lo];
    *limit = rc.out_row + rc.n;
    return rc.out_row + dgrp_place_row(rc.place, wl - rc.win_first, p.s);
}

typedef unsigned att_chunk __attribute__((ext_vector_type(4)));      // 16 bytes as a register vector (HIP's uint4 is a struct)
#define ATT_WPB 16                                       // windows per workgroup of the wave kernel below
// ------------------------------------------------------------------------------------------
// Models up to 64 units: ONE WAVE per window and no workgroup barrier.  Here
// lane <-> time step throughout: a lane reads its own avg[t] row (UP halves, contiguous) into registers, scores
// it against q (packed math, two units per instruction; 2^(c(q+a)) with c folded into the LDS copy of q, and
// tanh = 1 - 2r folded into the scale: e[t] = sum(scale) - 2 sum_k scale[k] r[k]), keeps the softmax over t
// online, and accumulates its OWN share p[t] avg[t,:] of the context in registers; the 64 per-lane partial
// contexts meet once per window in a 6-stage halving butterfly (after stage xor-M a lane keeps the half of the
// units whose bit M equals its own lane bit) that leaves unit k's total in lane k.
// ------------------------------------------------------------------------------------------
// whole-wave reductions without LDS round trips: DPP rotations inside the four 16-lane rows, then the four row
// results through v_readlane
__device__ __forceinline__ float lane_value(float x, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}
__device__ __forceinline__ float wave_allmax(float x)
{
    x = row_allmax(x);
    const float a = lane_value(x, 0), b = lane_value(x, 16), c = lane_value(x, 32), d = lane_value(x, 48);
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ float wave_allsum(float x)
{
    x = row_allsum(x);
    const float a = lane_value(x, 0), b = lane_value(x, 16), c = lane_value(x, 32), d = lane_value(x, 48);
    return (a + b) + (c + d);
}

// ------------------------------------------------------------------------------------------
// Models of 65-256 units: ONE WAVE per window, no LDS, no barrier (the workgroup-per-window kernel this replaces staged 64-step
// tiles in LDS behind four barriers each and read its fp32 spill at 2.5 TB/s; this one: 4.4 TB/s).  lane <-> EPL consecutive units throughout
// (2 up to 128 units, 4 beyond), so a step's row is ONE coalesced load per lane and q, the scale and the context are per-lane
// registers.  A tile = TT steps (8 or 4) held in registers; small tiles keep the kernel at 3-4 waves per SIMD, which cover each
// other's loads (tiles of 16 steps: 12.4 ms per 400 k windows of 128 units, of 8: 9.3 ms; 32 with the next tile prefetched: spills):
//   scores    every lane's share  sum_j s2[j] r[j]  of each step (r = 1 / (1 + 2^(c (q + a))), tanh = 1 - 2 r folded into s2 = -2 scale);
//             the TT per-lane shares meet in a halving butterfly (after the stage of partner lane ^ M a lane keeps the upper half of
//             its values if its bit M is set, the lower half otherwise, each summed with the partner's): log2(TT) stages of
//             TT/2, TT/4, .. 1 exchanges leave step  lane >> (6 - log2 TT)  of the tile in every lane, complete after 6 - log2 TT
//             more pair sums;
//   softmax   over t kept online across tiles (running max / sum, context rescaled per tile), wave reductions by DPP + readlane;
//   context   ctx[j] += p[t] a[t, j] with p[t] read from the lane that holds it (v_readlane: a scalar operand of the fma).
// Then ctx . W_top by lane shares + wave sums; the output rows of the workgroup's four windows pre-merge in an LDS image (they
// overlap by T - s rows each) that is flushed with one atomic per non-zero entry (u = 128: +2 %, with the fp16 spill +7 %).
// ------------------------------------------------------------------------------------------
template <typename AT, int EPL, int TT>
__global__ void __launch_bounds__(256, 3) attention_row_kernel(const att_params p)
{
    constexpr int LT = TT == 32 ? 5 : TT == 16 ? 4 : TT == 8 ? 3 : 2, SH = 6 - LT;    // a tile's step t ends up in the 2^SH lanes t << SH ..
    struct alignas(sizeof(AT) * EPL) row_t { AT v[EPL]; };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wl_ = (int64_t)blockIdx.x * 4 + wave;
    const bool alive = wl_ < p.nw;                             // (a wave without a window walks the last one and emits nothing)
    const int64_t wl = alive ? wl_ : p.nw - 1;
    const int T = p.T, UP = p.UP, C = p.C;
    // merge: the rows the workgroup's four windows cover pre-merge in an LDS image (they overlap by T - s rows each), flushed once
    extern __shared__ __attribute__((aligned(16))) unsigned char att_row_dyn[];
    unsigned *obuf = reinterpret_cast<unsigned *>(att_row_dyn);
    int64_t img_lo = 0;
    const bool image = p.merge && p.ospan > 0 && !p.recs;
    if (image) {
        const int64_t w_first = (int64_t)blockIdx.x * 4, w_last = min(w_first + 3, p.nw - 1);
        const int64_t a = dgrp_place_row(p.place, p.w0 + w_first, p.s), b = dgrp_place_row(p.place, p.w0 + w_last, p.s);
        img_lo = a < b ? a : b;
        for (int i = threadIdx.x; i < p.ospan * C; i += 256) obuf[i] = 0u;
        __syncthreads();
    }
    const int k0 = lane * EPL;
    const bool on = k0 < UP;                                  // UP / EPL lanes carry units
    const AT *avg = reinterpret_cast<const AT *>(p.avg) + wl * (int64_t)T * UP + (on ? k0 : 0);
    const float cexp = 2.8853900817779268f;                   // tanh(x) = 1 - 2 / (1 + 2^(c x))
    float cq[EPL], s2[EPL], ctx[EPL];
    float ssum = 0.0f;
    {
        const row_t qr = *reinterpret_cast<const row_t *>(avg + (int64_t)(T - 1) * UP);          // Average of the two final states = avg[T-1]
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const float sc = on ? p.scale[k0 + j] : 0.0f;     // idle lanes read lane 0's units and weigh them with 0
            cq[j] = cexp * (float)qr.v[j];
            s2[j] = -2.0f * sc;
            ssum += sc;
            ctx[j] = 0.0f;
        }
    }
    ssum = wave_allsum(ssum);                                 // e[t] = sum(scale) - 2 sum_k scale[k] r[t, k]
    float run_m = -INFINITY, run_l = 0.0f;
    for (int t0 = 0; t0 < T; t0 += TT) {
        row_t rows[TT];
#pragma unroll
        for (int i = 0; i < TT; ++i)                          // (steps behind the window's end: its last row again, weight 0 below)
            rows[i] = *reinterpret_cast<const row_t *>(avg + (int64_t)min(t0 + i, T - 1) * UP);
        float ep[TT];
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf((float)rows[i].v[j], cexp, cq[j])));
                acc = __builtin_fmaf(s2[j], r, acc);
            }
            ep[i] = acc;
        }
        // halving butterfly over the lanes: TT values per lane -> one
#pragma unroll
        for (int n = TT / 2, M = 32; n >= 1; n >>= 1, M >>= 1) {
            const bool up = (lane & M) != 0;
#pragma unroll
            for (int i = 0; i < n; ++i) {
                const float send = up ? ep[i] : ep[i + n], keep = up ? ep[i + n] : ep[i];
                ep[i] = keep + __shfl_xor(send, M);
            }
        }
        float e = ep[0];
#pragma unroll
        for (int M = (1 << SH) >> 1; M >= 1; M >>= 1) e += __shfl_xor(e, M);
        const int tl = lane >> SH;                            // the step of the tile this lane holds
        const bool valid = t0 + tl < T;
        e = valid ? ssum + e : -INFINITY;
        const float new_m = fmaxf(run_m, wave_allmax(e));
        const float alpha = __builtin_amdgcn_exp2f(1.4426950408889634f * (run_m - new_m));     // 0 on the first tile (run_m = -inf)
        const float pt = valid ? __builtin_amdgcn_exp2f(1.4426950408889634f * (e - new_m)) : 0.0f;
        run_l = run_l * alpha + wave_allsum((lane & ((1 << SH) - 1)) == 0 ? pt : 0.0f);       // a step sits in 2^SH lanes: count it once
        run_m = new_m;
#pragma unroll
        for (int j = 0; j < EPL; ++j) ctx[j] *= alpha;
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            const float w = lane_value(pt, i << SH);
#pragma unroll
            for (int j = 0; j < EPL; ++j) ctx[j] = __builtin_fmaf(w, (float)rows[i].v[j], ctx[j]);
        }
    }
    // ctx . W_top (rows of W_top: this lane's units), normalised; the same value in every lane
    const float inv = 1.0f / run_l;
    float cl[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float part = 0.0f;
        if (c < C && on) {
#pragma unroll
            for (int j = 0; j < EPL; ++j) part = __builtin_fmaf(ctx[j], p.wtop[(k0 + j) * 16 + c], part);
        }
        cl[c] = c < C ? wave_allsum(part) * inv : 0.0f;
    }
    int64_t limit = p.n;
    const int64_t row0 = p.merge ? att_window_row(p, wl, &limit) : wl * (int64_t)T;
    const int64_t ioff = row0 - img_lo;                       // the window's first row in the image, if it lies inside
    const bool in_image = image && ioff >= 0 && ioff + T <= p.ospan;
    const float *pl = p.pl + wl * (int64_t)T * C;
    for (int t = lane; t < T && alive; t += 64) {
        float lg[16];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < C) { lg[c] = pl[t * C + c] + cl[c]; mx = fmaxf(mx, lg[c]); }
        float den = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < C) { lg[c] = __expf(lg[c] - mx); den += lg[c]; }
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < C) {
                const float v = lg[c] / den;
                if (in_image) {
                    lds_atomic_max(obuf + (ioff + t) * C + c, __float_as_uint(v));
                } else if (p.merge) {
                    if (row0 + t < limit) atomicMax(reinterpret_cast<unsigned *>(p.out) + (row0 + t) * C + c, __float_as_uint(v));
                } else {
                    p.out[(row0 + t) * C + c] = v;
                }
            }
    }
    if (image) {
        __syncthreads();
        unsigned *gout = reinterpret_cast<unsigned *>(p.out) + img_lo * C;
        const int64_t lim = (p.n - img_lo) * C;
        for (int i = threadIdx.x; i < p.ospan * C; i += 256) {
            const unsigned v = obuf[i];
            if (v != 0u && i < lim) global_atomic_max(gout + i, v);
        }
    }
}

// element k of a row held as 16-byte chunks (fp16: 8 per chunk, fp32: 4)
template <typename AT, int N>
__device__ __forceinline__ float att_elem(const att_chunk (&row)[N], int k)
{
    if constexpr (sizeof(AT) == 2) return (float)__builtin_bit_cast(half8, row[k / 8])[k % 8];
    else return __builtin_bit_cast(f32x4, row[k / 4])[k % 4];
}

#ifndef DGRP_ATT_OCC
#define DGRP_ATT_OCC 1
#endif
#ifndef DGRP_ATT_PIPE_MINUP
#define DGRP_ATT_PIPE_MINUP 48
#endif
#ifndef DGRP_ATT_DEPTH2_MINUP
#define DGRP_ATT_DEPTH2_MINUP 48
#endif
template <int UP, int CM, typename AT>
__global__ void __launch_bounds__(256, (UP <= 64 && sizeof(AT) == 4) ? DGRP_ATT_OCC : 2) attention_wave_kernel(const att_params p)
{
    constexpr int CT = UP <= 64 ? 64 : 128;                  // context registers per lane (butterfly width)
    constexpr int EPC = 16 / sizeof(AT);                     // elements per 16-byte chunk
    __shared__ __attribute__((aligned(16))) float qs[4][UP / 2][4];   // per wave and unit pair: {c q[k], c q[k+1], -2 scale[k], -2 scale[k+1]}
    __shared__ __attribute__((aligned(16))) AT tile[4][64][UP + EPC];
    extern __shared__ __attribute__((aligned(16))) unsigned char att_dyn[];
    unsigned *obuf = reinterpret_cast<unsigned *>(att_dyn);  // merge: max image of the rows the 16 windows cover
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int T = p.T, C = p.C;
    const int64_t wg0 = (int64_t)blockIdx.x * ATT_WPB;       // a workgroup owns 16 consecutive windows, 4 per wave
    const int nvalid = (int)min((int64_t)ATT_WPB, p.nw - wg0);
    int64_t lo = 0;
    if (p.merge) {
        int64_t lim_;
        const int64_t a = att_window_row(p, wg0, &lim_), b = att_window_row(p, wg0 + nvalid - 1, &lim_);
        lo = a < b ? a : b;
        for (int i = threadIdx.x; i < p.ospan * C; i += 256) obuf[i] = 0u;
        __syncthreads();
    }
    constexpr float C2 = 2.8853900817779268f;                // 2 log2 e
    // a tile of 64 steps is fetched with fully coalesced 16-byte loads (1 KiB per wave instruction; a lane reading
    // its own 2*UP-byte row would touch 64 cache lines per instruction and thrash the 16 KiB L1), ahead of its use,
    // and turned into row-per-lane through the wave's private LDS tile (pitch UP + 8 halves: conflict-free both ways)
    constexpr int CPR = UP / EPC, TP = UP + EPC;             // 16-byte chunks per row, LDS row pitch
    AT *mytile = &tile[wave][0][0];
    // Tiles in flight per wave, in registers: with the 512-register budget of a lone wave a second tile ahead is free.
    constexpr int DEPTH = (sizeof(AT) == 4 && UP <= 64 && UP >= DGRP_ATT_DEPTH2_MINUP && DGRP_ATT_OCC == 1) ? 2 : 1;
    constexpr bool AHEAD = UP * sizeof(AT) <= 128 || UP <= 64;   // beyond 64 units the registers for a tile in flight are gone
    constexpr int QN = (UP + 63) / 64;                       // units per lane of a window's q row (k = lane, lane + 64)
    constexpr bool PIPE = sizeof(AT) == 4 && UP <= 64 && UP >= DGRP_ATT_PIPE_MINUP && DGRP_ATT_OCC == 1;   // a lone wave per SIMD: registers to spare for the next window's first loads
    constexpr int PCH = (CM <= 8 && PIPE) ? 8 : 0;           // 64-step chunks of stored logit halves kept in registers (T <= 512)
    att_chunk nxt[DEPTH][CPR];
    auto fetch = [&](att_chunk (&dst)[CPR], const AT *avg, int t0) {
        if (t0 + 64 <= T) {                                  // a whole tile: one pointer, the chunks at constant offsets of 1 KiB
            const att_chunk *src = reinterpret_cast<const att_chunk *>(avg + (int64_t)t0 * UP) + lane;
#pragma unroll
            for (int j = 0; j < CPR; ++j) dst[j] = src[j * 64];
            return;
        }
#pragma unroll
        for (int j = 0; j < CPR; ++j) {
            // steps behind the window's end read its last row again (their weight below is 0): a predicated load sits in a basic
            // block of its own, and behind 16 such blocks the compiler's counter analysis falls back to s_waitcnt vmcnt(0) in front of
            // the LDS writes -- every tile in flight drained, the second tile ahead worth nothing
            const int c = j * 64 + lane, r = c / CPR;
            const int rr = min(t0 + r, T - 1);
            dst[j] = *reinterpret_cast<const att_chunk *>(avg + (int64_t)rr * UP + (c % CPR) * EPC);
        }
    };
    // what does not change from window to window: the scale and its sum, this lane's rows of W_top
    float sc2[QN], ssum = 0.0f, wt[QN][CM];
#pragma unroll
    for (int i = 0; i < QN; ++i) {
        const int k = lane + 64 * i;
        const float sc = k < UP ? p.scale[k] : 0.0f;
        sc2[i] = -2.0f * sc;
        ssum += sc;
#pragma unroll
        for (int c = 0; c < CM; ++c) wt[i][c] = (k < UP && c < C) ? p.wtop[k * 16 + c] : 0.0f;
    }
    ssum = wave_allsum(ssum);
    // A window's first loads -- its q row (the Average of the two final states = avg[T-1]) and its first tiles -- are requested while
    // the window before it is still in its butterfly and output phase: a wave walks four windows, and with every window starting cold
    // (q row, then the first tiles, then the stored logit halves chunk by chunk: ~7 of a window's 24 us) the kernel sat in s_waitcnt
    // 40 % of its time (SQ_WAIT_ANY) at 3.8 TB/s.
    AT qn[QN];
    auto start_window = [&](int64_t wl) {
        const AT *avg = reinterpret_cast<const AT *>(p.avg) + wl * (int64_t)T * UP;
#pragma unroll
        for (int i = 0; i < QN; ++i) qn[i] = avg[(int64_t)(T - 1) * UP + min(lane + 64 * i, UP - 1)];
        if (AHEAD) {
            fetch(nxt[0], avg, 0);
            if constexpr (DEPTH == 2) fetch(nxt[1], avg, 64);
        }
    };
    if (PIPE && wave < nvalid) start_window(wg0 + wave);
  for (int wi = wave; wi < nvalid; wi += 4) {
    const int64_t wl = wg0 + wi;
    const AT *avg = reinterpret_cast<const AT *>(p.avg) + wl * (int64_t)T * UP;
    if (!PIPE) start_window(wl);
#pragma unroll
    for (int i = 0; i < QN; ++i) {
        const int k = lane + 64 * i;
        if (k < UP) {
            qs[wave][k >> 1][k & 1] = C2 * (float)qn[i];
            qs[wave][k >> 1][2 + (k & 1)] = sc2[i];
        }
    }
    float ctx[CT];
#pragma unroll
    for (int k = 0; k < CT; ++k) ctx[k] = 0.0f;
    float run_m = -INFINITY, run_l = 0.0f;
    for (int tb = 0; tb < T; tb += 64 * DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int t0 = tb + 64 * d;
        if (t0 >= T) break;
        const int t = t0 + lane;
        const bool ok = t < T;
        if (!AHEAD) fetch(nxt[d], avg, t0);
#pragma unroll
        for (int j = 0; j < CPR; ++j) {
            const int c = j * 64 + lane;
            *reinterpret_cast<att_chunk *>(mytile + (c / CPR) * TP + (c % CPR) * EPC) = nxt[d][j];
        }
        if (AHEAD && t0 + 64 * DEPTH < T) fetch(nxt[d], avg, t0 + 64 * DEPTH);
        att_chunk row[CPR];
#pragma unroll
        for (int j = 0; j < CPR; ++j) row[j] = *reinterpret_cast<const att_chunk *>(mytile + lane * TP + j * EPC);
        // (the offset is laundered so that the 2 UP loop-invariant LDS values are re-read per tile instead of
        // living in registers for the whole kernel)
        int qoff = wave * UP * 2;
        asm volatile("" : "+v"(qoff));
        const float *qw = &qs[0][0][0] + qoff;
        // Eight unit pairs at a time, stage by stage (a transcendental's result is not ready for the very next instruction: with the
        // stages of one pair back to back the compiler pads every one of them with s_nop), the pairs' constants as they lie in LDS
        // ({q, q, s, s}: packed operands without a register shuffle), four accumulators instead of one dependent chain
        f32x2 acc4[4] = { { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f } };
#pragma unroll
        for (int k0 = 0; k0 < UP; k0 += 16) {                // 8 broadcast reads in flight, then 8 unit pairs
            f32x4 qv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) qv[i] = *reinterpret_cast<const f32x4 *>(qw + 2 * (k0 + 2 * i));      // same address in every lane
            f32x2 x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = k0 + 2 * i;
                const f32x2 a = { att_elem<AT>(row, k), att_elem<AT>(row, k + 1) };
                x[i] = a * C2 + f32x2{ qv[i][0], qv[i][1] };
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = f32x2{ __builtin_amdgcn_exp2f(x[i].x), __builtin_amdgcn_exp2f(x[i].y) };
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = x[i] + 1.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = f32x2{ __builtin_amdgcn_rcpf(x[i].x), __builtin_amdgcn_rcpf(x[i].y) };
#pragma unroll
            for (int i = 0; i < 8; ++i) acc4[i & 3] = x[i] * f32x2{ qv[i][2], qv[i][3] } + acc4[i & 3];
        }
        const f32x2 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
        const float et = ok ? ssum + acc.x + acc.y : -INFINITY;
        const float new_m = fmaxf(run_m, wave_allmax(et));
        const float alpha = __builtin_amdgcn_exp2f(1.4426950408889634f * (run_m - new_m));      // 0 on the first tile
        const float pt = ok ? __builtin_amdgcn_exp2f(1.4426950408889634f * (et - new_m)) : 0.0f;
        run_l = run_l * alpha + wave_allsum(pt);
        run_m = new_m;
#pragma unroll
        for (int j = 0; j < CPR; ++j) asm volatile("" : "+v"(row[j]));      // convert again rather than keep UP floats alive
#pragma unroll
        for (int k = 0; k < UP; k += 2) {
            const f32x2 a = { att_elem<AT>(row, k), att_elem<AT>(row, k + 1) };
            const f32x2 c = f32x2{ ctx[k], ctx[k + 1] } * alpha + a * pt;
            ctx[k] = c.x; ctx[k + 1] = c.y;
        }
      }
    }
    // ---- this window's stored logit halves (all of them, if they fit the registers set aside), then the NEXT window's first loads:
    // both are in flight during the butterfly, and the loads return in order, so the output phase below waits for the first only
    const float *pl = p.pl + wl * (int64_t)T * C;
    const bool hoist = PCH > 0 && T <= 64 * PCH;
    float ph[PCH > 0 ? PCH : 1][CM];
    if (hoist) {
#pragma unroll
        for (int ch = 0; ch < PCH; ++ch) {
            const int t = min(lane + 64 * ch, T - 1);
            if (64 * ch < T) {
#pragma unroll
                for (int c = 0; c < CM; ++c) ph[ch][c] = pl[(int64_t)t * C + (c < C ? c : C - 1)];
            }
        }
    }
    if (PIPE && wi + 4 < nvalid) start_window(wl + 4);
    // ---- butterfly: lane l ends with the window's context for unit l (and unit l + 64 when CT = 128).
    // xor 32 / xor 16: v_permlane32_swap / v_permlane16_swap exchange exactly the halves (rows) the two partners
    // discard, so "kept + received" is one swap and one add; xor 8, 2, 1 are DPP moves; xor 4 has none on gfx9.
#pragma unroll
    for (int base = 0; base < CT; base += 64) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ctx[base + j]), __float_as_uint(ctx[base + j + 32]), false, false);
            ctx[base + j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(ctx[base + j]), __float_as_uint(ctx[base + j + 16]), false, false);
            ctx[base + j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        {
            const bool up = (lane & 8) != 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float lo = ctx[base + j] + row_ror<8>(ctx[base + j]), hi = ctx[base + j + 8] + row_ror<8>(ctx[base + j + 8]);
                ctx[base + j] = up ? hi : lo;
            }
        }
        {
            const bool up = (lane & 4) != 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float keep = up ? ctx[base + j + 4] : ctx[base + j];
                const float send = up ? ctx[base + j] : ctx[base + j + 4];
                ctx[base + j] = keep + __shfl_xor(send, 4);
            }
        }
        {
            const bool up = (lane & 2) != 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float lo = ctx[base + j] + quad_perm<0x4E>(ctx[base + j]), hi = ctx[base + j + 2] + quad_perm<0x4E>(ctx[base + j + 2]);
                ctx[base + j] = up ? hi : lo;
            }
        }
        {
            const float lo = ctx[base] + quad_perm<0xB1>(ctx[base]), hi = ctx[base + 1] + quad_perm<0xB1>(ctx[base + 1]);
            ctx[base] = (lane & 1) ? hi : lo;
        }
    }
    const float inv = __builtin_amdgcn_rcpf(run_l);
    float ctop[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        float part = ctx[0] * wt[0][c];
        if constexpr (CT == 128) part += ctx[64] * wt[QN - 1][c];
        ctop[c] = wave_allsum(part) * inv;
    }
    // ---- logits[t] = ctx.W_top + (avg[t].W_bot + b), softmax over classes, merge / store: lane <-> t
    int64_t limit = p.n;
    const int64_t row0 = p.merge ? att_window_row(p, wl, &limit) : wl * (int64_t)T;
    const int off = p.merge && row0 >= lo && row0 - lo + T <= p.ospan ? (int)(row0 - lo) : -1;
    // Straight-line over a compile-time class bound (addresses clamped, surplus classes at -inf): with a runtime
    // class loop every load sat in its own basic block and was waited for on its own -- five HBM latencies per tile.
    auto finish = [&](int t, const float (&pv)[CM]) {
        float lg[CM];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < CM; ++c) { lg[c] = c < C ? pv[c] + ctop[c] : -INFINITY; mx = fmaxf(mx, lg[c]); }
        float den = 0.0f;
#pragma unroll
        for (int c = 0; c < CM; ++c) { lg[c] = __builtin_amdgcn_exp2f(1.4426950408889634f * (lg[c] - mx)); den += lg[c]; }
        const float rden = __builtin_amdgcn_rcpf(den);
#pragma unroll
        for (int c = 0; c < CM; ++c)
            if (c < C) {
                const float v = lg[c] * rden;
                if (!p.merge) p.out[(row0 + t) * C + c] = v;
                else if (row0 + t >= limit) { }                  // beyond the record (partial-batch placement): dropped
                else if (off >= 0) lds_atomic_max(obuf + (off + t) * C + c, __float_as_uint(v));
                else global_atomic_max(reinterpret_cast<unsigned *>(p.out) + (row0 + t) * C + c, __float_as_uint(v));
            }
    };
    if (hoist) {
#pragma unroll
        for (int ch = 0; ch < PCH; ++ch)
            if (lane + 64 * ch < T) finish(lane + 64 * ch, ph[ch]);
    } else {
        // (windows of more than 512 steps, more than 8 classes: chunk by chunk, the next chunk requested before this one is used)
        float pn[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) pn[c] = pl[(int64_t)(lane < T ? lane : 0) * C + (c < C ? c : C - 1)];
        for (int t = lane; t < T; t += 64) {
            float pv[CM];
#pragma unroll
            for (int c = 0; c < CM; ++c) pv[c] = pn[c];
            const int tn = t + 64 < T ? t + 64 : t;
#pragma unroll
            for (int c = 0; c < CM; ++c) pn[c] = pl[(int64_t)tn * C + (c < C ? c : C - 1)];
            finish(t, pv);
        }
    }
  }
    if (p.merge && p.ospan > 0) {
        __syncthreads();
        // flush the pre-merged image: contiguous rows -> full-line atomic wave-instructions (as the GRU kernel does)
        unsigned *gout = reinterpret_cast<unsigned *>(p.out) + lo * C;
        const int64_t lim = (p.n - lo) * C;
        for (int i = threadIdx.x; i < p.ospan * C; i += 256) {
            const unsigned v = obuf[i];
            if (v != 0u && i < lim) global_atomic_max(gout + i, v);
        }
    }
}

// ------------------------------------------------------------------------------------------
template <int NW, int MODE, bool ONERCP>
static int launch_gru_mode(const gru_params &p, int64_t groups, size_t lds, hipStream_t stream)
{
    static std::once_flag configured;        // per instantiation; the attribute is per function, not per launch (records run on a pool of host threads)
    static hipError_t cfg_err = hipSuccess;
    std::call_once(configured, [] { cfg_err = hipFuncSetAttribute((const void *)gru_fused_kernel<NW, MODE, ONERCP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    DGRP_HIP(cfg_err);
    hipLaunchKernelGGL((gru_fused_kernel<NW, MODE, ONERCP>), dim3((unsigned)groups), dim3(64 * NW), lds, stream, p);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

template <int NW>
static int launch_gru(const gru_params &p, int64_t groups, size_t lds, bool onercp, hipStream_t stream)
{
    if constexpr (NW <= 4) {
        if (onercp) {
            switch (p.mode) {
            case 0: return launch_gru_mode<NW, 0, true>(p, groups, lds, stream);
            case 1: return launch_gru_mode<NW, 1, true>(p, groups, lds, stream);
            default: return launch_gru_mode<NW, 2, true>(p, groups, lds, stream);
            }
        }
    }
    switch (p.mode) {
    case 0: return launch_gru_mode<NW, 0, false>(p, groups, lds, stream);
    case 1: return launch_gru_mode<NW, 1, false>(p, groups, lds, stream);
    default: return launch_gru_mode<NW, 2, false>(p, groups, lds, stream);
    }
}

// gru_split2.hip: the two-tile split-operand kernel of the 128-unit class
int dgrp_split2_launch(const gru_params &p, int64_t groups, int half_bytes, bool onercp, hipStream_t stream);

template <int NW, bool ONERCP>
static int launch_split_rcp(const gru_params &p, int64_t groups, size_t lds, hipStream_t stream)
{
    static std::once_flag configured;
    static hipError_t cfg_err = hipSuccess;
    std::call_once(configured, [] {
        auto set = [](const void *f) { const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (e != hipSuccess) cfg_err = e; };
        set((const void *)gru_split_kernel<NW, 0, ONERCP>);
        set((const void *)gru_split_kernel<NW, 1, ONERCP>);
        set((const void *)gru_split_kernel<NW, 2, ONERCP>);
    });
    DGRP_HIP(cfg_err);
    if (p.mode == 0)
        hipLaunchKernelGGL((gru_split_kernel<NW, 0, ONERCP>), dim3((unsigned)groups), dim3(64 * NW), lds, stream, p);
    else if (p.mode == 1)
        hipLaunchKernelGGL((gru_split_kernel<NW, 1, ONERCP>), dim3((unsigned)groups), dim3(64 * NW), lds, stream, p);
    else
        hipLaunchKernelGGL((gru_split_kernel<NW, 2, ONERCP>), dim3((unsigned)groups), dim3(64 * NW), lds, stream, p);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}
template <int NW>
static int launch_split(const gru_params &p, int64_t groups, size_t lds, hipStream_t stream)
{
    return p.zfold != 0.0f ? launch_split_rcp<NW, true>(p, groups, lds, stream) : launch_split_rcp<NW, false>(p, groups, lds, stream);
}

// split-operand kernel selected (dgrp_model_set_precision) and applicable to this launch
// (the LSTM cell beyond 128 units has the streamed split-operand kernel only: it runs at either precision level)
static bool use_split(const dgrp_model *m, int)
{
    if (m->cell == 1 && m->NW > 4 && m->d_stream) return true;
    return m->precision == 1 && ((m->cell == 0 && m->NW <= 4 && m->d_pack_lo) || m->d_stream);
}

template <int NW>
static int launch_lstm(const gru_params &p, int64_t groups, size_t lds, hipStream_t stream)
{
    static std::once_flag configured;
    static hipError_t cfg_err = hipSuccess;
    std::call_once(configured, [] {
        auto set = [](const void *f) { const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (e != hipSuccess) cfg_err = e; };
        set((const void *)lstm_fused_kernel<NW, 0>);
        set((const void *)lstm_fused_kernel<NW, 1>);
    });
    DGRP_HIP(cfg_err);
    if (p.mode == 0)
        hipLaunchKernelGGL((lstm_fused_kernel<NW, 0>), dim3((unsigned)groups), dim3(64 * NW), lds, stream, p);
    else
        hipLaunchKernelGGL((lstm_fused_kernel<NW, 1>), dim3((unsigned)groups), dim3(64 * NW), lds, stream, p);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

// LDS carve of one row tile: fixed part, output image (mode 0: as many of the rows 16 windows span as `budget` bytes allow), and
// behind them the lo halves of the hidden tile (split kernels).  Sets p.ospan and p.lo_tile_off, returns the tile's bytes.
static size_t gru_tile_carve(const dgrp_model *m, gru_params &p, int mode, int64_t s, bool split, int pad, int64_t budget)
{
    const int lo_tiles = split ? gru_lds_hbuf(m->UP, pad) : 0;
    const int fixed = gru_lds_hbuf(m->UP, pad) + gru_lds_dpart(m->NW) + gru_lds_seq(p.Tp) + gru_lds_meta();
    p.ospan = 0;
    if (mode == 0) {
        const int64_t want = (DGRP_WG_WINDOWS - 1) * s + m->T;
        const int64_t cap = (budget - fixed - lo_tiles) / (m->C * 4);
        p.ospan = (int)(want < cap ? want : cap);
        if (p.ospan < m->T) p.ospan = 0;
    }
    p.lo_tile_off = (int)dgrp_align_up(fixed + (int64_t)p.ospan * m->C * 4, 16);
    return split ? (size_t)p.lo_tile_off + lo_tiles : (size_t)fixed + (size_t)p.ospan * m->C * 4;
}
// LDS a one-tile workgroup may carve (the fixed part, then as many rows of the merged-output image as fit; windows whose rows
// do not fit go to HBM with atomics of their own: the result is the same bit for bit).  The registers allow two waves per SIMD, i.e.
// 8 / NW workgroups per CU -- the image must not be what keeps them out: with 72 KiB a 32-unit model ran two of its eight
// workgroups per CU's worth of LDS (2 300 -> 2 930 Mbp/s with 19 KiB), a 64-unit split-operand model three of four (1 120 -> 1 210
// with 39 KiB; its fp16-operand kernel is 1 % better off with the larger image).  Five waves and more: one workgroup per CU.
static int64_t tile_budget(const dgrp_model *m, bool split)
{
    if (const char *e = getenv("DGRP_TILE_BUDGET_KB")) return (int64_t)atoi(e) * 1024;
    if (m->NW == 1) return 19 * 1024;
    if (m->NW == 2 && split) return 39 * 1024;
    return (m->NW > 4 ? 144 : 72) * 1024;
}
// gru_split2_kernel (gru_split2.hip): two tile carves with row pitch UP + 16, then the input-projection table
#define DGRP_SPLIT2_PAD 16
#define DGRP_SPLIT2_XTAB_BYTES (5 * (4 * 128 * 4 + 32))
static bool split2_applies(const dgrp_model *m) { return m->NW == 4 && m->d_pack16 && !getenv("DGRP_SPLIT_ONE_TILE"); }
// gru_wave_kernel (gru_wave.hip): GRU up to 64 units; four waves' carves and the table must fit the CU's LDS -- a property of the model's
// window and step, never of the record
// (17-32 units as an attention pre-pass: the one-wave workgroups of gru_split_kernel<1> -- no partner wave to meet either -- measured 5-7 %
// faster than two unit groups here, r03 `tools/bench_shapes.py`; every other count of unit groups and every merged / window-output launch
// is faster on this kernel)
static bool wave_applies(const dgrp_model *m, int mode)
{
    return m->cell == 0 && m->NU16 > 0 && m->d_packw && !(m->NU16 == 2 && mode == 2) && !getenv("DGRP_SPLIT_ONE_TILE");
}
static int wave_try(const dgrp_model *m, gru_params &p, int mode, int64_t s);
// Row length (elements) of the avg[t] spill between the attention pre-pass and the second kernel: the pre-pass kernel's unit padding --
// 16 NU for gru_wave_kernel (a 36-unit model spills 48 floats per step, not 64), the model's UP (multiple of 32) otherwise.  A property
// of the model and its precision level only: the pre-pass never needs an image, so whether four waves' carves fit depends on T alone.
int dgrp_spill_row(const dgrp_model *m)
{
    if (m->precision == 1 && wave_applies(m, 2)) {
        gru_params q;
        q.T = m->T; q.C = m->C; q.Tp = (int)dgrp_align_up(m->T, 16);
        if (wave_try(m, q, 2, 1)) return 16 * m->NU16;
    }
    return m->UP;
}
static int wave_try(const dgrp_model *m, gru_params &p, int mode, int64_t s)
{
    const int64_t budget = (160 * 1024 - dgrp_wave_table_bytes(m->NU16)) / 4 / 16 * 16;
    const int wb = dgrp_wave_carve(m->NU16, p, mode, s, budget);
    return wb <= budget ? wb : 0;
}

int dgrp_gru_launch(const dgrp_model *m, const uint8_t *d_idx, int64_t n, int64_t s, dgrp_placement place,
                    int64_t w0, int64_t nw, int mode, float *d_out, void *d_avg, hipStream_t stream)
{
    if (nw <= 0) return DGRP_OK;
    dgrp_timer_scope timed(stream, nw);                             // bench.py: HIP events around the launch (no-op unless enabled)
    gru_params p;
    p.idx = d_idx; p.n = n; p.s = s; p.w0 = w0; p.nw = nw; p.place = place;
    p.pack = m->d_pack; p.ffb = m->d_ffb; p.out = d_out; p.avg = d_avg; p.avg_f32 = m->precision == 1 ? 1 : 0;
    p.T = m->T; p.C = m->C; p.nfrag = m->nfrag; p.mode = mode;
    p.Tp = (int)dgrp_align_up(m->T, 16);
    p.stamps = nullptr;
    p.recs = nullptr; p.wg_first = nullptr; p.nrec = 0; p.avgw = 0;
#ifdef DGRP_STAMP
    static uint64_t *d_stamps = nullptr;
    const int64_t ngroups = (nw + DGRP_WG_WINDOWS - 1) / DGRP_WG_WINDOWS;
    const size_t stamp_bytes = (size_t)ngroups * m->NW * 16 * 8;
    if (getenv("DGRP_STAMP_DUMP")) {
        if (d_stamps) (void)hipFree(d_stamps);
        DGRP_HIP(hipMalloc((void **)&d_stamps, stamp_bytes));
        DGRP_HIP(hipMemset(d_stamps, 0, stamp_bytes));
        p.stamps = d_stamps;
    }
#endif
    const bool split = use_split(m, mode);
    p.pack_lo = m->d_pack_lo; p.zfold = m->onercp ? 1.0f : 0.0f;
    p.pack16 = m->d_pack16; p.xtab = m->d_xtab; p.xtab_off = 0; p.stream = m->d_stream;
    p.packw = m->d_packw; p.xtabw = m->d_xtabw; p.avg_up = dgrp_spill_row(m);
    const int64_t groups = (nw + DGRP_WG_WINDOWS - 1) / DGRP_WG_WINDOWS;
    DGRP_REQUIRE(groups < (1ll << 31), "too many windows in one launch (%lld)", (long long)nw);
    if (split && wave_applies(m, mode)) {
        if (const int wb = wave_try(m, p, mode, s)) return dgrp_wave_launch(p, m->NU16, groups, wb, m->onercp != 0, stream);
    }
    if (split && split2_applies(m)) {
        // 128-unit class: two row tiles per workgroup, everything resident, whenever two carves and the table fit the CU's LDS
        // (a property of the model's window size, not of the record: a record never changes kernels with the way it is batched)
        const int half_bytes = (int)dgrp_align_up((int64_t)gru_tile_carve(m, p, mode, s, true, DGRP_SPLIT2_PAD, 74 * 1024), 256);
        if (2 * half_bytes + DGRP_SPLIT2_XTAB_BYTES <= 160 * 1024) {
            p.xtab_off = 2 * half_bytes;
            return dgrp_split2_launch(p, groups, half_bytes, m->onercp != 0, stream);
        }
    }
    // rows spanned by 16 consecutive windows, capped so that the workgroups the registers allow fit a CU's 160 KiB (tile_budget)
    const size_t lds = gru_tile_carve(m, p, mode, s, split, 8, tile_budget(m, split));
    DGRP_REQUIRE(lds <= 160 * 1024, "window size %d: the workgroup's staged sequences (%d bytes of LDS) do not fit 160 KiB", m->T,
                 gru_lds_seq(p.Tp));
    if (split && m->d_stream && m->cell == 0 && m->d_xtab && !getenv("DGRP_STREAM_PLAIN")) {
        // GRU, 129-256 units: waves of 64 units, resident hi fragments (rnn_stream.hip); one workgroup per CU
        const size_t lds64 = dgrp_stream64_carve(m->NW, p, mode, s, 158 * 1024);
        if (lds64 <= 160 * 1024) return dgrp_stream64_launch(p, m->NW, groups, lds64, stream);
        (void)gru_tile_carve(m, p, mode, s, split, 8, tile_budget(m, split));          // (window too long for that carve: the all-streamed kernel)
    }
    if (split && m->d_stream) return dgrp_stream_launch(p, m->cell, m->NW, groups, lds, stream);
    if (split) {
        switch (m->NW) {
        case 1: return launch_split<1>(p, groups, lds, stream);
        case 2: return launch_split<2>(p, groups, lds, stream);
        case 3: return launch_split<3>(p, groups, lds, stream);
        default: return launch_split<4>(p, groups, lds, stream);
        }
    }
    if (m->cell == 1) {
        switch (m->NW) {
        case 1: return launch_lstm<1>(p, groups, lds, stream);
        case 2: return launch_lstm<2>(p, groups, lds, stream);
        case 3: return launch_lstm<3>(p, groups, lds, stream);
        case 4: return launch_lstm<4>(p, groups, lds, stream);
        default:
            dgrp_set_error("LSTM units=%d not supported (max 128)", m->u);
            return DGRP_EINVAL;
        }
    }
#ifdef DGRP_STAMP
    if (p.stamps && m->NW == 4) {
        int rc = launch_gru<4>(p, groups, lds, m->onercp != 0, stream);
        (void)hipStreamSynchronize(stream);
        std::vector<uint64_t> hst(stamp_bytes / 8);
        (void)hipMemcpy(hst.data(), p.stamps, stamp_bytes, hipMemcpyDeviceToHost);
        FILE *f = fopen(getenv("DGRP_STAMP_DUMP"), "wb");
        if (f) { fwrite(hst.data(), 1, stamp_bytes, f); fclose(f); }
        return rc;
    }
#endif
    switch (m->NW) {
    case 1: return launch_gru<1>(p, groups, lds, m->onercp != 0, stream);
    case 2: return launch_gru<2>(p, groups, lds, m->onercp != 0, stream);
    case 3: return launch_gru<3>(p, groups, lds, m->onercp != 0, stream);
    case 4: return launch_gru<4>(p, groups, lds, m->onercp != 0, stream);
    case 5: return launch_gru<5>(p, groups, lds, m->onercp != 0, stream);
    case 6: return launch_gru<6>(p, groups, lds, m->onercp != 0, stream);
    case 7: return launch_gru<7>(p, groups, lds, m->onercp != 0, stream);
    case 8: return launch_gru<8>(p, groups, lds, m->onercp != 0, stream);
    default:
        dgrp_set_error("units=%d not supported by the GRU kernel (max 256)", m->u);
        return DGRP_EINVAL;
    }
}

// GRU forward + merge for a batch of records in ONE launch (mode 0, no attention): the record table and the
// cumulative workgroup counts are device arrays prepared by the caller (api.hip: dgrp_predict_batch)
int dgrp_gru_launch_batch(const dgrp_model *m, const uint8_t *d_idx, int64_t s, const void *d_recs, const int64_t *d_wg_first,
                          int64_t nrec, int64_t total_groups, int mode, float *d_out, void *d_avg, hipStream_t stream)
{
    if (nrec <= 0 || total_groups <= 0) return DGRP_OK;
    dgrp_timer_scope timed(stream, total_groups * DGRP_WG_WINDOWS);
    DGRP_REQUIRE(m->NW <= 8 && (mode == 0 || (mode == 2 && m->cell == 0)), "dgrp_gru_launch_batch: mode 0, or the GRU attention pre-pass");
    gru_params p;
    p.idx = d_idx; p.n = 0; p.s = s; p.w0 = 0; p.nw = 0; p.place = dgrp_placement{ 0, 0 };
    p.pack = m->d_pack; p.ffb = m->d_ffb; p.out = d_out; p.avg = d_avg; p.avg_f32 = m->precision == 1 ? 1 : 0;
    p.T = m->T; p.C = m->C; p.nfrag = m->nfrag; p.mode = mode;
    p.Tp = (int)dgrp_align_up(m->T, 16);
    p.stamps = nullptr;
    p.recs = (const gru_rec *)d_recs; p.wg_first = d_wg_first; p.nrec = nrec; p.avgw = 0;
    const bool split = use_split(m, mode);
    p.pack_lo = m->d_pack_lo; p.zfold = m->onercp ? 1.0f : 0.0f;
    p.pack16 = m->d_pack16; p.xtab = m->d_xtab; p.xtab_off = 0; p.stream = m->d_stream;
    p.packw = m->d_packw; p.xtabw = m->d_xtabw; p.avg_up = dgrp_spill_row(m);
    DGRP_REQUIRE(total_groups < (1ll << 31), "too many windows in one launch");
    if (split && wave_applies(m, mode)) {
        if (const int wb = wave_try(m, p, mode, s)) return dgrp_wave_launch(p, m->NU16, total_groups, wb, m->onercp != 0, stream);
    }
    if (split && split2_applies(m)) {
        const int half_bytes = (int)dgrp_align_up((int64_t)gru_tile_carve(m, p, mode, s, true, DGRP_SPLIT2_PAD, 74 * 1024), 256);
        if (2 * half_bytes + DGRP_SPLIT2_XTAB_BYTES <= 160 * 1024) {
            p.xtab_off = 2 * half_bytes;
            return dgrp_split2_launch(p, total_groups, half_bytes, m->onercp != 0, stream);
        }
    }
    const size_t lds = gru_tile_carve(m, p, mode, s, split, 8, tile_budget(m, split));
    DGRP_REQUIRE(lds <= 160 * 1024, "window size %d: the workgroup's staged sequences (%d bytes of LDS) do not fit 160 KiB", m->T,
                 gru_lds_seq(p.Tp));
    if (split && m->d_stream && m->cell == 0 && m->d_xtab && !getenv("DGRP_STREAM_PLAIN")) {
        const size_t lds64 = dgrp_stream64_carve(m->NW, p, mode, s, 158 * 1024);
        if (lds64 <= 160 * 1024) return dgrp_stream64_launch(p, m->NW, total_groups, lds64, stream);
        (void)gru_tile_carve(m, p, mode, s, split, 8, tile_budget(m, split));
    }
    if (split && m->d_stream) return dgrp_stream_launch(p, m->cell, m->NW, total_groups, lds, stream);
    if (split) {
        switch (m->NW) {
        case 1: return launch_split<1>(p, total_groups, lds, stream);
        case 2: return launch_split<2>(p, total_groups, lds, stream);
        case 3: return launch_split<3>(p, total_groups, lds, stream);
        default: return launch_split<4>(p, total_groups, lds, stream);
        }
    }
    if (m->cell == 1) {
        switch (m->NW) {
        case 1: return launch_lstm<1>(p, total_groups, lds, stream);
        case 2: return launch_lstm<2>(p, total_groups, lds, stream);
        case 3: return launch_lstm<3>(p, total_groups, lds, stream);
        default: return launch_lstm<4>(p, total_groups, lds, stream);
        }
    }
    switch (m->NW) {
    case 1: return launch_gru<1>(p, total_groups, lds, m->onercp != 0, stream);
    case 2: return launch_gru<2>(p, total_groups, lds, m->onercp != 0, stream);
    case 3: return launch_gru<3>(p, total_groups, lds, m->onercp != 0, stream);
    case 4: return launch_gru<4>(p, total_groups, lds, m->onercp != 0, stream);
    case 5: return launch_gru<5>(p, total_groups, lds, m->onercp != 0, stream);
    case 6: return launch_gru<6>(p, total_groups, lds, m->onercp != 0, stream);
    case 7: return launch_gru<7>(p, total_groups, lds, m->onercp != 0, stream);
    default: return launch_gru<8>(p, total_groups, lds, m->onercp != 0, stream);
    }
}

// d_recs / nrec: record table of a batch (windows numbered through the batch, n = rows of the whole batch) or NULL / 0
int dgrp_attention_launch_recs(const dgrp_model *m, int64_t s, dgrp_placement place, int64_t w0, int64_t nw,
                               int merge, int64_t n, const void *d_avg, const float *d_pl, float *d_out,
                               const void *d_recs, int64_t nrec, hipStream_t stream)
{
    if (nw <= 0) return DGRP_OK;
    att_params p;
    p.avg = d_avg; p.pl = d_pl; p.scale = m->d_scale; p.wtop = m->d_wtop; p.out = d_out;
    p.n = n; p.s = s; p.w0 = w0; p.nw = nw; p.place = place;
    p.T = m->T; p.C = m->C; p.UP = dgrp_spill_row(m); p.merge = merge;
    p.recs = (const gru_rec *)d_recs; p.nrec = nrec;
    const int UPs = p.UP;                                     // row length of the spill: the pre-pass kernel's unit padding
    // element type of the avg[t] spill: fp32 behind a split-operand pre-pass (the level the model is set to), else fp16
    const bool f32 = m->precision == 1;
    const int esz = f32 ? 4 : 2;
    DGRP_REQUIRE(nw < (1ll << 31), "too many windows in one launch");
    p.ospan = 0;
    const int64_t want = (ATT_WPB - 1) * s + m->T;
    const unsigned grid = (unsigned)((nw + ATT_WPB - 1) / ATT_WPB);
    if (UPs <= 64) {
        // one wave per window; beyond 64 units its per-lane context (UP registers) no longer fits beside the row
        const int stat = 4 * UPs * 2 * 4 + 4 * 64 * (UPs + 16 / esz) * esz;  // qs + tiles (static LDS of the kernel)
        if (merge) {
            const int64_t budget = (stat <= 40 * 1024 && !f32) || DGRP_ATT_OCC == 2 ? 78 * 1024 : 156 * 1024;   // two workgroups per CU where tiles and registers leave room
            const int64_t cap = (budget - stat) / (m->C * 4);
            p.ospan = (int)(want < cap ? want : cap);
            if (p.ospan < m->T) p.ospan = 0;
        }
        const size_t dyn = (size_t)p.ospan * m->C * 4;
        static std::once_flag configured;
        static hipError_t cfg_err = hipSuccess;
        std::call_once(configured, [] {                          // static + dynamic LDS may use the whole 160 KiB
#define ATT_CFG(UPv, CMv, ATv) do { hipError_t e_ = hipFuncSetAttribute((const void *)attention_wave_kernel<UPv, CMv, ATv>, \
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (4 * UPv * 8 + 256 * (UPv + (int)(16 / sizeof(ATv))) * (int)sizeof(ATv))); \
        if (e_ != hipSuccess) cfg_err = e_; } while (0)
            ATT_CFG(32, 8, _Float16); ATT_CFG(32, 16, _Float16); ATT_CFG(64, 8, _Float16); ATT_CFG(64, 16, _Float16);
            ATT_CFG(32, 8, float); ATT_CFG(32, 16, float); ATT_CFG(64, 8, float); ATT_CFG(64, 16, float);
            ATT_CFG(16, 8, float); ATT_CFG(16, 16, float); ATT_CFG(48, 8, float); ATT_CFG(48, 16, float);
#undef ATT_CFG
        });
        DGRP_HIP(cfg_err);
#define ATT_GO(UPv, CMv) do { if (f32) hipLaunchKernelGGL((attention_wave_kernel<UPv, CMv, float>), dim3(grid), dim3(256), dyn, stream, p); \
                              else hipLaunchKernelGGL((attention_wave_kernel<UPv, CMv, _Float16>), dim3(grid), dim3(256), dyn, stream, p); } while (0)
#define ATT_GO32(UPv, CMv) hipLaunchKernelGGL((attention_wave_kernel<UPv, CMv, float>), dim3(grid), dim3(256), dyn, stream, p)
        if (UPs == 16 || UPs == 48) {                            // rows of 16 / 48 floats: only gru_wave_kernel writes them (fp32 spill)
            DGRP_REQUIRE(f32, "attention: a %d-unit spill row is fp32", UPs);
            if (UPs == 16 && m->C <= 8) ATT_GO32(16, 8);
            else if (UPs == 16) ATT_GO32(16, 16);
            else if (m->C <= 8) ATT_GO32(48, 8);
            else ATT_GO32(48, 16);
        }
        else if (UPs == 32 && m->C <= 8) ATT_GO(32, 8);
        else if (UPs == 32) ATT_GO(32, 16);
        else if (m->C <= 8) ATT_GO(64, 8);
        else ATT_GO(64, 16);
#undef ATT_GO
#undef ATT_GO32
        DGRP_LAUNCH_CHECK();
        return DGRP_OK;
    }
    // 65-256 units: one wave per window straight from HBM (attention_row_kernel), 2 units per lane up to 128 units, 4 beyond
    const unsigned rgrid = (unsigned)((nw + 3) / 4);
    // merge: the rows four consecutive windows cover, as an LDS image (if they fit 48 KiB; batched records: rows of different records)
    const int64_t span = 3 * s + m->T;
    p.ospan = merge && !d_recs && span * m->C * 4 <= 48 * 1024 ? (int)span : 0;
    const size_t rdyn = (size_t)p.ospan * m->C * 4;
    if (m->UP <= 128) {
        if (f32) hipLaunchKernelGGL((attention_row_kernel<float, 2, 8>), dim3(rgrid), dim3(256), rdyn, stream, p);
        else hipLaunchKernelGGL((attention_row_kernel<_Float16, 2, 4>), dim3(rgrid), dim3(256), rdyn, stream, p);
    } else {
        if (f32) hipLaunchKernelGGL((attention_row_kernel<float, 4, 8>), dim3(rgrid), dim3(256), rdyn, stream, p);
        else hipLaunchKernelGGL((attention_row_kernel<_Float16, 4, 8>), dim3(rgrid), dim3(256), rdyn, stream, p);
    }
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

int dgrp_attention_launch(const dgrp_model *m, int64_t s, dgrp_placement place, int64_t w0, int64_t nw,
                          int merge, int64_t n, const void *d_avg, const float *d_pl, float *d_out,
                          hipStream_t stream)
{
    return dgrp_attention_launch_recs(m, s, place, w0, nw, merge, n, d_avg, d_pl, d_out, nullptr, 0, stream);
}


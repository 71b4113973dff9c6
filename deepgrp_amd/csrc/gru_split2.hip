// A4 (+A5/A6 fused), split operands, the 128-unit class (97-128 units): the default forward kernel of the benchmark model.
//
// Same mathematics as gru_split_kernel (gru_kernel.hip): U = U_hi + U_lo and h_{t-1} = h_hi + h_lo as fp16 pairs,
// U.h ~ U_hi.h_hi + U_hi.h_lo + U_lo.h_hi on the matrix cores with fp32 accumulation (72 + 4 v_mfma_f32_32x32x16_f16 and
// 6 v_mfma_f32_16x16x32_f16 per wave and step), the gate chain of gru_shared.h link for link -- bit-identical results.
//
// What differs is where everything lives and when it is issued:
//   * ONE wave per SIMD with the 512-register budget.  The 54 weight fragments of the wave's 32 units (U_hi, U_lo, input
//     projection, Dense: 216 registers) are loaded ONCE, straight into the accumulation half of the register file (asm
//     loads with AGPR destinations), and every MFMA names them there as its A operand: no copies, nothing streams, and
//     the compiler's 256 architectural VGPRs stay free for two tiles' accumulators and state.
//   * TWO row tiles (2 x 16 windows) per workgroup, software-pipelined against each other: a "phase" is the 82 MFMAs of
//     one tile's step (X) with the whole epilogue of the other tile's step (Y) -- gate chains, fp16 hi/lo publish, the
//     workgroup barrier, softmax + max-merge of the step before, the one-hot operand and first fragments of Y's next
//     step -- cut into single operations and dropped into the gaps BETWEEN X's MFMAs.  A wave issues in order and an MFMA
//     occupies the matrix pipe for 32 cycles but the issue port for 8: up to ~24 cycles of vector work per gap run in the
//     MFMA's shadow (tools/ubench/mfma_stage_vs_bulk.hip).  The barrier itself sits in the middle of X's MFMA stream, so
//     the pipe has work while the four waves meet and while the first LDS reads behind the barrier are in flight.
//   * The order of that interleave is generated (tools/gen_split2_schedule.py -> gru_split2_phase.inc) from a small cost
//     model (transcendental 8, plain VALU 4, 24 per gap) and pinned with sched_barrier between gaps; the MFMAs are
//     asm volatile, which the compiler keeps in program order.
//
// Inline-asm obligations (cdna_hip_programming.md 5.7) and how they are met:
//   - an MFMA's result is read by compiler-scheduled code only several MFMAs later (the generator keeps the first two
//     gaps of a phase free of anything that touches the previous phase's accumulators or Dense result);
//   - MFMA operands written by VALU code (the one-hot operand) are written one phase earlier;
//   - LDS reads feeding an asm MFMA are the compiler's own loads: it waits for them in front of the statement;
//   - the weight loads carry their own s_waitcnt inside the statement that issues them.
#include "gru_shared.h"
#include <type_traits>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// experiment switch (never set in the product build): 1 drops the U_lo.h_hi pass, 2 the U_hi.h_lo pass -- what a two-pass
// split would cost in accuracy (tools/twopass_probe.py)
#ifndef DGRP_SPLIT_DROP
#define DGRP_SPLIT_DROP 0
#endif

namespace {

constexpr int NW = 4, UP = 128, KS = 8, HS = UP + 8;

struct split2_weights {                   // 54 fragments = 216 AGPRs per lane, resident for the whole kernel
    u32x4 Br[KS + 1], Bg[KS + 1], Bz[KS + 1];      // U_hi of the gates, [KS] = the input k-step (kernel rows + biases, hi|lo)
    u32x4 Lr[KS], Lg[KS], Lz[KS];                  // U_lo
    u32x4 Bxh, Bd_hi, Bd_lo;                       // candidate's input projection; Dense hi / lo
};

#define LOAD3(a, pa, b, pb, c, pc)                                                                                      \
    asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %4, off\n\tglobal_load_dwordx4 %2, %5, off\n\t" \
                 "s_waitcnt vmcnt(0)"                                                                                   \
                 : "=&a"(a), "=&a"(b), "=&a"(c)                                                                         \
                 : "v"(pa), "v"(pb), "v"(pc)                                                                            \
                 : "memory")

#define MFMA32(acc, Wf, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(Wf), "v"(b))
#define MFMA32Z(acc, Wf, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(Wf), "v"(b))
#define MFMA16(acc, a, Wf) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(Wf))
#define MFMA16Z(acc, a, Wf) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(Wf))

struct tile_state {
    gru_params p;                         // batched records: the two tiles may belong to different records
    wg_ctx ctx;
    unsigned hcur, hnxt, lcur, lnxt;      // byte offsets of the hidden tiles (hi, lo; ping-pong) in the dynamic LDS
    const uint8_t *myseq;
    float h[16];                          // gate state of the lane's 16 (row, unit) pairs: h, or h - 1 (ONERCP)
    f32x16 ar, ag, az, ax;                // pre-activations of the step in flight (ax: the candidate's input projection)
    f32x4 dpl;                            // Dense partial logits of the previous step
    half8 xa;                             // one-hot operand of the NEXT contraction
    half8 f0h, f0l;                       // its first hidden-tile fragments
    int p_off;                            // placement of the wave's logit register (reg = wave): row in the LDS image or -1
    int64_t p_row0;
    bool pr_on;
};

template <int MODE, bool ONERCP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) gru_split2_kernel(const gru_params pin, int half_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = pin.T, C = pin.C;

    split2_weights W;
    {
        const uint4 *mypack = pin.pack + (size_t)wave * pin.nfrag * 64 + lane;
        const uint4 *mylo = pin.pack_lo + (size_t)wave * 3 * KS * 64 + lane;
#pragma unroll
        for (int k = 0; k <= KS; ++k)      // gate order of the pack: z, r, g (api.hip)
            LOAD3(W.Bz[k], mypack + (size_t)k * 64, W.Br[k], mypack + (size_t)(KS + 1 + k) * 64, W.Bg[k], mypack + (size_t)(2 * (KS + 1) + k) * 64);
#pragma unroll
        for (int k = 0; k < KS; ++k)       // lo stream: k-step major, gates r, g, z
            LOAD3(W.Lr[k], mylo + (size_t)(3 * k) * 64, W.Lg[k], mylo + (size_t)(3 * k + 1) * 64, W.Lz[k], mylo + (size_t)(3 * k + 2) * 64);
        LOAD3(W.Bxh, mypack + (size_t)(3 * (KS + 1)) * 64, W.Bd_hi, mypack + (size_t)(3 * (KS + 1) + 1) * 64, W.Bd_lo,
              mypack + (size_t)(3 * (KS + 1) + 2) * 64);
    }

    const int r = lane & 31, wi_a = r & 15, dir = r >> 4, khalf = lane >> 5;
    const int cls = lane & 15;
    const float fbias = cls < C ? pin.ffb[cls] : 0.0f;
    const int pwi = 4 * (lane >> 4) + wave;                       // window of the wave's logit register
    const unsigned frag_lane = (unsigned)(r * HS + 8 * khalf) * 2;                          // this lane's part of a fragment address
    const unsigned dense_lane = (unsigned)((lane & 15) * HS + 32 * wave + 8 * (lane >> 4)) * 2;
    const unsigned pub_lane = (unsigned)((lane & 31) * HS + 32 * wave + 4 * khalf) * 2;

    tile_state S0, S1;                      // two named objects, never indexed: they must stay in registers
    auto setup = [&](tile_state &Z, int x) {
        unsigned char *base = smem + (size_t)x * half_bytes;
        _Float16 *lbuf = reinterpret_cast<_Float16 *>(base + pin.lo_tile_off);
        for (int i = tid; i < 32 * HS; i += 256) lbuf[i] = (_Float16)0.0f;
        Z.p = pin;
        const int64_t bid = wg_record_at<MODE>(pin, Z.p, 2 * (int64_t)blockIdx.x + x);
        Z.ctx = wg_setup<NW, MODE>(Z.p, base, bid);                                    // ends with a barrier
        Z.hcur = (unsigned)x * half_bytes;                        // hbuf is the first item of the carve
        Z.hnxt = Z.hcur + 32 * HS * 2;
        Z.lcur = (unsigned)x * half_bytes + pin.lo_tile_off;
        Z.lnxt = Z.lcur + 32 * HS * 2;
        Z.myseq = Z.ctx.seqs + wi_a * pin.Tp;
#pragma unroll
        for (int i = 0; i < 16; ++i) Z.h[i] = ONERCP ? -1.0f : 0.0f;
        Z.p_off = Z.ctx.rowoff[pwi];
        Z.p_row0 = Z.ctx.row0s[pwi];
        Z.pr_on = cls < C && pwi < Z.ctx.nvalid;
    };
    setup(S0, 0);
    setup(S1, 1);

    auto lds16 = [&](unsigned off) -> half8 { return *reinterpret_cast<const half8 *>(smem + off); };
    // one-hot(base) | 1 of step tn as the B operand of the input k-step (complement table [3,2,1,0,4], model.py:233-237)
    auto onehot = [&](uint32_t b) -> half8 {
        if (dir) b = b < 4 ? 3 - b : 4;
        const uint32_t one = 0x3C00u << ((b & 1) * 16);
        const uint32_t sel = b >> 1;
        const uint4 xu = make_uint4(sel == 0 ? one : 0u, sel == 1 ? one : 0u, (sel == 2 ? one : 0u) | 0x3C000000u, 0u);
        return __builtin_bit_cast(half8, xu);
    };
    auto step_base = [&](const tile_state &Z, int tn) -> uint32_t { return Z.myseq[dir ? T - 1 - tn : tn]; };

    // ---- one phase: tile X's step tx on the matrix pipe, tile Y's epilogue of step ty in the gaps --------------------
    struct frag_ring { half8 h[2], l[2]; };
    struct dense_ops { half8 a0, a1, l0, l1; };
    struct fin_state { float d[NW], lg, m, e, s; };
    auto phase = [&](auto do_x, auto do_y, tile_state &X, tile_state &Y, int tx, int ty) __attribute__((always_inline)) {
        constexpr bool DO_X = decltype(do_x)::value, DO_Y = decltype(do_y)::value;
        frag_ring F;
        dense_ops D;
        fin_state fs;
        split_gate_tmp gt[16];
        float pbh[4][4];
        uint2 pbhv[4], pblv[4];
        uint32_t xp_b = 4;
        const int tf = ty - 1;                                    // step whose logits Y finishes in this phase
        if constexpr (DO_X) { F.h[0] = X.f0h; F.l[0] = X.f0l; }
#define GAP __builtin_amdgcn_sched_barrier(0);
#define M_IN(i)                                                                                                  \
    if constexpr (DO_X) {                                                                                        \
        if constexpr ((i) == 0) MFMA32Z(X.ar, W.Br[KS], X.xa);                                                   \
        else if constexpr ((i) == 1) MFMA32Z(X.ag, W.Bg[KS], X.xa);                                              \
        else if constexpr ((i) == 2) MFMA32Z(X.az, W.Bz[KS], X.xa);                                              \
        else MFMA32Z(X.ax, W.Bxh, X.xa);                                                                         \
    }
#define M_K(k, j)                                                                                                \
    if constexpr (DO_X && !(DGRP_SPLIT_DROP == 1 && (j) >= 6) && !(DGRP_SPLIT_DROP == 2 && (j) >= 3 && (j) < 6)) {                                                                                      \
        if constexpr ((j) == 0) MFMA32(X.ar, W.Br[k], F.h[(k) & 1]);                                             \
        else if constexpr ((j) == 1) MFMA32(X.ag, W.Bg[k], F.h[(k) & 1]);                                        \
        else if constexpr ((j) == 2) MFMA32(X.az, W.Bz[k], F.h[(k) & 1]);                                        \
        else if constexpr ((j) == 3) MFMA32(X.ar, W.Br[k], F.l[(k) & 1]);                                        \
        else if constexpr ((j) == 4) MFMA32(X.ag, W.Bg[k], F.l[(k) & 1]);                                        \
        else if constexpr ((j) == 5) MFMA32(X.az, W.Bz[k], F.l[(k) & 1]);                                        \
        else if constexpr ((j) == 6) MFMA32(X.ar, W.Lr[k], F.h[(k) & 1]);                                        \
        else if constexpr ((j) == 7) MFMA32(X.ag, W.Lg[k], F.h[(k) & 1]);                                        \
        else MFMA32(X.az, W.Lz[k], F.h[(k) & 1]);                                                                \
    }
#define M_D(i)                                                                                                   \
    if constexpr (DO_X) {                                                                                        \
        if constexpr ((i) == 0) MFMA16Z(X.dpl, D.a0, W.Bd_hi);                                                   \
        else if constexpr ((i) == 1) MFMA16(X.dpl, D.a1, W.Bd_hi);                                               \
        else if constexpr ((i) == 2) MFMA16(X.dpl, D.a0, W.Bd_lo);                                               \
        else if constexpr ((i) == 3) MFMA16(X.dpl, D.a1, W.Bd_lo);                                               \
        else if constexpr ((i) == 4) MFMA16(X.dpl, D.l0, W.Bd_hi);                                               \
        else MFMA16(X.dpl, D.l1, W.Bd_hi);                                                                       \
    }
        // fragments of k-step k of X's hidden tile (requested one k-step ahead)
#define PF(k)                                                                                                    \
    if constexpr (DO_X) {                                                                                        \
        F.h[(k) & 1] = lds16(X.hcur + frag_lane + 32 * (k));                                                     \
        F.l[(k) & 1] = lds16(X.lcur + frag_lane + 32 * (k));                                                     \
    }
        // Dense operands of X: h_{tx-1} of the wave's 32 units, window rows and their reverse complements (= the Average)
#define RDD                                                                                                      \
    if constexpr (DO_X) {                                                                                        \
        D.a0 = lds16(X.hcur + dense_lane); D.a1 = lds16(X.hcur + dense_lane + 16 * HS * 2);                      \
        D.l0 = lds16(X.lcur + dense_lane); D.l1 = lds16(X.lcur + dense_lane + 16 * HS * 2);                      \
        if (MODE == 2 && tx > 0 && (lane & 15) < X.ctx.nvalid)                                                   \
            split_avg_store(X.p, X.ctx.wg_w, tx - 1, UP, wave, D.a0, D.a1, D.l0, D.l1);                          \
    }
        // ---- Y's epilogue ------------------------------------------------------------------------------------
#define DS(i)                                                                                                    \
    if constexpr (DO_Y) Y.ctx.dpart[((size_t)(tf & 1) * 4 * NW + wave) * 64 + lane + (i) * NW * 64] = Y.dpl[i];
#define G(e, op) \
    if constexpr (DO_Y) split_gate_op<ONERCP, op>(gt[e], Y.ar[e], Y.ag[e], Y.az[e], Y.ax[e], Y.h[e]);
#define PB(g, op)                                                                                                \
    if constexpr (DO_Y) {                                                                                        \
        if constexpr ((op) == 0) {                                                                               \
            pbh[g][0] = split_state_h<ONERCP>(Y.h[4 * (g)]); pbh[g][1] = split_state_h<ONERCP>(Y.h[4 * (g) + 1]); \
            pbh[g][2] = split_state_h<ONERCP>(Y.h[4 * (g) + 2]); pbh[g][3] = split_state_h<ONERCP>(Y.h[4 * (g) + 3]); \
        } else if constexpr ((op) == 1) {                                                                        \
            pbhv[g] = split_pack4(pbh[g]);                                                                         \
        } else if constexpr ((op) == 2) {                                                                        \
            split_residual4(pbh[g], pbhv[g]);                                                                          \
        } else if constexpr ((op) == 3) {                                                                        \
            pblv[g] = split_pack4(pbh[g]);                                                                         \
        } else {                                                                                                 \
            *reinterpret_cast<uint2 *>(smem + Y.hnxt + pub_lane + 16 * (g)) = pbhv[g];                           \
            *reinterpret_cast<uint2 *>(smem + Y.lnxt + pub_lane + 16 * (g)) = pblv[g];                           \
        }                                                                                                        \
    }
        // h_t of Y is complete in LDS: meet the other waves, then flip Y's ping-pong
#define BAR                                                                                                      \
    if constexpr (DO_Y) {                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                          \
        unsigned sw_ = Y.hcur; Y.hcur = Y.hnxt; Y.hnxt = sw_;                                                    \
        sw_ = Y.lcur; Y.lcur = Y.lnxt; Y.lnxt = sw_;                                                             \
    }
#define RD0                                                                                                      \
    if constexpr (DO_Y) { Y.f0h = lds16(Y.hcur + frag_lane); Y.f0l = lds16(Y.lcur + frag_lane); }
#define XP(op)                                                                                                   \
    if constexpr (DO_Y) {                                                                                        \
        if constexpr ((op) == 0) xp_b = step_base(Y, ty + 1);                                                    \
        else Y.xa = onehot(xp_b);                                                                                \
    }
        // softmax + merge of step tf's logits for the wave's register (window = 4*(lane>>4) + wave, class = lane & 15)
#define FN(op)                                                                                                   \
    if constexpr (DO_Y) {                                                                                        \
        if constexpr ((op) == 0) {                                                                               \
            const float *dp_ = Y.ctx.dpart + ((size_t)(tf & 1) * 4 + wave) * NW * 64 + lane;                     \
            fs.d[0] = dp_[0]; fs.d[1] = dp_[64]; fs.d[2] = dp_[128]; fs.d[3] = dp_[192];                         \
        } else if constexpr ((op) == 1) {                                                                        \
            const float sum_ = ((fs.d[0] + fs.d[1]) + fs.d[2]) + fs.d[3];                                        \
            fs.lg = cls < C ? sum_ + fbias : -INFINITY;                                                          \
        } else if constexpr ((op) == 2) { if (MODE != 2) fs.m = row_max_ror<8>(fs.lg); }                         \
        else if constexpr ((op) == 3) { if (MODE != 2) fs.m = row_max_ror<4>(fs.m); }                            \
        else if constexpr ((op) == 4) { if (MODE != 2) fs.m = row_max_ror<2>(fs.m); }                            \
        else if constexpr ((op) == 5) { if (MODE != 2) fs.m = row_max_ror<1>(fs.m); }                            \
        else if constexpr ((op) == 6) { if (MODE != 2) fs.e = __builtin_amdgcn_exp2f(1.4426950408889634f * (fs.lg - fs.m)); } \
        else if constexpr ((op) == 7) { if (MODE != 2) fs.s = fs.e + row_ror<8>(fs.e); }                         \
        else if constexpr ((op) == 8) { if (MODE != 2) fs.s += row_ror<4>(fs.s); }                               \
        else if constexpr ((op) == 9) { if (MODE != 2) fs.s += row_ror<2>(fs.s); }                               \
        else if constexpr ((op) == 10) { if (MODE != 2) fs.s += row_ror<1>(fs.s); }                              \
        else if constexpr ((op) == 11) { if (MODE != 2) fs.e *= __builtin_amdgcn_rcpf(fs.s); }                   \
        else { if (Y.pr_on && tf >= 0) emit_value<MODE>(Y.p, Y.ctx, Y.p_off, Y.p_row0, tf, cls, MODE == 2 ? fs.lg : fs.e); } \
    }
#include "gru_split2_phase.inc"
#undef GAP
#undef M_IN
#undef M_K
#undef M_D
#undef PF
#undef RDD
#undef DS
#undef G
#undef PB
#undef BAR
#undef RD0
#undef XP
#undef FN
    };
    const std::true_type yes;
    const std::false_type no;

    // prologue: the operands of both tiles' step 0 (h_{-1} = 0 is in LDS), then tile 0's step 0 with nothing beside it
    S0.xa = onehot(step_base(S0, 0)); S0.f0h = lds16(S0.hcur + frag_lane); S0.f0l = lds16(S0.lcur + frag_lane);
    S1.xa = onehot(step_base(S1, 0)); S1.f0h = lds16(S1.hcur + frag_lane); S1.f0l = lds16(S1.lcur + frag_lane);
    asm volatile("s_nop 7" : "+v"(S0.xa));                        // VALU-written operand -> asm MFMA: wait states nobody else pads
    phase(yes, no, S0, S1, 0, 0);
    for (int t = 0; t + 1 < T; ++t) {
        phase(yes, yes, S1, S0, t, t);          // tile 1's step t      ||  tile 0 finishes step t
        phase(yes, yes, S0, S1, t + 1, t);      // tile 0's step t + 1  ||  tile 1 finishes step t
    }
    phase(yes, yes, S1, S0, T - 1, T - 1);
    phase(no, yes, S0, S1, T, T - 1);

    // drain: Dense and softmax/merge of the last step, image flush
    auto drain = [&](tile_state &Z) {
        const half8 a0 = lds16(Z.hcur + dense_lane), a1 = lds16(Z.hcur + dense_lane + 16 * HS * 2);
        const half8 l0 = lds16(Z.lcur + dense_lane), l1 = lds16(Z.lcur + dense_lane + 16 * HS * 2);
        if (MODE == 2 && (lane & 15) < Z.ctx.nvalid) split_avg_store(Z.p, Z.ctx.wg_w, T - 1, UP, wave, a0, a1, l0, l1);
        f32x4 d;
        MFMA16Z(d, a0, W.Bd_hi); MFMA16(d, a1, W.Bd_hi); MFMA16(d, a0, W.Bd_lo); MFMA16(d, a1, W.Bd_lo); MFMA16(d, l0, W.Bd_hi); MFMA16(d, l1, W.Bd_hi);
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(d));           // MFMA result -> VALU/LDS read, no compiler padding behind asm
        float *dw = Z.ctx.dpart + ((size_t)((T - 1) & 1) * 4 * NW + wave) * 64 + lane;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) dw[reg * NW * 64] = d[reg];
        __syncthreads();
        finish_register<NW, MODE>(Z.p, Z.ctx, T - 1, wave, fbias, Z.p_off, Z.p_row0);
        if (MODE == 0 && pin.ospan > 0) flush_image<NW>(Z.p, Z.ctx);
    };
    drain(S0);
    drain(S1);
}

template <int MODE, bool ONERCP>
int launch_split2(const gru_params &p, int64_t groups, int half_bytes, hipStream_t stream)
{
    static bool configured = false;
    if (!configured) {
        DGRP_HIP(hipFuncSetAttribute((const void *)gru_split2_kernel<MODE, ONERCP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
    }
    hipLaunchKernelGGL((gru_split2_kernel<MODE, ONERCP>), dim3((unsigned)((groups + 1) / 2)), dim3(256), (size_t)2 * half_bytes, stream, p, half_bytes);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

}  // namespace

int dgrp_split2_launch(const gru_params &p, int64_t groups, int half_bytes, bool onercp, hipStream_t stream)
{
    if (onercp)
        return p.mode == 0 ? launch_split2<0, true>(p, groups, half_bytes, stream)
             : p.mode == 1 ? launch_split2<1, true>(p, groups, half_bytes, stream) : launch_split2<2, true>(p, groups, half_bytes, stream);
    return p.mode == 0 ? launch_split2<0, false>(p, groups, half_bytes, stream)
         : p.mode == 1 ? launch_split2<1, false>(p, groups, half_bytes, stream) : launch_split2<2, false>(p, groups, half_bytes, stream);
}

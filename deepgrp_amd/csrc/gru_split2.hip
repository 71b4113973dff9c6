// A4 (+A5/A6 fused), split operands, the 128-unit class (97-128 units): the default forward kernel of the benchmark model.
//
// Same mathematics as gru_split_kernel (gru_kernel.hip): U = U_hi + U_lo and h_{t-1} = h_hi + h_lo as fp16 pairs,
// U.h ~ U_hi.h_hi + U_hi.h_lo + U_lo.h_hi on the matrix cores with fp32 accumulation, the gate chain of gru_shared.h link
// for link.  What differs is the MFMA shape, where everything lives and when it is issued:
//   * v_mfma_f32_16x16x32_f16 (144 + 6 per wave and step).  The kernel runs at the chip's POWER wall, not at an issue or
//     pipe limit: the same instruction stream on all-zero operands takes 23 % less time, and at equal cycles per flop the
//     16x16x32 shape sustains 19 % more flop/s on random data than 32x32x16 (tools/ubench/mfma_shape_power.hip;
//     MI355X_MICROARCH.md, "DVFS give-back", item 7).  Tile: weights are the A operand (rows = units), the hidden tile the
//     B operand (columns = recurrent rows), D[unit][row]: a lane holds 4 consecutive units of ONE window's forward row and of
//     its reverse-complement row (sub-tiles: unit halves x row halves), so the gate math stays lane-local and a publish is
//     an 8-byte store per sub-tile.
//   * ONE wave per SIMD with the 512-register budget.  The 50 weight fragments of the wave's 32 units (U_hi, U_lo, Dense:
//     200 registers) are loaded ONCE, straight into the accumulation half of the register file (asm loads with AGPR
//     destinations), and every MFMA names them there: no copies, nothing streams, and the compiler's 256 architectural
//     VGPRs stay free for two tiles' accumulators and state.
//   * The input projection is not an MFMA: inputs are one-hot, so x.W + b is one of five rows.  The rows (exp2 domain,
//     both biases folded) sit in a 10 KB LDS table; a step's accumulators START as the table rows of the two bases of the
//     lane's window (forward and complemented reverse), read as the MFMA chain's C operand.
//   * TWO row tiles (2 x 16 windows) per workgroup, software-pipelined against each other: a "phase" is the 150 MFMAs of
//     one tile's step (X) with the whole epilogue of the other tile's step (Y) -- gate chains, fp16 hi/lo publish, the
//     workgroup barrier, table rows and first fragments of Y's next step -- and the softmax + max-merge of X's own logits of
//     two steps ago (stored while X was the epilogue tile, a barrier since: nothing in this phase feeds them), cut
//     into single operations and dropped into the gaps BETWEEN X's MFMAs (a wave issues in order; an MFMA holds the issue
//     port for 8 of its 16 cycles).  The barrier itself sits inside X's MFMA stream, four MFMAs from its end.
//   * The order of that interleave is generated (tools/gen_split2_schedule.py -> gru_split2_phase.inc) and pinned with
//     sched_barrier between gaps; the MFMAs are asm volatile, which the compiler keeps in program order.
//
// Inline-asm obligations (cdna_hip_programming.md 5.7) and how they are met -- by construction, and CHECKED on the compiler's
// output at every build (tools/lint_split2_isa.py, run by the Makefile): the register allocator is free to put a v_mov copy of an
// accumulator directly in front of the MFMA that takes it as C, and did so for one schedule variant (wrong results, no fault):
//   - an MFMA's result is read by compiler-scheduled code only many MFMAs later (the generator keeps the first gaps of a
//     phase free of anything that touches the previous phase's accumulators or Dense result);
//   - no MFMA operand is written by VALU code: fragments, table rows and Dense operands come straight from LDS reads, which
//     are the compiler's own loads (it waits for them in front of the asm statement);
//   - the weight loads carry their own s_waitcnt inside the statement that issues them.
#include "gru_shared.h"
#include <mutex>
#include <type_traits>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// experiment switch (never set in the product build): 1 drops the U_lo.h_hi pass, 2 the U_hi.h_lo pass -- what a two-pass
// split would cost in accuracy (tools/twopass_probe.py)
#ifndef DGRP_SPLIT_DROP
#define DGRP_SPLIT_DROP 0
#endif
// order of the 36 MFMAs of a k-step: 0 = pass, gate, unit half, row half (two in a row share the A fragment); 1 = pass, row half, gate, unit
// half (six share B); 2 = gate, unit half, pass, row half (four in a row share U_hi, then two U_lo; an accumulator's passes two MFMAs apart)
#ifndef DGRP_MK_ORDER
#define DGRP_MK_ORDER 0
#endif

namespace {

constexpr int NW = 4, UP = 128, KS = 4, HPAD = 16, HS = UP + HPAD;   // k-steps of 32; row pitch 144 halves: conflict-free ds_read_b128
constexpr int XT_PITCH = 4 * 128 * 4 + 32;                            // table row of one base: 4 kinds x 128 units fp32 + 32 B (bank spread)

struct split2_weights {                   // 50 fragments = 200 AGPRs per lane, resident for the whole kernel
    u32x4 hi[3][KS][2], lo[3][KS][2];     // [gate r, g, z][k-step][unit half]: U_hi, U_lo
    u32x4 Bd_hi, Bd_lo;                   // Dense
};

#define LOAD2(a, pa, b, pb)                                                                                             \
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(0)"            \
                 : "=&a"(a), "=&a"(b) : "v"(pa), "v"(pb) : "memory")

// recurrent MFMA: A = weights (AGPR), B = hidden fragment; Dense MFMA: A = hidden rows, B = weights (AGPR)
#define MFMA_R(acc, Wf, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(Wf), "v"(b))
#define MFMA_D(acc, a, Wf) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(Wf))
#define MFMA_DZ(acc, a, Wf) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(Wf))

struct tile_state {
    gru_params p;                         // batched records: the two tiles may belong to different records
    wg_ctx ctx;
    unsigned hcur, hnxt, lcur, lnxt;      // byte offsets of the hidden tiles (hi, lo; ping-pong) in the dynamic LDS
    unsigned myseq;                       // LDS byte offset of the class indices of window (lane & 15)
    float h[16];                          // gate state of the lane's 16 (row, unit) pairs: h, or h - 1 (ONERCP); element 4*sub + i
    f32x4 ar[4], ag[4], az[4], ax[4];     // pre-activations, sub-tile = 2 * unit half + row half (ax: the candidate's input projection)
    f32x4 dpl;                            // Dense partial logits of the previous step
    unsigned tabf, tabr;                  // LDS byte offsets of the table rows of the step in flight: forward base, complemented reverse base
    half8 f0h[2], f0l[2];                 // first fragments (k-step 0, row halves) of the NEXT contraction
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
        const uint4 *p16 = pin.pack16 + (size_t)wave * 48 * 64 + lane;        // [(pass*3 + gate)*8 + kstep*2 + unit half][64]
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                LOAD2(W.hi[g][ks][0], p16 + (size_t)(g * 8 + ks * 2) * 64, W.hi[g][ks][1], p16 + (size_t)(g * 8 + ks * 2 + 1) * 64);
                LOAD2(W.lo[g][ks][0], p16 + (size_t)(24 + g * 8 + ks * 2) * 64, W.lo[g][ks][1], p16 + (size_t)(24 + g * 8 + ks * 2 + 1) * 64);
            }
        const uint4 *mypack = pin.pack + (size_t)wave * pin.nfrag * 64 + lane;    // Dense fragments of the 32x32 pack (api.hip): same shape
        LOAD2(W.Bd_hi, mypack + (size_t)(3 * (2 * KS + 1) + 1) * 64, W.Bd_lo, mypack + (size_t)(3 * (2 * KS + 1) + 2) * 64);
    }
    // the input-projection table: [5 bases][4 kinds][128 units] fp32 -> LDS rows of XT_PITCH bytes
    for (int i = tid; i < 5 * 512; i += 256)
        *reinterpret_cast<float *>(smem + pin.xtab_off + (i / 512) * XT_PITCH + (i % 512) * 4) = pin.xtab[i];

    const int wi = lane & 15, q4 = lane >> 4;                     // window of this lane; its k-group / unit group
    const int cls = lane & 15;
    const float fbias = cls < C ? pin.ffb[cls] : 0.0f;
    const int pwi = 4 * (lane >> 4) + wave;                       // window of the wave's logit register
    const unsigned frag_lane = (unsigned)(wi * HS + 8 * q4) * 2;                            // fragment: row wi (+16: its reverse complement), k 8 q4 ..
    const unsigned dense_lane = (unsigned)(wi * HS + 32 * wave + 8 * q4) * 2;
    const unsigned pub_lane = (unsigned)(wi * HS + 32 * wave + 4 * q4) * 2;                 // publish: 4 units 32 w + 16 uh + 4 q4 ..
    const unsigned tab_lane = (unsigned)pin.xtab_off + (unsigned)(32 * wave + 4 * q4) * 4;  // table: the same 4 units, + 64 B per unit half

    tile_state S0, S1;                      // two named objects, never indexed: they must stay in registers
    auto setup = [&](tile_state &Z, int x) {
        unsigned char *base = smem + (size_t)x * half_bytes;
        _Float16 *lbuf = reinterpret_cast<_Float16 *>(base + pin.lo_tile_off);
        for (int i = tid; i < 32 * HS; i += 256) lbuf[i] = (_Float16)0.0f;
        Z.p = pin;
        const int64_t bid = wg_record_at<MODE>(pin, Z.p, 2 * (int64_t)blockIdx.x + x);
        Z.ctx = wg_setup<NW, MODE, HPAD>(Z.p, base, bid);                              // ends with a barrier
        Z.hcur = (unsigned)x * half_bytes;                        // hbuf is the first item of the carve
        Z.hnxt = Z.hcur + 32 * HS * 2;
        Z.lcur = (unsigned)x * half_bytes + pin.lo_tile_off;
        Z.lnxt = Z.lcur + 32 * HS * 2;
        Z.myseq = (unsigned)(Z.ctx.seqs - smem) + wi * pin.Tp;
#pragma unroll
        for (int i = 0; i < 16; ++i) Z.h[i] = ONERCP ? -1.0f : 0.0f;
        Z.p_off = Z.ctx.rowoff[pwi];
        Z.p_row0 = Z.ctx.row0s[pwi];
        Z.pr_on = cls < C && pwi < Z.ctx.nvalid;
    };
    setup(S0, 0);
    setup(S1, 1);

    auto lds16 = [&](unsigned off) -> half8 { return *reinterpret_cast<const half8 *>(smem + off); };
    auto ldsf4 = [&](unsigned off) -> f32x4 { return *reinterpret_cast<const f32x4 *>(smem + off); };
    // table rows of step tn for this lane's window: forward base seq[tn]; reverse strand base comp(seq[T-1-tn]), complement
    // table [3,2,1,0,4] (model.py:233-237)
    auto tab_rows = [&](tile_state &Z, uint32_t bf, uint32_t br) {
        bf = bf < 4 ? bf : 4;                                     // (the step behind the last one reads a byte outside the window)
        br = br < 4 ? 3 - br : 4;
        Z.tabf = tab_lane + bf * XT_PITCH;
        Z.tabr = tab_lane + br * XT_PITCH;
    };
    // accumulators of the next step start as the table rows (kinds 0 r, 1 g, 2 z); sub-tile = 2 * unit half + row half
    auto acc_init = [&](tile_state &Z, int g, int sub) {
        const f32x4 v = ldsf4(((sub & 1) ? Z.tabr : Z.tabf) + g * 512 + (sub >> 1) * 64);
        if (g == 0) Z.ar[sub] = v; else if (g == 1) Z.ag[sub] = v; else Z.az[sub] = v;
    };

#ifdef DGRP_STAMP
    // diagnostic build: cycles per phase section (ST(2) phase start .. ST(0) in front of the barrier .. ST(1) behind it)
    uint32_t stamp_acc[3] = { 0, 0, 0 };
    uint64_t stamp_prev = 0;
#endif
    // ---- one phase: tile X's step tx on the matrix pipe, tile Y's epilogue of step ty in the gaps --------------------
    struct frag_ring { half8 h[2][2], l[2][2]; };                 // [k-step parity][row half]
    struct dense_ops { half8 a0, a1, l0, l1; };
    struct fin_state { float d[NW], lg, m, e, s; };
    auto phase = [&](auto do_x, auto do_y, auto do_f, tile_state &X, tile_state &Y, int tx, int ty) __attribute__((always_inline)) {
        constexpr bool DO_X = decltype(do_x)::value, DO_Y = decltype(do_y)::value, DO_F = decltype(do_f)::value;
        frag_ring F;
        dense_ops D;
        fin_state fs;
        split_gate_tmp gt[16];
        float pbh[4][4];
        uint2 pbhv[4], pblv[4];
        uint32_t xp_bf = 4, xp_br = 4;
        const int td = ty - 1;                                    // step whose Dense partials Y stores in this phase
        const int tf = tx - 2;                                    // step whose logits X finishes in this phase (stored one phase ago, a barrier since)
        if constexpr (DO_X) { F.h[0][0] = X.f0h[0]; F.h[0][1] = X.f0h[1]; F.l[0][0] = X.f0l[0]; F.l[0][1] = X.f0l[1]; }
#define GAP __builtin_amdgcn_sched_barrier(0);
#ifdef DGRP_STAMP
#define ST(i) if constexpr (DO_X && DO_Y) { __builtin_amdgcn_sched_barrier(0); const uint64_t now_ = __builtin_amdgcn_s_memtime(); \
        stamp_acc[i] += (uint32_t)(now_ - stamp_prev); stamp_prev = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define ST(i)
#endif
        // MFMA n (0..35) of k-step ks: pass n / 12 (hi.h_hi, hi.h_lo, lo.h_hi), then gate, unit half, row half
#define M_K(ks, n)                                                                                               \
    if constexpr (DO_X && !(DGRP_MK_ORDER != 2 && DGRP_SPLIT_DROP == 1 && (n) >= 24) && !(DGRP_MK_ORDER != 2 && DGRP_SPLIT_DROP == 2 && (n) >= 12 && (n) < 24)) { \
        constexpr int ps_ = DGRP_MK_ORDER == 2 ? ((n) % 6) / 2 : (n) / 12,                                                          \
                      g_ = DGRP_MK_ORDER == 0 ? ((n) % 12) / 4 : DGRP_MK_ORDER == 2 ? (n) / 12 : ((n) % 6) / 2,                    \
                      uh_ = DGRP_MK_ORDER == 0 ? ((n) % 4) / 2 : DGRP_MK_ORDER == 2 ? ((n) / 6) % 2 : (n) % 2,                     \
                      rh_ = DGRP_MK_ORDER == 1 ? ((n) % 12) / 6 : (n) % 2, sub_ = 2 * uh_ + rh_;                                   \
        const half8 &b_ = ps_ == 1 ? F.l[(ks) & 1][rh_] : F.h[(ks) & 1][rh_];                                    \
        const u32x4 &w_ = ps_ == 2 ? W.lo[g_][ks][uh_] : W.hi[g_][ks][uh_];                                      \
        if constexpr (g_ == 0) MFMA_R(X.ar[sub_], w_, b_);                                                       \
        else if constexpr (g_ == 1) MFMA_R(X.ag[sub_], w_, b_);                                                  \
        else MFMA_R(X.az[sub_], w_, b_);                                                                         \
    }
#define M_D(i)                                                                                                   \
    if constexpr (DO_X) {                                                                                        \
        if constexpr ((i) == 0) MFMA_DZ(X.dpl, D.a0, W.Bd_hi);                                                   \
        else if constexpr ((i) == 1) MFMA_D(X.dpl, D.a1, W.Bd_hi);                                               \
        else if constexpr ((i) == 2) MFMA_D(X.dpl, D.a0, W.Bd_lo);                                               \
        else if constexpr ((i) == 3) MFMA_D(X.dpl, D.a1, W.Bd_lo);                                               \
        else if constexpr ((i) == 4) MFMA_D(X.dpl, D.l0, W.Bd_hi);                                               \
        else MFMA_D(X.dpl, D.l1, W.Bd_hi);                                                                       \
    }
        // fragments of k-step ks of X's hidden tile (requested one k-step ahead): both row halves, hi and lo
#define PF(ks)                                                                                                   \
    if constexpr (DO_X) {                                                                                        \
        F.h[(ks) & 1][0] = lds16(X.hcur + frag_lane + 64 * (ks)); F.h[(ks) & 1][1] = lds16(X.hcur + frag_lane + 16 * HS * 2 + 64 * (ks)); \
        F.l[(ks) & 1][0] = lds16(X.lcur + frag_lane + 64 * (ks)); F.l[(ks) & 1][1] = lds16(X.lcur + frag_lane + 16 * HS * 2 + 64 * (ks)); \
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
        // the candidate's input projection of the step being finished (table kind 3), sub-tile by sub-tile
#define AXL(sub) \
    if constexpr (DO_Y) Y.ax[sub] = ldsf4((((sub) & 1) ? Y.tabr : Y.tabf) + 3 * 512 + ((sub) >> 1) * 64);
#define DS(i)                                                                                                    \
    if constexpr (DO_Y) Y.ctx.dpart[((size_t)(td & 1) * 4 * NW + wave) * 64 + lane + (i) * NW * 64] = Y.dpl[i];
#define G(e, op) \
    if constexpr (DO_Y) split_gate_op<ONERCP, op>(gt[e], Y.ar[(e) / 4][(e) % 4], Y.ag[(e) / 4][(e) % 4], Y.az[(e) / 4][(e) % 4], Y.ax[(e) / 4][(e) % 4], Y.h[e]);
        // publish sub-tile g = 2 * unit half + row half: 4 consecutive units of row 16 rh + (lane & 15)
#define PB(g, op)                                                                                                \
    if constexpr (DO_Y) {                                                                                        \
        if constexpr ((op) == 0) {                                                                               \
            pbh[g][0] = split_state_h<ONERCP>(Y.h[4 * (g)]); pbh[g][1] = split_state_h<ONERCP>(Y.h[4 * (g) + 1]); \
            pbh[g][2] = split_state_h<ONERCP>(Y.h[4 * (g) + 2]); pbh[g][3] = split_state_h<ONERCP>(Y.h[4 * (g) + 3]); \
        } else if constexpr ((op) == 1) {                                                                        \
            pbhv[g] = split_pack4(pbh[g]);                                                                       \
        } else if constexpr ((op) == 2) {                                                                        \
            split_residual4(pbh[g], pbhv[g]);                                                                    \
        } else if constexpr ((op) == 3) {                                                                        \
            pblv[g] = split_pack4(pbh[g]);                                                                       \
        } else {                                                                                                 \
            *reinterpret_cast<uint2 *>(smem + Y.hnxt + pub_lane + ((g) & 1) * 16 * HS * 2 + ((g) >> 1) * 32) = pbhv[g]; \
            *reinterpret_cast<uint2 *>(smem + Y.lnxt + pub_lane + ((g) & 1) * 16 * HS * 2 + ((g) >> 1) * 32) = pblv[g]; \
        }                                                                                                        \
    }
        // h_t of Y is complete in LDS: meet the other waves, then flip Y's ping-pong
#define BAR                                                                                                      \
    if constexpr (DO_Y) {                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                          \
        unsigned sw_ = Y.hcur; Y.hcur = Y.hnxt; Y.hnxt = sw_;                                                    \
        sw_ = Y.lcur; Y.lcur = Y.lnxt; Y.lnxt = sw_;                                                             \
    }
        // Y's next step: first fragments, the two bases of the lane's window and their table rows, the accumulators' start
#define RD0                                                                                                      \
    if constexpr (DO_Y) {                                                                                        \
        Y.f0h[0] = lds16(Y.hcur + frag_lane); Y.f0h[1] = lds16(Y.hcur + frag_lane + 16 * HS * 2);                \
        Y.f0l[0] = lds16(Y.lcur + frag_lane); Y.f0l[1] = lds16(Y.lcur + frag_lane + 16 * HS * 2);                \
    }
#define XP(op)                                                                                                   \
    if constexpr (DO_Y) {                                                                                        \
        if constexpr ((op) == 0) { xp_bf = smem[Y.myseq + ty + 1]; xp_br = smem[Y.myseq + T - 2 - ty]; }                       \
        else tab_rows(Y, xp_bf, xp_br);                                                                          \
    }
#define CI(g, sub) \
    if constexpr (DO_Y) acc_init(Y, g, sub);
        // softmax + merge of step tf's logits of tile X (its partials were stored while X was the epilogue tile, one phase and one
        // barrier ago) for the wave's register (window = 4*(lane>>4) + wave, class = lane & 15)
#define FN(op)                                                                                                   \
    if constexpr (DO_F) {                                                                                        \
        if constexpr ((op) == 0) {                                                                               \
            const float *dp_ = X.ctx.dpart + ((size_t)(tf & 1) * 4 + wave) * NW * 64 + lane;                     \
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
        else { if (X.pr_on && tf >= 0) emit_value<MODE>(X.p, X.ctx, X.p_off, X.p_row0, tf, cls, MODE == 2 ? fs.lg : fs.e); } \
    }
#include "gru_split2_phase.inc"
#undef GAP
#undef ST
#undef M_K
#undef M_D
#undef PF
#undef RDD
#undef AXL
#undef DS
#undef G
#undef PB
#undef BAR
#undef RD0
#undef XP
#undef CI
#undef FN
    };
    const std::true_type yes;
    const std::false_type no;

    // prologue: the operands of both tiles' step 0 (h_{-1} = 0 is in LDS), then tile 0's step 0 with nothing beside it
    auto first_step = [&](tile_state &Z) {
        tab_rows(Z, smem[Z.myseq], smem[Z.myseq + T - 1]);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) acc_init(Z, g, sub);
        Z.f0h[0] = lds16(Z.hcur + frag_lane); Z.f0h[1] = lds16(Z.hcur + frag_lane + 16 * HS * 2);
        Z.f0l[0] = lds16(Z.lcur + frag_lane); Z.f0l[1] = lds16(Z.lcur + frag_lane + 16 * HS * 2);
    };
    first_step(S0);
    first_step(S1);
    phase(yes, no, no, S0, S1, 0, 0);
#ifdef DGRP_STAMP
    stamp_prev = __builtin_amdgcn_s_memtime();
    const uint64_t stamp_t0 = stamp_prev, stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // Register copies the allocator places on a control-flow edge (loop entry, back edge, exit) are VALU reads it does not know to
    // keep away from the asm MFMAs in front of them: the last MFMAs of a phase have to be complete before the edge.
#define EDGE_PAD do { asm volatile("s_nop 15\n\ts_nop 7"); __builtin_amdgcn_sched_barrier(0); } while (0)
    EDGE_PAD;
    for (int t = 0; t + 1 < T; ++t) {
        phase(yes, yes, yes, S1, S0, t, t);          // tile 1's step t      ||  tile 0 finishes step t
        phase(yes, yes, yes, S0, S1, t + 1, t);      // tile 0's step t + 1  ||  tile 1 finishes step t
        EDGE_PAD;
    }
#ifdef DGRP_STAMP
    if (pin.stamps && lane == 0) {
        uint64_t *o = pin.stamps + ((size_t)blockIdx.x * NW + wave) * 8;
        o[0] = stamp_acc[0]; o[1] = stamp_acc[1]; o[2] = stamp_acc[2];
        o[3] = __builtin_amdgcn_s_memtime() - stamp_t0;
        o[4] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
    }
#endif
    phase(yes, yes, yes, S1, S0, T - 1, T - 1);
    phase(no, yes, yes, S0, S1, T, T - 1);      // (tile 0's logits of step T - 2 are finished here; tile 1's in its drain)

    // drain: Dense and softmax/merge of the last step, image flush
    auto drain = [&](tile_state &Z, bool prev_open) {
        const half8 a0 = lds16(Z.hcur + dense_lane), a1 = lds16(Z.hcur + dense_lane + 16 * HS * 2);
        const half8 l0 = lds16(Z.lcur + dense_lane), l1 = lds16(Z.lcur + dense_lane + 16 * HS * 2);
        if (MODE == 2 && (lane & 15) < Z.ctx.nvalid) split_avg_store(Z.p, Z.ctx.wg_w, T - 1, UP, wave, a0, a1, l0, l1);
        f32x4 d;
        MFMA_DZ(d, a0, W.Bd_hi); MFMA_D(d, a1, W.Bd_hi); MFMA_D(d, a0, W.Bd_lo); MFMA_D(d, a1, W.Bd_lo); MFMA_D(d, l0, W.Bd_hi); MFMA_D(d, l1, W.Bd_hi);
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(d));           // MFMA result -> VALU/LDS read, no compiler padding behind asm
        float *dw = Z.ctx.dpart + ((size_t)((T - 1) & 1) * 4 * NW + wave) * 64 + lane;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) dw[reg * NW * 64] = d[reg];
        __syncthreads();
        if (prev_open && T > 1) finish_register<NW, MODE>(Z.p, Z.ctx, T - 2, wave, fbias, Z.p_off, Z.p_row0);
        finish_register<NW, MODE>(Z.p, Z.ctx, T - 1, wave, fbias, Z.p_off, Z.p_row0);
        if (MODE == 0 && pin.ospan > 0) flush_image<NW>(Z.p, Z.ctx);
    };
    drain(S0, false);
    drain(S1, true);                            // tile 1 stored its partials of step T - 2 in the last phase: no phase of its own follows
}

template <int MODE, bool ONERCP>
int launch_split2(const gru_params &p, int64_t groups, int half_bytes, hipStream_t stream)
{
    static std::once_flag configured;        // (records run on a pool of host threads: an unsynchronised flag was a data race)
    static hipError_t cfg_err = hipSuccess;
    std::call_once(configured, [] { cfg_err = hipFuncSetAttribute((const void *)gru_split2_kernel<MODE, ONERCP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    DGRP_HIP(cfg_err);
#ifdef DGRP_STAMP
    if (const char *dump = getenv("DGRP_STAMP_DUMP")) {
        gru_params q = p;
        const size_t bytes = (size_t)((groups + 1) / 2) * NW * 8 * 8;
        DGRP_HIP(hipMalloc((void **)&q.stamps, bytes));
        DGRP_HIP(hipMemset(q.stamps, 0, bytes));
        hipLaunchKernelGGL((gru_split2_kernel<MODE, ONERCP>), dim3((unsigned)((groups + 1) / 2)), dim3(256), (size_t)2 * half_bytes + 5 * XT_PITCH, stream, q, half_bytes);
        DGRP_HIP(hipStreamSynchronize(stream));
        std::vector<uint64_t> hst(bytes / 8);
        DGRP_HIP(hipMemcpy(hst.data(), q.stamps, bytes, hipMemcpyDeviceToHost));
        if (FILE *f = fopen(dump, "wb")) { fwrite(hst.data(), 1, bytes, f); fclose(f); }
        (void)hipFree(q.stamps);
        return DGRP_OK;
    }
#endif
    hipLaunchKernelGGL((gru_split2_kernel<MODE, ONERCP>), dim3((unsigned)((groups + 1) / 2)), dim3(256), (size_t)2 * half_bytes + 5 * XT_PITCH, stream, p, half_bytes);
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

// A4 (+A5/A6 fused), split operands, GRU models up to 64 units -- the sizes of the reference's own models (defaults.toml: 60 units,
// window 342, attention; its hyper-parameter search draws gru_units ~ qnormal(34, 5, 2), notebooks/DeepGRP.ipynb:153-154).
//
// Same mathematics and the same gate chain (gru_shared.h, link for link) as gru_split2_kernel: U = U_hi + U_lo and
// h_{t-1} = h_hi + h_lo as fp16 pairs, U.h ~ U_hi.h_hi + U_hi.h_lo + U_lo.h_hi on v_mfma_f32_16x16x32_f16 with fp32 accumulation,
// input projection as a table row the accumulators start from, weights resident in AGPRs, two row tiles per wave software-pipelined
// against each other with a generated interleave (tools/gen_wave_schedule.py -> gru_wave_phase_nu*.inc).  What differs is the cut:
//
//   * a row tile is 8 windows x 2 strands = 16 rows = ONE MFMA column block, and a wave owns ALL units of its tiles (NU = ceil(u / 16)
//     unit groups of 16: a 36-unit model computes 48 units, not 64).  With <= 64 units the whole recurrent matrix as hi + lo
//     fragments is <= 192 registers per lane -- what a wave of gru_split2 holds for its 32 of 128 units.  Nothing is shared between
//     waves any more: the hidden tile a wave publishes is read back by that wave only, the Dense product completes inside the wave,
//     so the time loop has NO barrier (the one-tile kernel this replaces met its partner wave on another SIMD at every step, and the
//     convoy kept the matrix phases and the gate phases of a SIMD's waves aligned: matrix pipe 41 % busy, each wave issuing a third
//     of the time).  A workgroup is four such waves (one per SIMD, 512-register budget) that share nothing but the input table.
//   * D[unit][row]: a lane holds 4 consecutive units per unit group of ONE row (window lane & 7, strand (lane >> 3) & 1), so the gate
//     math is lane-local and a publish is one 16-byte store per PAIR of unit groups and half (the hidden tile's columns are permuted so
//     that a lane's units of two groups lie side by side; the packed weights follow the permutation).  One LDS tile per row tile and half, no
//     ping-pong: a tile's fragments are read in its MFMA phase and rewritten in its epilogue phase, in program order of one wave.
//   * Dense: the hidden tile's B fragments of the recurrent MFMAs ARE the A fragments of the Dense product (same lane map), so the
//     Dense layer costs 3 MFMAs per k-step and no LDS read; the result holds the forward rows' and the reverse-complement rows'
//     partial logits in opposite half-waves, and the Average is one v_permlane32_swap + add per register pair.
//   * attention pre-pass: avg[t] = (h_fwd + h_rc) / 2 comes from the fp32 state at publish time (the partner strand sits 8 lanes away
//     in the same DPP row), 16 bytes per lane and unit group.
//
// Inline-asm obligations as in gru_split2.hip; the compiler's output is audited at every build (tools/lint_split2_isa.py).
#include "gru_shared.h"
#include <mutex>
#include <type_traits>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WPAD = 16;                                       // row pitch KP + 16 halves: conflict-free ds_read_b128 (8 lanes per pass)

template <int NU> struct wave_cfg {
    static constexpr int UP16 = 16 * NU, KS = (NU + 1) / 2, KP = 32 * KS, HS = KP + WPAD;
    static constexpr int XT_PITCH = 4 * UP16 * 4 + 32;         // table row of one base: 4 kinds x UP16 units fp32 + 32 B (bank spread)
    static constexpr int TILE_BYTES = 16 * HS * 2;             // one half (hi or lo) of one row tile
};

template <int NU> struct wave_weights {                        // 6 KS NU + 2 KS fragments, resident in AGPRs for the whole kernel
    u32x4 hi[3][wave_cfg<NU>::KS][NU], lo[3][wave_cfg<NU>::KS][NU];   // [gate r, g, z][k-step][unit group]
    u32x4 Bd_hi[wave_cfg<NU>::KS], Bd_lo[wave_cfg<NU>::KS];           // Dense
};

#define WLOAD2(a, pa, b, pb)                                                                                            \
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(0)"            \
                 : "=&a"(a), "=&a"(b) : "v"(pa), "v"(pb) : "memory")

// Accumulators are tied ("+v"): untied (v_mfma D, A, B, C with a free D) the registers a chain leaves behind are reused by VALU code
// within the MFMA's latency (write-after-write against a result still in flight).  Tied, the allocator may put a v_mov copy of an
// accumulator directly in front of an MFMA; hipcc pads no wait states around inline asm, so the build patches them in
// (tools/lint_split2_isa.py --fix, deepgrp_amd/csrc/Makefile).
#define WMFMA_R(acc, Wf, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(Wf), "v"(b))
#define WMFMA_D(acc, a, Wf) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(Wf))
#define WMFMA_DZ(acc, a, Wf) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(Wf))

constexpr int P_OFF = -2, P_GLOBAL = -1;                       // placement of a lane's logit: nothing to emit / straight to HBM (mode 0)

template <int NU> struct wave_tile {
    unsigned hbuf, lbuf;                  // LDS byte offsets of the tile's hi and lo halves
    unsigned seq_at;                      // LDS byte offset of the base of the step in flight (forward rows walk up, rc rows down)
    unsigned tab;                         // LDS byte offset of the table row of the step in flight
    float h[4 * NU];                      // gate state of the lane's elements: h, or h - 1 (ONERCP); element 4 * ug + i
    f32x4 ar[NU], ag[NU], az[NU], ax[NU]; // pre-activations per unit group (ax: the candidate's input projection)
    f32x4 dpl;                            // Dense result of the last MFMA phase: partial logits, rows = tile rows, columns = classes
    half8 f0h, f0l;                       // first fragments (k-step 0) of the NEXT contraction
    int p_off[2];                         // per logit chain q: element offset of (window, class) in the image / output, P_GLOBAL or P_OFF
    int avg_off;                          // attention pre-pass: element offset of (window, 4 units of group 0) in the spill, or -1
    int wbase;                            // first window of the tile inside the wave's group (0 or 8)
};

struct wave_ctx {                         // per wave: its group of 16 windows
    gru_params p;
    uint8_t *seqs;
    int64_t *row0s;
    int *rowoff;
    unsigned *obuf;
    int64_t wg_w, lo;
    int nvalid;
};

template <int NU, int MODE, bool ONERCP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) gru_wave_kernel(const gru_params pin, int64_t total_groups, int wave_bytes)
{
    using cfg = wave_cfg<NU>;
    constexpr int KS = cfg::KS, HS = cfg::HS, UP16 = cfg::UP16, XT_PITCH = cfg::XT_PITCH, TILE_BYTES = cfg::TILE_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = pin.T, C = pin.C;

    wave_weights<NU> W;
    {
        const uint4 *pw = pin.packw + lane;                       // [((pass * 3 + gate) * KS + k-step) * NU + unit group][64], then Dense
        constexpr int NF = 6 * KS * NU;
        static_assert(NF % 2 == 0, "fragments are loaded in pairs");
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int ug = 0; ug < NU; ++ug)
                    WLOAD2(W.hi[g][ks][ug], pw + (size_t)((g * KS + ks) * NU + ug) * 64,
                           W.lo[g][ks][ug], pw + (size_t)(((3 + g) * KS + ks) * NU + ug) * 64);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            WLOAD2(W.Bd_hi[ks], pw + (size_t)(NF + 2 * ks) * 64, W.Bd_lo[ks], pw + (size_t)(NF + 2 * ks + 1) * 64);
    }
    // the input-projection table: [5 bases][4 kinds][UP16 units] fp32 -> LDS rows of XT_PITCH bytes (the one thing the waves share)
    for (int i = tid; i < 5 * 4 * UP16; i += 256)
        *reinterpret_cast<float *>(smem + (i / (4 * UP16)) * XT_PITCH + (i % (4 * UP16)) * 4) = pin.xtabw[i];
    __syncthreads();                                              // the only barrier of the kernel
    const int64_t group = 4 * (int64_t)blockIdx.x + wave;
    if (group >= total_groups) return;

    const int row = lane & 15, q4 = lane >> 4, win = row & 7;
    const bool rc = (row & 8) != 0;
    const int cls = lane & 15;
    const float fbias = cls < C ? pin.ffb[cls] : 0.0f;
    const unsigned frag_lane = (unsigned)(row * HS + 8 * q4) * 2;       // fragment: tile row `row`, k = 8 q4 .. (+ 64 B per k-step)
    // publish: the lane's 4 units of unit group 2 j and its 4 units of group 2 j + 1 go side by side into columns 32 j + 8 q4 .. + 7 --
    // the 16 bytes its own k-step-j fragment reads (ONE ds_write_b128 per pair of groups and half instead of two ds_write_b64 that
    // were 4-way bank conflicts at this row pitch: 12 % of the kernel's cycles in SQ_LDS_BANK_CONFLICT).  The recurrent and Dense
    // fragments are packed in that column order (api.hip, dgrp_wave_col).
    const unsigned tab_lane = (unsigned)(4 * q4) * 4;                   // table: the same 4 units (+ 64 B per unit group, + 4 UP16 B per kind)
    float vhalf = 0.5f;
    asm volatile("" : "+v"(vhalf));                                     // a register operand of v_fmac_f32_dpp (wave_half_sum)
    const uint32_t comp_xor = rc ? 3u : 0u;                             // complement [3, 2, 1, 0, 4] (model.py:233-237) = b ^ 3 for b < 4

    // ---- the wave's group of 16 windows: LDS carve, staged sequences, placement (what wg_setup does for a workgroup)
    unsigned char *const wbase_p = smem + 5 * XT_PITCH + (size_t)wave * wave_bytes;
    const unsigned wbase_off = (unsigned)(5 * XT_PITCH) + (unsigned)wave * (unsigned)wave_bytes;
    wave_ctx ctx;
    ctx.p = pin;
    const int64_t bid = wg_record_at<MODE>(pin, ctx.p, group);
    const gru_params &p = ctx.p;
    ctx.seqs = wbase_p + 4 * TILE_BYTES;
    ctx.row0s = reinterpret_cast<int64_t *>(ctx.seqs + gru_lds_seq(p.Tp));
    ctx.rowoff = reinterpret_cast<int *>(ctx.row0s + DGRP_WG_WINDOWS);
    ctx.obuf = reinterpret_cast<unsigned *>(ctx.rowoff + DGRP_WG_WINDOWS);
    ctx.wg_w = p.w0 + bid * DGRP_WG_WINDOWS;
    ctx.nvalid = (int)min((int64_t)DGRP_WG_WINDOWS, p.w0 + p.nw - ctx.wg_w);
    for (int i = lane; i < DGRP_WG_WINDOWS * T; i += 64) {
        const int wi = i / T, t = i - wi * T;
        ctx.seqs[wi * p.Tp + t] = wi < ctx.nvalid ? p.idx[(ctx.wg_w + wi) * p.s + t] : (uint8_t)4;
    }
    for (int i = lane; i < 4 * TILE_BYTES / 4; i += 64) reinterpret_cast<unsigned *>(wbase_p)[i] = 0u;      // h_{-1} = 0, hi and lo
    ctx.lo = 0;
    if (MODE == 0) {
        const int64_t a = dgrp_place_row(p.place, ctx.wg_w, p.s), b = dgrp_place_row(p.place, ctx.wg_w + ctx.nvalid - 1, p.s);
        ctx.lo = a < b ? a : b;
        for (int i = lane; i < p.ospan * C; i += 64) ctx.obuf[i] = 0u;
    }
    if (lane < DGRP_WG_WINDOWS) {
        int64_t r0 = -1;
        int off = -1;
        if (lane < ctx.nvalid) {
            r0 = MODE == 0 ? dgrp_place_row(p.place, ctx.wg_w + lane, p.s) : (ctx.wg_w + lane - p.w0 + p.avgw) * (int64_t)T;
            if (MODE == 0 && r0 >= ctx.lo && r0 - ctx.lo + T <= p.ospan) off = (int)(r0 - ctx.lo);
        }
        ctx.row0s[lane] = r0;
        ctx.rowoff[lane] = off;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                              // LDS operations of a wave complete in order: no s_barrier needed
    // modes 1, 2: rows of the output / spill buffers are numbered through the launch; element offsets relative to the group's first row
    float *const obase = MODE == 0 ? p.out : p.out + (ctx.wg_w - p.w0 + p.avgw) * (int64_t)T * C;
    float *const abase = MODE == 2 ? reinterpret_cast<float *>(p.avg) + (ctx.wg_w - p.w0 + p.avgw) * (int64_t)T * p.avg_up : nullptr;

    using tile = wave_tile<NU>;
    tile S0, S1;                          // two named objects, never indexed: they must stay in registers
    auto setup = [&](tile &Z, int x) {
        Z.hbuf = wbase_off + (unsigned)(2 * x) * TILE_BYTES;
        Z.lbuf = Z.hbuf + TILE_BYTES;
        Z.wbase = 8 * x;
        Z.seq_at = (unsigned)(ctx.seqs - smem) + (unsigned)(8 * x + win) * p.Tp + (rc ? T - 1 : 0);
#pragma unroll
        for (int i = 0; i < 4 * NU; ++i) Z.h[i] = ONERCP ? -1.0f : 0.0f;
        // logit chain q finishes window 8 x + q + 4 ((lane >> 4) & 1) + 2 (lane >> 5), class lane & 15 (see FS below)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int wi = 8 * x + q + 4 * ((lane >> 4) & 1) + 2 * (lane >> 5);
            int off = P_OFF;
            if (cls < C && wi < ctx.nvalid) {
                if (MODE == 0) off = ctx.rowoff[wi] >= 0 ? ctx.rowoff[wi] * C + cls : P_GLOBAL;
                else off = wi * T * C + cls;
            }
            Z.p_off[q] = off;
        }
        Z.avg_off = (MODE == 2 && !rc && 8 * x + win < ctx.nvalid) ? (8 * x + win) * T * p.avg_up + 4 * q4 : -1;
    };
    setup(S0, 0);
    setup(S1, 1);

    auto lds16 = [&](unsigned off) -> half8 { return *reinterpret_cast<const half8 *>(smem + off); };
    auto ldsf4 = [&](unsigned off) -> f32x4 { return *reinterpret_cast<const f32x4 *>(smem + off); };
    auto tab_row = [&](tile &Z, uint32_t b) {
        b = b < 4 ? b ^ comp_xor : 4;                             // (the step behind the last one reads a byte outside the window)
        Z.tab = tab_lane + b * XT_PITCH;
    };
    auto acc_init = [&](tile &Z, int g, int ug) {                 // kinds 0 r, 1 g (recurrent bias only), 2 z
        const f32x4 v = ldsf4(Z.tab + g * (UP16 * 4) + ug * 64);
        if (g == 0) Z.ar[ug] = v; else if (g == 1) Z.ag[ug] = v; else Z.az[ug] = v;
    };
    // one finished value of logit chain q of tile Z at step te
    auto emit = [&](tile &Z, int q, int te, float val) {
        const int off = Z.p_off[q];
        if (off == P_OFF) return;
        if (MODE == 0) {
            if (off >= 0) {
                lds_atomic_max(ctx.obuf + off + te * C, __float_as_uint(val));
            } else {
                const int wi = Z.wbase + q + 4 * ((lane >> 4) & 1) + 2 * (lane >> 5);
                const int64_t r = ctx.row0s[wi] + te;
                if (r < p.n) global_atomic_max(reinterpret_cast<unsigned *>(p.out) + r * C + cls, __float_as_uint(val));
            }
        } else {
            obase[off + te * C] = val;
        }
    };

    // ---- one phase: tile X's step tx on the matrix pipe, tile Y's epilogue of step ty in the gaps ----------------------------
    struct frag_ring { half8 h[2], l[2]; };                       // [k-step parity]
    struct fin_state { float x[2], lg[2], m[2], e[2], s[2]; };
    auto phase = [&](auto do_x, auto do_y, tile &X, tile &Y, int tx, int ty) __attribute__((always_inline)) {
        constexpr bool DO_X = decltype(do_x)::value, DO_Y = decltype(do_y)::value;
        frag_ring F;
        fin_state fs;
        split_gate_tmp gt[4 * NU];
        float pbh[NU][4];
        uint2 pbhv[NU], pblv[NU];
        f32x4 avv[NU];
        uint32_t xp_b = 4;
        const int te = ty - 1;                                    // step whose logits Y finishes in this phase (Dense of its last MFMA phase)
        if constexpr (DO_X) { F.h[0] = X.f0h; F.l[0] = X.f0l; }
#define GAP __builtin_amdgcn_sched_barrier(0);
        // recurrent MFMA n (0 .. 9 NU - 1) of k-step ks: pass n / (3 NU) (hi.h_hi, hi.h_lo, lo.h_hi), then gate, unit group
#define M_K(ks, n)                                                                                               \
    if constexpr (DO_X) {                                                                                        \
        constexpr int ps_ = (n) / (3 * NU), g_ = ((n) % (3 * NU)) / NU, ug_ = (n) % NU;                          \
        const half8 &b_ = ps_ == 1 ? F.l[(ks) & 1] : F.h[(ks) & 1];                                              \
        const u32x4 &w_ = ps_ == 2 ? W.lo[g_][ks][ug_] : W.hi[g_][ks][ug_];                                      \
        if constexpr (g_ == 0) WMFMA_R(X.ar[ug_], w_, b_);                                                       \
        else if constexpr (g_ == 1) WMFMA_R(X.ag[ug_], w_, b_);                                                  \
        else WMFMA_R(X.az[ug_], w_, b_);                                                                         \
    }
        // Dense of step tx - 1 from the same fragments (rows = tile rows): hi.hi, hi.lo(weights), lo(state).hi
#define M_D(ks, i)                                                                                               \
    if constexpr (DO_X) {                                                                                        \
        if constexpr ((ks) == 0 && (i) == 0) WMFMA_DZ(X.dpl, F.h[0], W.Bd_hi[0]);                                \
        else if constexpr ((i) == 0) WMFMA_D(X.dpl, F.h[(ks) & 1], W.Bd_hi[ks]);                                 \
        else if constexpr ((i) == 1) WMFMA_D(X.dpl, F.h[(ks) & 1], W.Bd_lo[ks]);                                 \
        else WMFMA_D(X.dpl, F.l[(ks) & 1], W.Bd_hi[ks]);                                                         \
    }
#define PF(ks)                                                                                                   \
    if constexpr (DO_X) { F.h[(ks) & 1] = lds16(X.hbuf + frag_lane + 64 * (ks)); F.l[(ks) & 1] = lds16(X.lbuf + frag_lane + 64 * (ks)); }
        // ---- Y's epilogue
#define AXL(ug) \
    if constexpr (DO_Y) Y.ax[ug] = ldsf4(Y.tab + 3 * (UP16 * 4) + (ug) * 64);
#define G(e, op) \
    if constexpr (DO_Y) split_gate_op<ONERCP, op>(gt[e], Y.ar[(e) / 4][(e) % 4], Y.ag[(e) / 4][(e) % 4], Y.az[(e) / 4][(e) % 4], Y.ax[(e) / 4][(e) % 4], Y.h[e]);
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
        } else if constexpr ((g) % 2 == 1) {     /* a pair of unit groups: 16 bytes per lane and half (column order: dgrp_wave_col) */ \
            *reinterpret_cast<uint4 *>(smem + Y.hbuf + frag_lane + ((g) / 2) * 64) = make_uint4(pbhv[(g) - 1].x, pbhv[(g) - 1].y, pbhv[g].x, pbhv[g].y); \
            *reinterpret_cast<uint4 *>(smem + Y.lbuf + frag_lane + ((g) / 2) * 64) = make_uint4(pblv[(g) - 1].x, pblv[(g) - 1].y, pblv[g].x, pblv[g].y); \
        } else if constexpr ((g) == NU - 1) {    /* the last group of an odd count on its own */                  \
            *reinterpret_cast<uint2 *>(smem + Y.hbuf + frag_lane + ((g) / 2) * 64) = pbhv[g];                    \
            *reinterpret_cast<uint2 *>(smem + Y.lbuf + frag_lane + ((g) / 2) * 64) = pblv[g];                    \
        }                                                                                                        \
    }
        // attention pre-pass: avg[ty] of unit group g = (h_fwd + h_rc) / 2 from the fp32 state (pbh[g] holds h between PB 0 and PB 2);
        // the other strand of the lane's window is 8 lanes away in the same row of 16
#define AV(g, op)                                                                                                \
    if constexpr (DO_Y && MODE == 2) {                                                                           \
        if constexpr ((op) == 0) {                                                                               \
            avv[g] = f32x4{ wave_half_sum(pbh[g][0], vhalf), wave_half_sum(pbh[g][1], vhalf), wave_half_sum(pbh[g][2], vhalf), wave_half_sum(pbh[g][3], vhalf) }; \
        } else if (Y.avg_off >= 0) {                                                                             \
            *reinterpret_cast<f32x4 *>(abase + Y.avg_off + ty * p.avg_up + 16 * (g)) = avv[g];                   \
            if constexpr ((g) == NU - 1 && (NU & 1)) {                                                           \
                if (p.avg_up > UP16) *reinterpret_cast<f32x4 *>(abase + Y.avg_off + ty * p.avg_up + UP16) = f32x4{ 0, 0, 0, 0 }; \
            }                                                                                                    \
        }                                                                                                        \
    }
#define RD0 \
    if constexpr (DO_Y) { Y.f0h = lds16(Y.hbuf + frag_lane); Y.f0l = lds16(Y.lbuf + frag_lane); }
#define XP(op)                                                                                                   \
    if constexpr (DO_Y) {                                                                                        \
        if constexpr ((op) == 0) { Y.seq_at += rc ? -1 : 1; xp_b = smem[Y.seq_at]; }                             \
        else tab_row(Y, xp_b);                                                                                   \
    }
#define CI(g, ug) \
    if constexpr (DO_Y) acc_init(Y, g, ug);
        // The Dense tile holds rows 0-7 (forward strands of windows 0-7) in lanes 0-31 and rows 8-15 (their reverse complements)
        // in lanes 32-63: v_permlane32_swap of registers (0, 2) and (1, 3) + one add each leaves the full logits of windows
        // {0, 4, 2, 6}[lane >> 4] in x[0] and of {1, 5, 3, 7}[lane >> 4] in x[1] -- both half-waves busy on different windows
#define FS                                                                                                       \
    if constexpr (DO_Y) {                                                                                        \
        asm volatile("" : "+v"(Y.dpl));       /* a result the compiler can prove unused would free its registers under the MFMA */ \
        const auto s02_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(Y.dpl[0]), __float_as_uint(Y.dpl[2]), false, false); \
        const auto s13_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(Y.dpl[1]), __float_as_uint(Y.dpl[3]), false, false); \
        fs.x[0] = __uint_as_float(s02_[0]) + __uint_as_float(s02_[1]);                                           \
        fs.x[1] = __uint_as_float(s13_[0]) + __uint_as_float(s13_[1]);                                           \
    }
#define FN(q, op)                                                                                                \
    if constexpr (DO_Y) {                                                                                        \
        if constexpr ((op) == 1) fs.lg[q] = cls < C ? fs.x[q] + fbias : -INFINITY;                               \
        else if constexpr ((op) == 2) { if (MODE != 2) fs.m[q] = row_max_ror<8>(fs.lg[q]); }                     \
        else if constexpr ((op) == 3) { if (MODE != 2) fs.m[q] = row_max_ror<4>(fs.m[q]); }                      \
        else if constexpr ((op) == 4) { if (MODE != 2) fs.m[q] = row_max_ror<2>(fs.m[q]); }                      \
        else if constexpr ((op) == 5) { if (MODE != 2) fs.m[q] = row_max_ror<1>(fs.m[q]); }                      \
        else if constexpr ((op) == 6) { if (MODE != 2) fs.e[q] = __builtin_amdgcn_exp2f(1.4426950408889634f * (fs.lg[q] - fs.m[q])); } \
        else if constexpr ((op) == 7) { if (MODE != 2) fs.s[q] = fs.e[q] + row_ror<8>(fs.e[q]); }                \
        else if constexpr ((op) == 8) { if (MODE != 2) fs.s[q] += row_ror<4>(fs.s[q]); }                         \
        else if constexpr ((op) == 9) { if (MODE != 2) fs.s[q] += row_ror<2>(fs.s[q]); }                         \
        else if constexpr ((op) == 10) { if (MODE != 2) fs.s[q] += row_ror<1>(fs.s[q]); }                        \
        else if constexpr ((op) == 11) { if (MODE != 2) fs.e[q] *= __builtin_amdgcn_rcpf(fs.s[q]); }             \
        else { if (te >= 0) emit(Y, q, te, MODE == 2 ? fs.lg[q] : fs.e[q]); }                                    \
    }
        if constexpr (NU == 1) {
#include "gru_wave_phase_nu1.inc"
        } else if constexpr (NU == 2) {
#include "gru_wave_phase_nu2.inc"
        } else if constexpr (NU == 3) {
#include "gru_wave_phase_nu3.inc"
        } else {
#include "gru_wave_phase_nu4.inc"
        }
#undef GAP
#undef M_K
#undef M_D
#undef PF
#undef AXL
#undef G
#undef PB
#undef AV
#undef RD0
#undef XP
#undef CI
#undef FS
#undef FN
    };
    const std::true_type yes;
    const std::false_type no;

    // prologue: the operands of both tiles' step 0 (h_{-1} = 0 is in LDS), then tile 0's step 0 with nothing beside it
    auto first_step = [&](tile &Z) {
        tab_row(Z, smem[Z.seq_at]);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ug = 0; ug < NU; ++ug) acc_init(Z, g, ug);
        Z.f0h = lds16(Z.hbuf + frag_lane);
        Z.f0l = lds16(Z.lbuf + frag_lane);
    };
    first_step(S0);
    first_step(S1);
    // Register copies the allocator places on a control-flow edge (loop entry, back edge, exit) are VALU accesses it does not know to
    // keep away from the asm MFMAs in front of them: the last MFMAs of a phase have to be complete before the edge.
#define EDGE_PAD do { asm volatile("s_nop 15\n\ts_nop 7"); __builtin_amdgcn_sched_barrier(0); } while (0)
    phase(yes, no, S0, S1, 0, 0);
    EDGE_PAD;
    for (int t = 0; t + 1 < T; ++t) {
        phase(yes, yes, S1, S0, t, t);          // tile 1's step t      ||  tile 0 finishes step t
        phase(yes, yes, S0, S1, t + 1, t);      // tile 0's step t + 1  ||  tile 1 finishes step t
        EDGE_PAD;
    }
    phase(yes, yes, S1, S0, T - 1, T - 1);
    EDGE_PAD;
    phase(no, yes, S0, S1, T, T - 1);

    // drain: Dense and softmax / merge of the last step, image flush
    auto drain = [&](tile &Z) {
        f32x4 d;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const half8 fh = lds16(Z.hbuf + frag_lane + 64 * ks), fl = lds16(Z.lbuf + frag_lane + 64 * ks);
            if (ks == 0) WMFMA_DZ(d, fh, W.Bd_hi[0]); else WMFMA_D(d, fh, W.Bd_hi[ks]);
            WMFMA_D(d, fh, W.Bd_lo[ks]);
            WMFMA_D(d, fl, W.Bd_hi[ks]);
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(d));           // MFMA result -> VALU read, no compiler padding behind asm
        const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[0]), __float_as_uint(d[2]), false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[1]), __float_as_uint(d[3]), false, false);
        const float x[2] = { __uint_as_float(s02[0]) + __uint_as_float(s02[1]), __uint_as_float(s13[0]) + __uint_as_float(s13[1]) };
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float lg = cls < C ? x[q] + fbias : -INFINITY;
            float val = lg;
            if (MODE != 2) {
                const float m = row_max_ror<1>(row_max_ror<2>(row_max_ror<4>(row_max_ror<8>(lg))));
                const float e = __builtin_amdgcn_exp2f(1.4426950408889634f * (lg - m));
                float s = e + row_ror<8>(e);
                s += row_ror<4>(s); s += row_ror<2>(s); s += row_ror<1>(s);
                val = e * __builtin_amdgcn_rcpf(s);
            }
            emit(Z, q, T - 1, val);
        }
    };
    // (tile 0's logits of step T - 2 were finished in its last epilogue phase, tile 1's in the phase above)
    drain(S0);
    drain(S1);
    if (MODE == 0 && p.ospan > 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        unsigned *gout = reinterpret_cast<unsigned *>(p.out) + ctx.lo * C;
        const int64_t lim = (p.n - ctx.lo) * C;
        for (int i = lane; i < p.ospan * C; i += 64) {
            const unsigned v = ctx.obuf[i];
            if (v != 0u && i < lim) global_atomic_max(gout + i, v);
        }
    }
}

template <int NU, int MODE, bool ONERCP>
int launch_wave_one(const gru_params &p, int64_t groups, int wave_bytes, hipStream_t stream)
{
    static std::once_flag configured;
    static hipError_t cfg_err = hipSuccess;
    std::call_once(configured, [] {
        cfg_err = hipFuncSetAttribute((const void *)gru_wave_kernel<NU, MODE, ONERCP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    DGRP_HIP(cfg_err);
    const size_t lds = (size_t)5 * wave_cfg<NU>::XT_PITCH + (size_t)4 * wave_bytes;
    hipLaunchKernelGGL((gru_wave_kernel<NU, MODE, ONERCP>), dim3((unsigned)((groups + 3) / 4)), dim3(256), lds, stream, p, groups, wave_bytes);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

template <int NU>
int launch_wave(const gru_params &p, int64_t groups, int wave_bytes, bool onercp, hipStream_t stream)
{
    if (onercp)
        return p.mode == 0 ? launch_wave_one<NU, 0, true>(p, groups, wave_bytes, stream)
             : p.mode == 1 ? launch_wave_one<NU, 1, true>(p, groups, wave_bytes, stream) : launch_wave_one<NU, 2, true>(p, groups, wave_bytes, stream);
    return p.mode == 0 ? launch_wave_one<NU, 0, false>(p, groups, wave_bytes, stream)
         : p.mode == 1 ? launch_wave_one<NU, 1, false>(p, groups, wave_bytes, stream) : launch_wave_one<NU, 2, false>(p, groups, wave_bytes, stream);
}

}  // namespace

// LDS of one wave's group of 16 windows: two row tiles (hi + lo halves), the staged sequences, placement, and (mode 0) as many rows
// of the merged-output image as `budget` allows.  Sets p.ospan; returns the bytes (a multiple of 16).
int dgrp_wave_carve(int NU, gru_params &p, int mode, int64_t s, int64_t budget)
{
    const int KS = (NU + 1) / 2, HS = 32 * KS + WPAD;
    const int fixed = 4 * 16 * HS * 2 + gru_lds_seq(p.Tp) + gru_lds_meta();
    p.ospan = 0;
    if (mode == 0) {
        const int64_t want = (DGRP_WG_WINDOWS - 1) * s + p.T;
        const int64_t cap = (budget - fixed) / (p.C * 4);
        p.ospan = (int)(want < cap ? want : cap);
        if (p.ospan < p.T) p.ospan = 0;
    }
    return (int)dgrp_align_up(fixed + (int64_t)p.ospan * p.C * 4, 16);
}

int dgrp_wave_table_bytes(int NU) { return 5 * (4 * 16 * NU * 4 + 32); }

int dgrp_wave_launch(const gru_params &p, int NU, int64_t groups, int wave_bytes, bool onercp, hipStream_t stream)
{
    switch (NU) {
    case 1: return launch_wave<1>(p, groups, wave_bytes, onercp, stream);
    case 2: return launch_wave<2>(p, groups, wave_bytes, onercp, stream);
    case 3: return launch_wave<3>(p, groups, wave_bytes, onercp, stream);
    case 4: return launch_wave<4>(p, groups, wave_bytes, onercp, stream);
    default:
        dgrp_set_error("dgrp_wave_launch: %d unit groups (1..4)", NU);
        return DGRP_EINVAL;
    }
}

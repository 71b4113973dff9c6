// Pieces shared by the recurrent kernels (gru_kernel.hip, gru_split2.hip): launch parameters, LDS carve, workgroup
// set-up, softmax/merge of one logit register, image flush.
#pragma once
#include "dgrp_model.h"
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#define DGRP_WG_WINDOWS 16

struct gru_rec {
    int64_t idx_off, n, out_row, nwin;
    dgrp_placement place;
    int64_t win_first;   // windows of the records before this one (row of its first window in the avg / pl spill)
    int64_t pad_;
};

struct gru_params {
    const uint8_t *idx;   // class index per base [n]
    int64_t n, s, w0, nw; // this launch covers windows w0 .. w0+nw-1 (absolute indices)
    dgrp_placement place;
    const uint4 *pack;
    const float *ffb;
    float *out;           // mode 0: merged [n, C]; mode 1: probs [nw, T, C]; mode 2: logits part [nw, T, C]
    void *avg;            // mode 2: [nw, T, UP], fp16 (fp16-operand kernels) or fp32 (split-operand kernels: avg_f32)
    int avg_f32;
    int T, C, nfrag, mode;
    int Tp;               // T rounded up to 16 (row pitch of the staged sequences)
    int ospan;            // rows of the LDS output image (mode 0), 0 = none
    uint64_t *stamps;
    // batched records (mode 0): workgroup b belongs to record r with wg_first[r] <= b < wg_first[r+1]; idx / out / n /
    // placement then come from recs[r] and windows count from 0 inside the record
    const struct gru_rec *recs;
    const int64_t *wg_first;
    int64_t nrec;
    int64_t avgw;         // modes 1, 2: row of window w0 in the output / spill buffers (batched records: the record's first)
    // split-operand kernel only: the lo halves of the recurrent fragments ([NW][KS][3][64]: k-step major, gates r, g, z), the byte
    // offset of the lo hidden tiles in the dynamic LDS, and 1.0 if the packed z bias carries the one-reciprocal "+1"
    const uint4 *pack_lo;
    int lo_tile_off;
    float zfold;
    // gru_split2_kernel only: 16x16x32 A fragments and the input-projection table (dgrp_model.h), byte offset of the table's LDS copy
    const uint4 *pack16;
    const float *xtab;
    int xtab_off;
    // rnn_split_stream_kernel only (rnn_stream.hip): hi and lo recurrent fragments in consumption order, [NW][KS][2 G][64]
    const uint4 *stream;
    // gru_wave_kernel only (gru_wave.hip, GRU up to 64 units): all-unit 16x16x32 A fragments + Dense B fragments, the input-projection
    // table [5][4][16 NU], and the row length (floats) of the avg[t] spill (the model's UP)
    const uint4 *packw;
    const float *xtabw;
    int avg_up;
};

// The packed gate weights carry the exp2 scale (-log2 e for z and r, 2 log2 e for the candidate),
// so the accumulators feed v_exp_f32 directly:  sigmoid(x) = 1/(1 + 2^(-x log2 e)),
// tanh(x) = 1 - 2/(1 + 2^(2 x log2 e)).
__device__ __forceinline__ float sigmoid_from_scaled(float a) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a)); }
__device__ __forceinline__ float tanh_from_scaled(float a) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a)); }
// Two gate values at a time: the transcendentals are scalar instructions, the "1 +" between them is one
// packed add (v_pk_add_f32 does two lanes' worth per issue slot).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 rcp1p_exp2_pair(float a0, float a1)
{
    f32x2 e = { __builtin_amdgcn_exp2f(a0), __builtin_amdgcn_exp2f(a1) };
    e = e + 1.0f;
    return f32x2{ __builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y) };
}
__device__ __forceinline__ float fast_tanh(float x) { return tanh_from_scaled(2.8853900817779268f * x); }

// rotate within each row of 16 lanes (DPP row_ror): an all-reduce over the 16 class lanes in 4 steps
template <int N>
__device__ __forceinline__ float row_ror(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 | N, 0xf, 0xf, false));
}
// max(x, x rotated by N) as ONE v_max_f32_dpp (fmaxf on a DPP move costs a zero fill, the move and two
// canonicalising maxes); the s_nop covers the VALU-write -> DPP-read hazard the assembler does not see.
template <int N>
__device__ __forceinline__ float row_max_ror(float x)
{
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(N));
    return r;
}
// exchange inside quads: 0x4E = lanes [2,3,0,1] (xor 2), 0xB1 = [1,0,3,2] (xor 1)
template <int CTRL>
__device__ __forceinline__ float quad_perm(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_allmax(float x)
{
    return row_max_ror<1>(row_max_ror<2>(row_max_ror<4>(row_max_ror<8>(x))));
}
__device__ __forceinline__ float row_allsum(float x)
{
    x += row_ror<8>(x); x += row_ror<4>(x); x += row_ror<2>(x); return x + row_ror<1>(x);
}

// The two destinations of the max-merge as distinct instructions (ds_max_u32 / global_atomic_umax): left to
// atomicMax on generic pointers the compiler merges both branches into one flat atomic, whose latency then
// sits in front of every later LDS wait of the wave.
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(1))) unsigned glb_u32;
__device__ __forceinline__ void lds_atomic_max(unsigned *p, unsigned v)
{
    (void)__hip_atomic_fetch_max((lds_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void global_atomic_max(unsigned *p, unsigned v)
{
    (void)__hip_atomic_fetch_max((glb_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS carve (bytes) -- shared by host and device
__host__ __device__ static inline int gru_lds_hbuf(int UP, int pad = 8) { return 2 * 32 * (UP + pad) * 2; }   // pad: row pitch UP + pad halves
__host__ __device__ static inline int gru_lds_dpart(int NW) { return 2 * NW * 64 * 16; }
__host__ __device__ static inline int gru_lds_seq(int Tp) { return DGRP_WG_WINDOWS * Tp; }
__host__ __device__ static inline int gru_lds_meta() { return DGRP_WG_WINDOWS * 8 + DGRP_WG_WINDOWS * 4; }

// ---- pieces shared by the GRU and the LSTM kernel --------------------------------------------------------------
struct wg_ctx {                       // LDS carve of a workgroup and what its 16 windows are
    _Float16 *hbuf;                   // [2][32][HS] hidden tile, ping-pong
    float *dpart;                     // [2][4 regs][NW][64] partial logits of the waves
    uint8_t *seqs;                    // [16][Tp] class indices of the windows
    int64_t *row0s;                   // first output row of each window (mode 0: merged row, else row in [nw*T])
    int *rowoff;                      // mode 0: row in the LDS image, -1 = goes to HBM directly
    unsigned *obuf;                   // mode 0: max image of the rows the windows cover
    int64_t wg_w, lo;                 // first window; first row of the image
    int nvalid;
};

// batched records: rewrite the launch-wide parameters into those of the record this workgroup belongs to (last r with
// wg_first[r] <= blockIdx.x; uniform over the workgroup: scalar loads) and return the workgroup's index inside it
template <int MODE>
__device__ __forceinline__ int64_t wg_record_at(const gru_params &pin, gru_params &p, int64_t g)
{
    if (MODE == 1 || !pin.recs) return g;
    int64_t lo_r = 0, hi_r = pin.nrec;
    while (hi_r - lo_r > 1) {
        const int64_t mid = (lo_r + hi_r) >> 1;
        if (pin.wg_first[mid] <= g) lo_r = mid; else hi_r = mid;
    }
    const gru_rec rc = pin.recs[lo_r];
    p.idx = pin.idx + rc.idx_off;
    p.n = rc.n;
    if (MODE == 0) p.out = pin.out + rc.out_row * pin.C;
    p.place = rc.place;
    p.w0 = 0;
    p.nw = rc.nwin;
    p.avgw = rc.win_first;
    return g - pin.wg_first[lo_r];
}
template <int MODE>
__device__ __forceinline__ int64_t wg_record(const gru_params &pin, gru_params &p) { return wg_record_at<MODE>(pin, p, blockIdx.x); }

// carve, stage the windows' sequences, zero the state and the image, work out the placement (ends with a barrier)
template <int NW, int MODE, int HPAD = 8>
__device__ __forceinline__ wg_ctx wg_setup(const gru_params &p, unsigned char *smem, int64_t bid)
{
    constexpr int UP = 32 * NW, HS = UP + HPAD;
    wg_ctx c;
    c.hbuf = reinterpret_cast<_Float16 *>(smem);
    c.dpart = reinterpret_cast<float *>(smem + gru_lds_hbuf(UP, HPAD));
    c.seqs = smem + gru_lds_hbuf(UP, HPAD) + gru_lds_dpart(NW);
    c.row0s = reinterpret_cast<int64_t *>(c.seqs + gru_lds_seq(p.Tp));
    c.rowoff = reinterpret_cast<int *>(c.row0s + DGRP_WG_WINDOWS);
    c.obuf = reinterpret_cast<unsigned *>(c.rowoff + DGRP_WG_WINDOWS);
    const int tid = threadIdx.x, T = p.T, C = p.C;
    c.wg_w = p.w0 + bid * DGRP_WG_WINDOWS;
    c.nvalid = (int)min((int64_t)DGRP_WG_WINDOWS, p.w0 + p.nw - c.wg_w);
    for (int i = tid; i < DGRP_WG_WINDOWS * T; i += 64 * NW) {
        const int wi = i / T, t = i - wi * T;
        c.seqs[wi * p.Tp + t] = wi < c.nvalid ? p.idx[(c.wg_w + wi) * p.s + t] : (uint8_t)4;
    }
    for (int i = tid; i < 32 * HS; i += 64 * NW) c.hbuf[i] = (_Float16)0.0f;          // h_{-1} = 0
    c.lo = 0;
    if (MODE == 0) {
        // smallest placement row of the two ends (the partial-batch shift keeps rows monotone inside
        // each regime); windows that fall outside [lo, lo + ospan) go to HBM directly
        const int64_t a = dgrp_place_row(p.place, c.wg_w, p.s), b = dgrp_place_row(p.place, c.wg_w + c.nvalid - 1, p.s);
        c.lo = a < b ? a : b;
        for (int i = tid; i < p.ospan * C; i += 64 * NW) c.obuf[i] = 0u;
    }
    if (tid < DGRP_WG_WINDOWS) {
        int64_t r0 = -1;
        int off = -1;
        if (tid < c.nvalid) {
            r0 = MODE == 0 ? dgrp_place_row(p.place, c.wg_w + tid, p.s) : (c.wg_w + tid - p.w0 + p.avgw) * (int64_t)T;
            if (MODE == 0 && r0 >= c.lo && r0 - c.lo + T <= p.ospan) off = (int)(r0 - c.lo);
        }
        c.row0s[tid] = r0;
        c.rowoff[tid] = off;
    }
    __syncthreads();
    return c;
}

// merge (mode 0) or store one finished value of window `wi`, step t, class `cls`
template <int MODE>
__device__ __forceinline__ void emit_value(const gru_params &p, const wg_ctx &c, int off, int64_t row0, int t, int cls, float val)
{
    if (MODE == 0) {
        if (off >= 0) {
            lds_atomic_max(c.obuf + (off + t) * p.C + cls, __float_as_uint(val));
        } else {
            const int64_t row = row0 + t;
            if (row < p.n) global_atomic_max(reinterpret_cast<unsigned *>(p.out) + row * p.C + cls, __float_as_uint(val));
        }
    } else {
        p.out[(row0 + t) * p.C + cls] = val;
    }
}

// Softmax + merge of step t's partial logits for accumulator register `reg`: the 16x16 logit tile (window =
// 4*(lane>>4) + reg, class = lane & 15) is split by register over the waves, one value per lane.
template <int NW, int MODE>
__device__ __forceinline__ void finish_register(const gru_params &p, const wg_ctx &c, int t, int reg, float fbias, int off, int64_t row0)
{
    const int lane = threadIdx.x & 63, cls = lane & 15;
    const float *dp = c.dpart + ((size_t)(t & 1) * 4 + reg) * NW * 64 + lane;
    float sum = dp[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) sum += dp[w * 64];
    const int wi = 4 * (lane >> 4) + reg;
    const float lg = cls < p.C ? sum + fbias : -INFINITY;
    float val = lg;
    if (MODE != 2) {                                  // attention: softmax happens in the second kernel
        const float m = row_allmax(lg);
        const float e = __builtin_amdgcn_exp2f(1.4426950408889634f * (lg - m));   // 0 for the padding lanes
        val = e * __builtin_amdgcn_rcpf(row_allsum(e));
    }
    if (cls < p.C && wi < c.nvalid) emit_value<MODE>(p, c, off, row0, t, cls, val);
}

// flush the pre-merged image: contiguous rows -> 256-byte atomic wave-instructions
template <int NW>
__device__ __forceinline__ void flush_image(const gru_params &p, const wg_ctx &c)
{
    __syncthreads();
    unsigned *gout = reinterpret_cast<unsigned *>(p.out) + c.lo * p.C;
    const int64_t lim = (p.n - c.lo) * p.C;
    for (int i = threadIdx.x; i < p.ospan * p.C; i += 64 * NW) {
        const unsigned v = c.obuf[i];
        if (v != 0u && i < lim) global_atomic_max(gout + i, v);
    }
}

#ifdef DGRP_STAMP
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const uint64_t now_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += (uint32_t)(now_ - stamp_prev); stamp_prev = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
#ifndef DGRP_PIPE
#define DGRP_PIPE 1
#endif

// attention pre-pass of the split-operand kernels: avg[t] of this wave's 32 units for window (lane & 15), summed from the
// hi and lo halves of both strands (the second kernel's operand)
__device__ __forceinline__ void split_avg_store_lane(const gru_params &p, int64_t wg_w, int tt, int UP, int wave, int lane, half8 a0, half8 a1,
                                                     half8 l0, half8 l1)
{
    const int64_t at = ((wg_w + (lane & 15) - p.w0 + p.avgw) * (int64_t)p.T + tt) * UP + 32 * wave + 8 * (lane >> 4);
    if (p.avg_f32) {
        // fp32 spill: avg = ((f_hi + f_lo) + (r_hi + r_lo)) / 2 in float, so that the attention kernel sees what the recurrence computed
        // to fp32 rounding (an fp16 spill costs the second kernel 1e-4 in the class probabilities)
        // Per value: the strand's half sum (hi + lo) / 2 as ONE v_fma_mix_f32 -- fma(hi, 0.5, lo / 2) with both halves widened by the
        // instruction itself, exact: hi + lo has at most 23 significant bits and lo / 2 is a packed multiply by a power of two (inexact only
        // for a lo at the bottom of fp16's subnormal range: 3e-8 absolute) -- then one add of
        // the two strands (the only rounding, the same value as rounding the whole sum and halving it).  32 vector instructions per
        // step instead of the 64 of converting the four fp16 tiles to float first.
        float *dst = reinterpret_cast<float *>(p.avg) + at;
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        const _Float16 hf = (_Float16)0.5f;
        const u32x4_ A0 = __builtin_bit_cast(u32x4_, a0), A1 = __builtin_bit_cast(u32x4_, a1);
        const u32x4_ L0 = __builtin_bit_cast(u32x4_, l0 * hf), L1 = __builtin_bit_cast(u32x4_, l1 * hf);
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f0, f1, r0, r1;
            asm("v_fma_mix_f32 %0, %1, 0.5, %2 op_sel_hi:[1,0,1]" : "=v"(f0) : "v"(A0[k]), "v"(L0[k]));
            asm("v_fma_mix_f32 %0, %1, 0.5, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(f1) : "v"(A0[k]), "v"(L0[k]));
            asm("v_fma_mix_f32 %0, %1, 0.5, %2 op_sel_hi:[1,0,1]" : "=v"(r0) : "v"(A1[k]), "v"(L1[k]));
            asm("v_fma_mix_f32 %0, %1, 0.5, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r1) : "v"(A1[k]), "v"(L1[k]));
            v[2 * k] = f0 + r0;
            v[2 * k + 1] = f1 + r1;
        }
        *reinterpret_cast<f32x4 *>(dst) = f32x4{ v[0], v[1], v[2], v[3] };
        *reinterpret_cast<f32x4 *>(dst + 4) = f32x4{ v[4], v[5], v[6], v[7] };
        return;
    }
    // packed fp16 arithmetic (the halvings are exact; one rounding per add): a quarter of the instructions of a float detour
    const _Float16 hf = (_Float16)0.5f;
    const half8 av = (a0 * hf + a1 * hf) + (l0 + l1) * hf;
    *reinterpret_cast<half8 *>(reinterpret_cast<_Float16 *>(p.avg) + at) = av;
}
__device__ __forceinline__ void split_avg_store(const gru_params &p, int64_t wg_w, int tt, int UP, int wave, half8 a0, half8 a1,
                                                half8 l0, half8 l1)
{
    split_avg_store_lane(p, wg_w, tt, UP, wave, threadIdx.x & 63, a0, a1, l0, l1);
}

// Gate math of ONE (row, unit) of the split-operand kernels as a chain of single operations, written once so that the
// one-tile kernel (which runs the chain in one go) and the two-tile kernel (which drops the links one by one into the gaps
// between another tile's MFMAs) round identically: a record must not change its calls with the way it was batched.
// Accumulators arrive in the exp2 domain (scales folded into the packed weights).  Every a*b+c is an explicit fma, every
// other operation a lone add or multiply, so the compiler has nothing to contract differently in the two kernels.
//   ONERCP (the model constructor proved (1 + 2^az)(1 + 2^ag) finite; the packed z bias carries +1, so 2^az = 2 Ez):
//     state s = h - 1;   s' = [ s (1 + Eg) - 2 Ez ] / [ (1 + Eg)(1 + Ez) ]          -- 5 transcendentals
//   otherwise: state s = h;  s' = hh + z (s - hh),  hh = 1 - 2 / (1 + Eg),  z = 1 / (1 + Ez)   -- 6 transcendentals
struct split_gate_tmp { float er, e2, g, A, zt, d, n; };
template <bool ONERCP, int OP>
__device__ __forceinline__ void split_gate_op(split_gate_tmp &t, float ar, float ag, float az, float ax, float &s)
{
#pragma clang fp contract(off)
    if constexpr (OP == 0) t.er = __builtin_amdgcn_exp2f(ar);
    else if constexpr (OP == 1) t.e2 = __builtin_amdgcn_exp2f(az);
    else if constexpr (OP == 2) t.er = t.er + 1.0f;
    else if constexpr (OP == 3) t.er = __builtin_amdgcn_rcpf(t.er);                         // r
    else if constexpr (OP == 4) t.g = __builtin_fmaf(t.er, ag, ax);                         // x.W_h + b_in_h + r * (h.U_h + b_rec_h)
    else if constexpr (OP == 5) t.A = __builtin_amdgcn_exp2f(t.g);
    else if constexpr (OP == 6) t.A = t.A + 1.0f;                                           // 1 + Eg
    else if constexpr (ONERCP) {
        if constexpr (OP == 7) t.zt = __builtin_fmaf(0.5f, t.e2, 1.0f);                     // 1 + Ez
        else if constexpr (OP == 8) t.d = t.A * t.zt;
        else if constexpr (OP == 9) t.n = __builtin_fmaf(s, t.A, -t.e2);
        else if constexpr (OP == 10) t.d = __builtin_amdgcn_rcpf(t.d);
        else s = t.n * t.d;
    } else {
        if constexpr (OP == 7) t.zt = t.e2 + 1.0f;
        else if constexpr (OP == 8) t.d = __builtin_amdgcn_rcpf(t.A);
        else if constexpr (OP == 9) t.n = __builtin_fmaf(-2.0f, t.d, 1.0f);                 // hh = tanh
        else if constexpr (OP == 10) t.zt = __builtin_amdgcn_rcpf(t.zt);                    // z
        else { const float df = s - t.n; s = __builtin_fmaf(t.zt, df, t.n); }               // z*h + (1-z)*hh
    }
}
#define DGRP_SPLIT_GATE_OPS 12
template <bool ONERCP>
__device__ __forceinline__ float split_gate_chain(float ar, float ag, float az, float ax, float s)
{
    split_gate_tmp t;
    split_gate_op<ONERCP, 0>(t, ar, ag, az, ax, s); split_gate_op<ONERCP, 1>(t, ar, ag, az, ax, s);
    split_gate_op<ONERCP, 2>(t, ar, ag, az, ax, s); split_gate_op<ONERCP, 3>(t, ar, ag, az, ax, s);
    split_gate_op<ONERCP, 4>(t, ar, ag, az, ax, s); split_gate_op<ONERCP, 5>(t, ar, ag, az, ax, s);
    split_gate_op<ONERCP, 6>(t, ar, ag, az, ax, s); split_gate_op<ONERCP, 7>(t, ar, ag, az, ax, s);
    split_gate_op<ONERCP, 8>(t, ar, ag, az, ax, s); split_gate_op<ONERCP, 9>(t, ar, ag, az, ax, s);
    split_gate_op<ONERCP, 10>(t, ar, ag, az, ax, s); split_gate_op<ONERCP, 11>(t, ar, ag, az, ax, s);
    return s;
}
// the hidden state behind the gate state, and its fp16 pair: hi = fp16(h), lo = fp16(h - hi)
template <bool ONERCP> __device__ __forceinline__ float split_state_h(float s)
{
#pragma clang fp contract(off)
    return ONERCP ? s + 1.0f : s;
}
// Four state values as an fp16 pair, link by link (the two-tile kernel drops the links into MFMA gaps): two packed converts,
// the residuals h - hi as ONE v_fma_mix_f32 each (it widens the fp16 half it is pointed at: fma(hi, -1, h), exact), two
// packed converts.  Plain asm (not volatile): single instructions the scheduler may move like any other.
__device__ __forceinline__ uint2 split_pack4(const float h[4])
{
    uint2 r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.x) : "v"(h[0]), "v"(h[1]));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.y) : "v"(h[2]), "v"(h[3]));
    return r;
}
__device__ __forceinline__ void split_residual4(float h[4], uint2 hi)
{
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(h[0]) : "v"(hi.x));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(h[1]) : "v"(hi.x));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(h[2]) : "v"(hi.y));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(h[3]) : "v"(hi.y));
}
__device__ __forceinline__ void split_hi_lo4(const float h[4], uint2 &hv, uint2 &lv)
{
    float r[4] = { h[0], h[1], h[2], h[3] };
    hv = split_pack4(r);
    split_residual4(r, hv);
    lv = split_pack4(r);
}

// attention pre-pass of gru_wave_kernel: (h + h of the other strand) / 2, the other strand's row sitting 8 lanes away in the same DPP
// row of 16 -- one rounding (halving is exact), the value of rounding the sum and halving it.  Two instructions: h / 2, then
// v_fmac_f32_dpp adds the rotated h times 1/2 (through the builtin the compiler spent two v_mov per value on the rotation; the
// s_nop covers the VALU-write -> DPP-read hazard the assembler does not see).  `half` = a register holding 0.5.
__device__ __forceinline__ float wave_half_sum(float h, float half)
{
#pragma clang fp contract(off)
    float t = half * h;
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(h), "v"(half));
    return t;
}

// LSTM cell update of one (row, unit) for the split-operand kernel, every a*b+c an explicit fma and nothing left to contract, so
// that the MODE instantiations of a kernel (merged output / window probabilities) round identically.  Accumulators arrive in the
// exp2 domain: sigmoid(x) = 1/(1 + 2^a), tanh(x) = 1 - 2/(1 + 2^a).   c' = f c + i tanh(z_c);  h' = o tanh(c')
__device__ __forceinline__ void split_lstm_cell(float ai, float af, float ac, float ao, float &c, float &h)
{
#pragma clang fp contract(off)
    const float ig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ai));
    const float fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(af));
    const float tc = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ac)), 1.0f);
    const float it = ig * tc;
    c = __builtin_fmaf(fg, c, it);
    const float og = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ao));
    const float e2 = __builtin_amdgcn_exp2f(2.8853900817779268f * c);
    const float th = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e2), 1.0f);
    h = og * th;
}

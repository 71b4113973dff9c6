// Maximal scoring segments with x-drop (A9/A10: deepgrp/_mss/pymss.pyx, mss.c) -- an exact parallelisation of a double-precision
// left fold; DESIGN.md 3.3.  Compiled with -ffp-contract=off: the float64 expressions below must round exactly where mss.c rounds.
#include "dgrp_common.h"
#include "scan.h"
#include <vector>
#include <math.h>

// ------------------------------------------------------------------------------------------
// A10  mss_find_all (deepgrp/_mss/mss.c:50-101) -- "stretch-parallel, fixed-point, wave-chunked" form.
//
// The algorithm is a left fold in double precision (L += S[i]); re-associating it changes
// roundings.  Three exact observations make it parallel without changing a single bit:
//
//  (1) Cutting.  An x-drop reset (mss.c:89-92) returns the scan to a history-free state (stack
//      empty, max = -1e30).  A stretch of non-positive scores whose sum is below -(xdrop + 1)
//      forces such a reset inside it whenever the state was not already history-free (the running
//      max is >= L on entry and cannot grow inside the stretch).  Right after such a stretch the
//      scan remembers only the scalar L.  The sequence is cut into "stretches" at those points.
//  (2) Fixed point.  Every stretch is scanned (by one wave) with the reference's own arithmetic
//      from a guessed entry L; stretch k's guess in pass p is stretch k-1's final L of pass p-1.
//      When a pass changes no final L, induction over k shows every stretch started from the true
//      L, i.e. the result IS the sequential result.  A reset makes the final L independent of the
//      entry L, so two passes are the norm; one stretch (= the sequential scan) is the worst case.
//  (3) Chunks.  Inside a stretch the wave takes 64 scores at a time.  If every score of the chunk
//      and L are integer multiples of 2^-Q and |L| + sum|s| < 2^(53-Q), every partial sum the
//      reference forms is exactly representable, so its left fold equals the exact prefix sum and
//      a wave prefix scan reproduces it bit for bit ("certificate").  Positive runs, x-drop
//      triggers and run ends are then found with ballots; only the per-run stack work stays
//      serial.  Whole all-non-positive 64-blocks are skipped 64 at a time in the history-free
//      state under the same certificate.  A chunk whose certificate fails is folded element by
//      element.
//  Further levels below: flush cuts (mss_light_kernel), speculative light units, the growing-candidate shortcut, and pieces scanned
//  in parts and stitched (mss_subent, mss_stitch_kernel) -- DESIGN.md 3.3 lists all eight devices.
// ------------------------------------------------------------------------------------------
struct mss_cand { int32_t st, en; double L, R; int32_t pre, pad; };   // mss.c:24-28

#define MSS_SUB_MIN 256        // light units of at least this many 64-blocks may split a piece (sizes the dump slots)
#define MSS_SUBENT_BYTES 80
// A unit of the SUB-PIECE scan (r03; DESIGN.md 3.3 device 8): a piece, or the part of a piece behind a speculative light edge
// inside it.  The light walk's converged end states give every such edge its exact entry state, so the stack scan of a long piece
// (scores drifting upward under old peaks: millions of runs on one stack) runs on all its parts at once, each on a LOCAL stack
// above an unknown rest; mss_stitch_kernel then replays only the candidates whose search ran off the local stack's bottom.
struct mss_subent {
    int64_t start, run;       // first score; first stack/segment slot (= runs that start in front of the unit)
    double L, peak, run_L;    // entry state: running value, max since the last flush, the open run's start value
    int64_t run_st;           // the open run's start
    int32_t open, first;      // a run is open at the start; the unit opens its piece (its stack is the reference's own)
    int32_t vci, nst;         // compact index of the unit's speculative edge (dump slot 2 vci; a piece's head: 2 vci(next unit) + 1); OUT: its stack's size
    int32_t reset, pad;       // OUT: an x-drop reset fired (the piece ends with this unit)
    double Lf;                // entry state: the smallest run start value since the last flush (a closed run at or below it flushes)
};
static_assert(sizeof(mss_subent) == 80, "MSS_SUBENT_BYTES");
#define MSS_LCAP 160          // candidates kept in LDS per wave; deeper ones spill to HBM
#define MSS_NEG (-1e30)       // NEG_INF, mss.c:33
#define MSS_QNONE (-4096)
#define MSS_QBAD 4096

struct mss_layout {            // carve of the caller's workspace
    int64_t nblk;
    uint64_t *blk;             // [nblk+1] per-64-block (boundary flag << 32 | positive-run starts), then scanned
    double *blk_sum;           // [nblk] sum of the block's scores
    double *blk_abs;           // [nblk] sum of |score|
    int32_t *blk_q;            // [nblk] smallest Q with every score a multiple of 2^-Q
    uint8_t *flags;            // [nblk] bit0: contains a positive score
    int64_t nsup;              // 64-block groups (4096 scores)
    double *sup_sum, *sup_abs; // [nsup]
    int32_t *sup_q;            // [nsup]
    uint8_t *sup_flags;        // [nsup] bit0: contains a positive score
    uint64_t *tiles;           // scan scratch
    uint64_t *grand;           // [8]: [0] scan total, [1] changed flag, [2] error flag, [3] total kept segments
    int64_t *ustart;           // [nunits+1] stretch starts
    int64_t *urun;             // [nunits+1] first stack/segment slot of each stretch
    double *exitL[2];          // [nunits] final L of each stretch, ping-pong
    uint64_t *segcnt;          // [nunits+1] kept segments per stretch, then scanned
    uint64_t *cutcnt;          // [nunits+1] pieces per stretch (mss_light_kernel), then scanned
    uint64_t *lblk;            // [nblk+1] light-unit edge behind block b, then scanned
    int64_t *lstart, *lrun;    // [nl+1] light units (mss_lightunits_kernel): start, slot of the first run
    uint8_t *lforced;          // [nl+1] the unit starts at a forced reset (a stretch start)
    void *lstate[2];           // [nl] mss_light_state at the end of each unit, ping-pong
    int64_t *cut_st, *cut_run; // [nblk] flush cuts by the 64-block of their start (-1: none): start, slot of the run
    double *cut_L;             // [nblk] L in front of the cut's run
    int2 *cut_kj;              // [nblk] (stretch, ordinal of the piece in its stretch)
    int64_t *ustart2, *urun2;  // [npieces+1] pieces: start, first stack/segment slot  (npieces <= 2*nblk + 1)
    double *entry2;            // [npieces+1] pieces: L at the start; [npieces] = L behind the last element
    void *subs;                // [nsub+1] mss_subent: the pieces cut again at the speculative light edges inside them
    int64_t dump_cap;          // dump slots
    mss_cand *dump;            // [dump_cap][MSS_LCAP] LDS part of a sub-piece's stack at its end (what the stitch reads)
    mss_cand *stack;           // [nruns] overflow stack slots
    int32_t *segs;             // [nruns][2] kept segments in stretch-local slots
    int32_t *segs_out;         // [nruns][2] compacted
    int64_t bytes;
};

static mss_layout mss_carve(void *work, int64_t n)
{
    mss_layout l;
    unsigned char *p = (unsigned char *)work;
    auto take = [&](int64_t bytes) { unsigned char *q = p; p += dgrp_align_up(bytes, 256); return q; };
    l.nblk = (n + 63) / 64;
    const int64_t maxunits = l.nblk + 1, maxruns = n / 2 + 2;
    l.blk = (uint64_t *)take((l.nblk + 1) * 8);
    l.blk_sum = (double *)take(l.nblk * 8);
    l.blk_abs = (double *)take(l.nblk * 8);
    l.blk_q = (int32_t *)take(l.nblk * 4);
    l.flags = (uint8_t *)take(l.nblk + 64);
    l.nsup = (l.nblk + 63) / 64;
    l.sup_sum = (double *)take(l.nsup * 8);
    l.sup_abs = (double *)take(l.nsup * 8);
    l.sup_q = (int32_t *)take(l.nsup * 4);
    l.sup_flags = (uint8_t *)take(l.nsup + 64);
    l.tiles = (uint64_t *)take(((maxunits + 1 + SCAN_TILE - 1) / SCAN_TILE + 1) * 8);
    l.grand = (uint64_t *)take(64);
    l.ustart = (int64_t *)take((maxunits + 1) * 8);
    l.urun = (int64_t *)take((maxunits + 1) * 8);
    l.exitL[0] = (double *)take(maxunits * 8);
    l.exitL[1] = (double *)take(maxunits * 8);
    l.segcnt = (uint64_t *)take((maxunits + 1) * 8);
    l.cutcnt = (uint64_t *)take((maxunits + 1) * 8);
    l.lblk = (uint64_t *)take((l.nblk + 1) * 8);
    l.lstart = (int64_t *)take((maxunits + 1) * 8);
    l.lrun = (int64_t *)take((maxunits + 1) * 8);
    l.lforced = (uint8_t *)take(maxunits + 64);
    l.lstate[0] = take(maxunits * 64);
    l.lstate[1] = take(maxunits * 64);
    l.cut_st = (int64_t *)take(l.nblk * 8);
    l.cut_run = (int64_t *)take(l.nblk * 8);
    l.cut_L = (double *)take(l.nblk * 8);
    l.cut_kj = (int2 *)take(l.nblk * 8);
    l.ustart2 = (int64_t *)take((2 * maxunits + 2) * 8);
    l.urun2 = (int64_t *)take((2 * maxunits + 2) * 8);
    l.entry2 = (double *)take((2 * maxunits + 2) * 8);
    l.subs = take((maxunits + 8) * (int64_t)MSS_SUBENT_BYTES);      // (the sub-piece scan is taken only with at most nblk units)
    l.dump_cap = 2 * (l.nblk / MSS_SUB_MIN + 2);
    l.dump = (mss_cand *)take(l.dump_cap * MSS_LCAP * (int64_t)sizeof(mss_cand));
    l.stack = (mss_cand *)take(maxruns * (int64_t)sizeof(mss_cand));
    l.segs = (int32_t *)take(maxruns * 8);
    l.segs_out = (int32_t *)take(maxruns * 8);
    l.bytes = p - (unsigned char *)work;
    return l;
}

DGRP_EXPORT int64_t dgrp_mss_workspace_bytes(int64_t n)
{
    if (n < 0) return 0;
    return mss_carve(nullptr, n).bytes;
}

// smallest Q such that x is an integer multiple of 2^-Q
__device__ __forceinline__ int quantum_exp(double x)
{
    const uint64_t b = (uint64_t)__double_as_longlong(x) & 0x7fffffffffffffffull;
    if (b == 0) return MSS_QNONE;
    const int e = (int)(b >> 52);
    if (e == 0 || e == 0x7ff) return MSS_QBAD;                 // subnormal / inf / nan: never certified
    const uint64_t m = b & 0x000fffffffffffffull;
    const int tz = m ? __builtin_ctzll(m) : 52;
    return -((e - 1023) - 52 + tz);
}

// all partial sums of {L, s...} exactly representable?
__device__ __forceinline__ bool mss_certified(double L, double sumabs, int q_scores)
{
    const int qL = quantum_exp(L);
    const int Q = qL > q_scores ? qL : q_scores;
    if (Q >= MSS_QBAD) return false;
    const int ex = 53 - Q;
    if (ex > 1000) return true;
    if (ex < -1000) return false;
    return (fabs(L) + sumabs) * 1.0000001 < ldexp(1.0, ex);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int w = __shfl_xor(v, o); v = w > v ? w : v; }
    return v;
}

// value of lane `l` (wave-uniform index) in every lane: v_readlane, no LDS round trip
__device__ __forceinline__ double lane_value(double x, int l)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// One wave per 64-block: statistics used for cutting (1) and skipping (3).
__global__ void __launch_bounds__(256) mss_blockstat_kernel(const double *__restrict__ S, int64_t n,
                                                            uint64_t *__restrict__ blk, double *__restrict__ blk_sum,
                                                            double *__restrict__ blk_abs, int32_t *__restrict__ blk_q,
                                                            uint8_t *__restrict__ flags)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b * 64 >= n) return;
    const int64_t i = b * 64 + lane;
    const double s = i < n ? S[i] : 0.0;
    const bool pos = i < n && s > 0;
    const bool prevpos = i > 0 && i < n && S[i - 1] > 0;
    const unsigned long long mpos = __ballot(pos);
    const unsigned long long mstart = __ballot(pos && !prevpos);
    const double sum = wave_sum(s), sabs = wave_sum(fabs(s));
    const int q = wave_max(quantum_exp(s));
    if (lane == 0) {
        flags[b] = (uint8_t)(mpos != 0ull ? 1 : 0);
        blk[b] = (uint64_t)__popcll(mstart);
        blk_sum[b] = sum;
        blk_abs[b] = sabs;
        blk_q[b] = q;
    }
}

// Second level of the skip table: one wave per 64 blocks (4096 scores).
__global__ void __launch_bounds__(256) mss_superstat_kernel(const double *__restrict__ blk_sum, const double *__restrict__ blk_abs,
                                                            const int32_t *__restrict__ blk_q, const uint8_t *__restrict__ flags,
                                                            int64_t nblk, double *__restrict__ sup_sum, double *__restrict__ sup_abs,
                                                            int32_t *__restrict__ sup_q, uint8_t *__restrict__ sup_flags)
{
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g * 64 >= nblk) return;
    const int64_t b = g * 64 + lane;
    const bool in = b < nblk;
    const unsigned long long mp = __ballot(in && (flags[b] & 1));
    const double sum = wave_sum(in ? blk_sum[b] : 0.0), sabs = wave_sum(in ? blk_abs[b] : 0.0);
    const int q = wave_max(in ? blk_q[b] : MSS_QNONE);
    if (lane == 0) {
        sup_flags[g] = (uint8_t)(mp != 0ull ? 1 : 0);
        sup_sum[g] = sum;
        sup_abs[g] = sabs;
        sup_q[g] = q;
    }
}

// A cut goes after block b when b ends a chain of all-non-positive blocks, the next block has a
// positive score, and the last (up to 16) blocks of the chain sum below -thr (forced reset).
__device__ __forceinline__ bool mss_is_boundary(const uint8_t *flags, const double *blk_sum, int64_t b, int64_t nblk,
                                                int64_t n, double thr)
{
    if (!(thr > 0.0) || b + 1 >= nblk || (flags[b] & 1) || !(flags[b + 1] & 1)) return false;
    if (b * 64 + 64 > n) return false;
    double cum = 0.0;
    for (int64_t j = b; j >= 0 && j > b - 16; --j) {
        if (flags[j] & 1) break;
        cum += blk_sum[j];
        if (cum < -thr) return true;
    }
    return false;
}

__global__ void __launch_bounds__(256) mss_boundary_kernel(const uint8_t *__restrict__ flags, const double *__restrict__ blk_sum,
                                                           int64_t nblk, int64_t n, double thr, uint64_t *__restrict__ blk)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    if (mss_is_boundary(flags, blk_sum, b, nblk, n, thr)) blk[b] |= 1ull << 32;
}

// after the exclusive scan of blk: block b's entry holds (#boundaries before b, #run starts before b)
__global__ void __launch_bounds__(256) mss_units_kernel(const uint64_t *__restrict__ blk, const uint8_t *__restrict__ flags,
                                                        const double *__restrict__ blk_sum, int64_t nblk, int64_t n, double thr,
                                                        int64_t *__restrict__ ustart, int64_t *__restrict__ urun,
                                                        const uint64_t *__restrict__ grand)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) {
        ustart[0] = 0;
        urun[0] = 0;
        const int64_t nunits = (int64_t)(grand[0] >> 32) + 1;
        ustart[nunits] = n;
        urun[nunits] = (int64_t)(grand[0] & 0xffffffffull);
    }
    if (b >= nblk) return;
    if (mss_is_boundary(flags, blk_sum, b, nblk, n, thr)) {
        const int64_t k = (int64_t)(blk[b] >> 32) + 1;          // this cut opens stretch k
        ustart[k] = (b + 1) * 64;
        urun[k] = (int64_t)(blk[b] & 0xffffffffull);            // block b itself starts no run
    }
}

// Wave-wide scans on DPP moves (v_mov_b32_dpp: a few cycles each; __shfl_up is an LDS round trip of ~100 cycles, and the light walk is
// ONE wave working through its stretch alone: latency is all it has).  row_shr:n shifts inside a row of 16 lanes, row_bcast:15 /
// row_bcast:31 carry a row's last lane into the rows behind it (gfx9 family), wave_shr:1 shifts the whole wave by one lane.
// Lanes without a source keep `old` (bound_ctrl off), which every scan sets to its identity.
template <int CTRL, int RMASK>
__device__ __forceinline__ int dpp_i(int src, int old) { return __builtin_amdgcn_update_dpp(old, src, CTRL, RMASK, 0xf, false); }
template <int CTRL, int RMASK>
__device__ __forceinline__ double dpp_d(double src, double old)
{
    const long long s = __double_as_longlong(src), o = __double_as_longlong(old);
    const int lo = __builtin_amdgcn_update_dpp((int)o, (int)s, CTRL, RMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(s >> 32), CTRL, RMASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
#define DPP_SHR(n) (0x110 + (n))
#define DPP_BCAST15 0x142
#define DPP_BCAST31 0x143
#define DPP_WAVE_SHR1 0x138
// the six steps of an inclusive scan: STEP(ctrl, row_mask) combines each lane with the lane the control word points at
#define DPP_SCAN_STEPS(STEP) STEP(DPP_SHR(1), 0xf) STEP(DPP_SHR(2), 0xf) STEP(DPP_SHR(4), 0xf) STEP(DPP_SHR(8), 0xf) STEP(DPP_BCAST15, 0xa) STEP(DPP_BCAST31, 0xc)
__device__ __forceinline__ double scan_add_d(double x)
{
#define STEP(c, m) x = x + dpp_d<c, m>(x, 0.0);
    DPP_SCAN_STEPS(STEP)
#undef STEP
    return x;
}
__device__ __forceinline__ double scan_min_d(double x)
{
#define STEP(c, m) { const double y = dpp_d<c, m>(x, INFINITY); x = y < x ? y : x; }
    DPP_SCAN_STEPS(STEP)
#undef STEP
    return x;
}
__device__ __forceinline__ int scan_max_i(int x)
{
#define STEP(c, m) { const int y = dpp_i<c, m>(x, -1); x = y > x ? y : x; }
    DPP_SCAN_STEPS(STEP)
#undef STEP
    return x;
}
// segmented running maximum: a set flag restarts the maximum at its lane's value
__device__ __forceinline__ void scan_segmax_d(int &fl, double &v)
{
#define STEP(c, m) { const double yv = dpp_d<c, m>(v, -INFINITY); const int yf = dpp_i<c, m>(fl, 0); if (!fl) v = yv > v ? yv : v; fl |= yf; }
    DPP_SCAN_STEPS(STEP)
#undef STEP
}
// the value one lane down (lane 0: `first`)
__device__ __forceinline__ double lane_below_d(double x, double first) { return dpp_d<DPP_WAVE_SHR1, 0xf>(x, first); }
__device__ __forceinline__ int lane_below_i(int x, int first) { return dpp_i<DPP_WAVE_SHR1, 0xf>(x, first); }
__device__ __forceinline__ int lane_value_i(int x, int l) { return __builtin_amdgcn_readlane(x, l); }

// the reference's fold L += S[i] (mss.c:56-93) over lanes t..nvalid-1 of a chunk, starting from a0 in front of lane t: every
// round each lane adds its element to the value one lane down, so lane t+j is final after j+1 rounds; lanes below t idle at a0
// (their element counts as 0), which is also what lane t finds one lane down.  One v_add_f64 and two DPP moves per element.
__device__ __forceinline__ double mss_fold(double s, double a0, int t, int nvalid, int lane)
{
    const double sm = lane >= t && lane < nvalid ? s : 0.0;
    double X = a0;
    for (int j = t; j < nvalid; j += 4) {                    // (rounds beyond the last needed one change nothing)
        X = lane_below_d(X, a0) + sm;
        X = lane_below_d(X, a0) + sm;
        X = lane_below_d(X, a0) + sm;
        X = lane_below_d(X, a0) + sm;
    }
    return X;
}

// One wave per stretch.
__global__ void __launch_bounds__(64) mss_scan_kernel(const double *__restrict__ S, const int64_t *__restrict__ ustart,
                                                      const int64_t *__restrict__ urun, int64_t nunits,
                                                      const double *__restrict__ exit_prev, double *__restrict__ exit_cur,
                                                      mss_cand *stack_all, int32_t *__restrict__ segs_all,
                                                      uint64_t *__restrict__ segcnt, int min_sc, double xdrop,
                                                      uint64_t *__restrict__ grand, int pass,
                                                      const double *__restrict__ blk_sum, const double *__restrict__ blk_abs,
                                                      const int32_t *__restrict__ blk_q, const uint8_t *__restrict__ flags,
                                                      const double *__restrict__ sup_sum, const double *__restrict__ sup_abs,
                                                      const int32_t *__restrict__ sup_q, const uint8_t *__restrict__ sup_flags,
                                                      int have_stats, int independent, int cutmode,
                                                      mss_subent *subs, mss_cand *dump_all)
{
    __shared__ double2 sLR[MSS_LCAP];                           // (L, R)
    __shared__ int4 sIdx[MSS_LCAP];                             // (st, en, pre, left-open)
    const int lane = threadIdx.x;
    const int64_t k = blockIdx.x;
    // sub-piece mode (subs != NULL, with cutmode): unit k is subs[k]; `head` = it opens its piece, `tail` = it ends it.  A unit that is
    // not a head scans on a LOCAL stack: a candidate whose search runs off its bottom is pushed LEFT-OPEN (pad = 1) instead of flushing
    // -- no flush can happen there (it would have been filed as a cut) -- and the stack is left for mss_stitch_kernel.
    const bool sub = subs != nullptr;
    const bool head = !sub || subs[k].first != 0;
    const bool tail = !sub || k + 1 == nunits || subs[k + 1].first != 0;
    const int64_t begin = sub ? subs[k].start : ustart[k], end = sub ? subs[k + 1].start : ustart[k + 1];
    const int64_t slot0 = sub ? subs[k].run : urun[k];
    volatile mss_cand *ovf = stack_all + slot0;                 // slots MSS_LCAP.. of this stretch's stack
    int32_t *segs = segs_all + 2 * slot0;
    int64_t nst = 0, nseg = 0;
    // independent: every unit is a record of its own (a batch of records side by side): L starts at 0, the unit
    // ends with the end-of-sequence flush of mss.c:96, nothing is handed on
    double cur = sub ? subs[k].L : (k == 0 || independent) ? 0.0 : exit_prev[k - 1];   // L outside a run, R inside one
    double peak = sub && !head ? subs[k].peak : MSS_NEG;
    bool run_open = sub && !head && subs[k].open != 0;
    int64_t run_st = run_open ? subs[k].run_st : 0;
    double run_L = run_open ? subs[k].run_L : 0.0;
    bool was_reset = false;
    // (a part behind an edge cannot see the true stack's bottom: it follows the flush level as the light walk does -- mss.c:78-81 sets
    // max = R there, and the x-drop test reads max.  Only a run that was open across the edge can flush behind it: every other flush
    // is a cut)
    double Lf = sub && !head ? subs[k].Lf : INFINITY;

    struct cand_v { double L, R; int32_t st, en, pre, open; };
    auto get = [&](int64_t j) -> cand_v {
        cand_v c;
        if (j < MSS_LCAP) {
            const double2 lr = sLR[j];
            const int4 ix = sIdx[j];
            c.L = lr.x; c.R = lr.y; c.st = ix.x; c.en = ix.y; c.pre = ix.z; c.open = ix.w;
        } else {
            volatile mss_cand *o = ovf + (j - MSS_LCAP);
            c.L = o->L; c.R = o->R; c.st = o->st; c.en = o->en; c.pre = o->pre; c.open = o->pad;
        }
        return c;
    };

    // mss.c:35-47, lane-parallel over the stack
    auto flush = [&]() {
        for (int64_t base = 0; base < nst; base += 64) {
            const int64_t j = base + lane;
            bool keep = false;
            int32_t a = 0, b = 0;
            if (j < nst) {
                const cand_v c = get(j);
                keep = c.R - c.L >= min_sc;
                a = c.st;
                b = c.en;
            }
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int64_t slot = nseg + __popcll(m & ((1ull << lane) - 1ull));
                segs[2 * slot] = a;
                segs[2 * slot + 1] = b;
            }
            nseg += __popcll(m);
        }
        nst = 0;
    };

    // a positive run [run_st, en) with prefix values run_L (before) and R (after) is complete: mss.c:65-86
    auto close_run = [&](double R, int64_t en) {
        if (R > peak) peak = R;
        if (!head && run_L <= Lf) { Lf = run_L; peak = R; }          // a flush in truth (the stitch does it; the search below runs off the bottom)
        int32_t tst = (int32_t)run_st;
        double tL = run_L;
        int64_t j;
        for (;;) {
            j = nst - 1;
            cand_v c;
            while (j >= 0) {
                c = get(j);
                if (c.L < tL) break;
                j = c.pre >= 0 ? c.pre : j - 1;
            }
            if (j >= 0 && c.R < R) {
                tst = c.st;
                tL = c.L;
                nst = j;
                continue;
            }
            break;
        }
        const int lopen = j < 0 && !head ? 1 : 0;                // the search ran off a local stack: the stitch goes on from here
        if (j < 0 && head) { flush(); peak = R; }
        if (was_reset && lane == 0) { atomicOr((unsigned long long *)&grand[2], 1ull); atomicOr((unsigned long long *)&grand[4], 8ull); }   // (a run behind a reset opens the next piece)
        if (nst < MSS_LCAP) {
            if (lane == 0) {
                sLR[nst] = make_double2(tL, R);
                sIdx[nst] = make_int4(tst, (int32_t)en, (int32_t)j, lopen);
            }
        } else {
            // every lane stores the same record, so each lane later reads what it wrote itself
            volatile mss_cand *c = ovf + (nst - MSS_LCAP);
            c->st = tst; c->en = (int32_t)en; c->L = tL; c->R = R; c->pre = (int32_t)j; c->pad = lopen;
        }
        __threadfence_block();
        ++nst;
        run_open = false;
    };
    // x-drop reset (mss.c:89-92): everything is moved out.  On a local stack that is the stitch's job: the piece ends with this unit
    // (the next closed run is a cut), so the flush at the end of the piece IS this one.
    auto reset_flush = [&]() {
        if (head) flush(); else { was_reset = true; Lf = INFINITY; }
    };

    int64_t pos = begin;
    while (pos < end) {
        const int nvalid = (int)min((int64_t)64, end - pos);
        // ---- (3b) skip whole all-non-positive blocks in the history-free state --------------
        if (have_stats && !run_open && peak == MSS_NEG && (pos & 63) == 0) {
            if ((pos & 4095) == 0) {
                // 64 groups of 4096 scores at a time
                const int64_t gj = (pos >> 12) + lane;
                const bool stop2 = gj * 4096 + 4096 > end || (sup_flags[gj] & 1);
                const unsigned long long ms2 = __ballot(stop2);
                const int m2 = ms2 ? __builtin_ctzll(ms2) : 64;
                if (m2 > 0) {
                    const bool in = lane < m2;
                    const double gs = wave_sum(in ? sup_sum[gj] : 0.0);
                    const double ga = wave_sum(in ? sup_abs[gj] : 0.0);
                    const int gq = wave_max(in ? sup_q[gj] : MSS_QNONE);
                    if (mss_certified(cur, ga, gq)) {
                        cur += gs;
                        pos += (int64_t)m2 * 4096;
                        continue;
                    }
                }
            }
            const int64_t b0 = pos >> 6, bend = (end + 63) >> 6;
            const int64_t bj = b0 + lane;
            // stop at the next 4096-aligned position so that the coarser level takes over there
            const bool stop = bj >= bend || (flags[bj] & 1) || (bj * 64 + 64 > end) || (lane > 0 && (bj & 63) == 0);
            const unsigned long long mstop = __ballot(stop);
            const int m = mstop ? __builtin_ctzll(mstop) : 64;
            if (m > 0) {
                const bool in = lane < m;
                const double gs = wave_sum(in ? blk_sum[bj] : 0.0);
                const double ga = wave_sum(in ? blk_abs[bj] : 0.0);
                const int gq = wave_max(in ? blk_q[bj] : MSS_QNONE);
                if (mss_certified(cur, ga, gq)) {
                    cur += gs;
                    pos += (int64_t)m * 64;
                    continue;
                }
            }
        }
        const double s = lane < nvalid ? S[pos + lane] : 0.0;
        const bool ispos = lane < nvalid && s > 0;
        const unsigned long long vmask = nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull);
        const unsigned long long mpos = __ballot(ispos);
        const double sabs = wave_sum(fabs(s));
        const int q = wave_max(quantum_exp(s));
        const bool cert = mss_certified(cur, sabs, q);
        if (cert || (q < MSS_QBAD && quantum_exp(cur) < MSS_QBAD)) {
            // ---- (3a) V_i = base + pre_i is the reference's running value after element i: exact prefix sums where the
            // certificate holds; elsewhere the fold itself, lane after lane with the reference's roundings (mss_fold; base = 0)
            double pre = s;
            if (cert) {
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const double y = __shfl_up(pre, o);
                    if (lane >= o) pre += y;
                }
            } else {
                pre = mss_fold(s, cur, 0, nvalid, lane);
            }
            double base = cert ? cur : 0.0;
            const double front = cur;                             // the value in front of the chunk
            int p = 0;
            // ---- (3c) growing-candidate shortcut.  A run whose start lies above the TOP candidate's L and whose end exceeds its R
            // is absorbed by it (mss.c:68-76: found, merged, popped); the search then goes on from the top's own left neighbour b
            // (its `pre`: the nearest candidate with a smaller L -- the merged run carries the top's L, so it finds the same one):
            //   * no such neighbour (one candidate on the stack): "flush", pushed again with R = the run's end, max = R (mss.c:78-81);
            //   * a neighbour whose R the run's end does NOT exceed: pushed again on top of it with R = the run's end -- the stack as
            //     it was with a larger top (scores drifting upward under an old high peak: every run does exactly this, and
            //     run by run it was ~1 us of serial stack work each: 2.5 M runs of normal(+0.3, 1) scores took 2.8 s);
            //   * a neighbour the run's end exceeds: a second merge -- not handled here.
            // If every run that closes in the chunk does one of the first two -- run ends strictly increasing and above the top's R,
            // starts above its L, ends not above the neighbour's R, no x-drop trigger -- the chunk only moves the top's R / en and
            // max: no serial stack work.  A run may reach into the chunk (it closes at its first non-positive element, lane 0
            // included) and one may reach out of it (it stays open).
            if (nst >= 1 && nst <= MSS_LCAP) {
                const cand_v c0 = get(nst - 1);
                const int64_t nb = c0.pre;
                // (a LEFT-OPEN top, on a local stack: merged into it the run's search runs off the bottom again -- the top pushed again,
                // left-open, with the run's end; no flush, max as with a neighbour that is never exceeded)
                const bool lone = nst == 1 && !c0.open;                            // (then pre = -1)
                double lim = INFINITY;                                             // the neighbour's R: an end above it merges again
                if (!lone && nb >= 0) lim = get(nb).R;
                if (lone || nb >= 0 || c0.open) {
                const double V = base + pre;
                const bool edge_close = run_open && !(mpos & 1ull);          // the open run ended with the previous chunk
                const bool isend = ispos && lane + 1 < nvalid && !((mpos >> (lane + 1)) & 1ull);
                const bool isstart = ispos && (lane == 0 ? !run_open : !((mpos >> (lane - 1)) & 1ull));
                double vprev = __shfl_up(V, 1);
                if (lane == 0) vprev = cur;
                double rmax = isend ? V : -INFINITY;              // inclusive running max of run-end values
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const double y = __shfl_up(rmax, o);
                    if (lane >= o) rmax = y > rmax ? y : rmax;
                }
                double before = __shfl_up(rmax, 1);               // run ends strictly before this lane
                if (lane == 0) before = -INFINITY;
                const double base_r = edge_close ? cur : c0.R;
                // max in front of this lane: one candidate: every run close sets it to the run's end (flush); else the running maximum
                const double base_peak = edge_close ? (lone ? cur : (cur > peak ? cur : peak)) : peak;
                const double top_r = before > base_r ? before : base_r;
                const double peak_i = lone ? (before > -INFINITY ? before : base_peak) : (before > base_peak ? before : base_peak);
                bool ok = !run_open || c0.L < run_L;                             // the open run's own start (p->L < t.L)
                if (edge_close) ok = ok && cur > c0.R && !(lim < cur);
                if (isend) ok = ok && V > top_r && !(lim < V);                    // seg.a[j].R < t.R for the top, not for its neighbour
                if (isstart) ok = ok && (c0.L < vprev);                          // p->L < t.L
                if (lane < nvalid && !ispos && xdrop > 0.0 && V + xdrop < peak_i) ok = false;   // x-drop would fire
                if (__all(ok)) {
                    const unsigned long long mend = __ballot(isend);
                    if (mend || edge_close) {
                        const int last = mend ? 63 - __builtin_clzll(mend) : -1;  // last positive lane of the last closed run
                        const double newR = mend ? lane_value(V, last) : cur;
                        if (lane == 0) {
                            sLR[nst - 1] = make_double2(c0.L, newR);
                            sIdx[nst - 1] = make_int4(c0.st, (int32_t)(pos + last + 1), (int32_t)nb, c0.open);
                        }
                        __threadfence_block();
                        peak = lone ? newR : (newR > peak ? newR : peak);
                    }
                    if ((mpos >> (nvalid - 1)) & 1ull) {
                        // a run reaches out of the chunk: it starts behind the last non-positive lane (none: it reached in as well)
                        const unsigned long long znon = ~mpos & vmask;
                        if (znon || !run_open) {
                            const int ls = znon ? 64 - __builtin_clzll(znon) : 0;
                            run_st = pos + ls;
                            run_L = ls == 0 ? cur : lane_value(V, ls - 1);
                        }
                        run_open = true;
                    } else {
                        run_open = false;
                    }
                    cur = lane_value(V, nvalid - 1);
                    pos += nvalid;
                    continue;
                }
                }
            }
            while (p < nvalid) {
                const unsigned long long rest = vmask & ~((1ull << p) - 1ull);     // lanes >= p
                if ((mpos >> p) & 1ull) {
                    const unsigned long long z = ~mpos & rest;                     // first non-positive at or after p
                    const int e = z ? __builtin_ctzll(z) : nvalid;
                    if (!run_open) { run_open = true; run_st = pos + p; run_L = p == 0 ? front : base + lane_value(pre, p - 1); }
                    cur = base + lane_value(pre, e - 1);
                    p = e;
                    if (e < nvalid) close_run(cur, pos + e);
                } else {
                    if (run_open) close_run(cur, pos + p);                       // run ended exactly at the chunk edge
                    const unsigned long long z = mpos & rest;
                    const int e = z ? __builtin_ctzll(z) : nvalid;
                    if (xdrop > 0.0 && peak != MSS_NEG) {
                        const bool trig = lane >= p && lane < e && ((base + pre) + xdrop < peak);   // mss.c:89
                        const unsigned long long mt = __ballot(trig);
                        if (mt) {
                            const int t = __builtin_ctzll(mt);
                            reset_flush();
                            peak = MSS_NEG;
                            if (cert) base = t == 0 ? 0.0 : -lane_value(pre, t - 1);   // L = 0 before S[t] is added
                            else pre = mss_fold(s, 0.0, t, nvalid, lane);
                        }
                    }
                    cur = base + lane_value(pre, e - 1);
                    p = e;
                }
            }
        } else {
            // ---- certificate failed: the reference loop, element by element ------------------
            for (int i = 0; i < nvalid; ++i) {
                const double v = lane_value(s, i);
                if (v > 0) {
                    if (!run_open) { run_open = true; run_st = pos + i; run_L = cur; }
                    cur = cur + v;
                } else {
                    if (run_open) close_run(cur, pos + i);
                    if (xdrop > 0.0 && cur + v + xdrop < peak) { reset_flush(); cur = 0.0; peak = MSS_NEG; }
                    cur += v;
                }
            }
        }
        pos += nvalid;
    }
    if (run_open && tail) close_run(cur, end);                // (a run open at a speculative edge goes on in the next unit)
    if (sub && !(head && tail)) {
        // part of a longer piece: the stack stays as it is for mss_stitch_kernel -- its first MSS_LCAP candidates from LDS to the
        // unit's dump slot (a head: the slot behind its first edge's), the deeper ones are in the unit's own stack slots already
        const int64_t slot = head ? 2 * (int64_t)subs[k + 1].vci + 1 : 2 * (int64_t)subs[k].vci;
        mss_cand *d = dump_all + slot * MSS_LCAP;
        for (int64_t j = lane; j < nst && j < MSS_LCAP; j += 64) {
            const double2 lr = sLR[j];
            const int4 ix = sIdx[j];
            mss_cand c;
            c.st = ix.x; c.en = ix.y; c.L = lr.x; c.R = lr.y; c.pre = ix.z; c.pad = ix.w;
            d[j] = c;
        }
        if (lane == 0) {
            subs[k].nst = (int32_t)(nst < (1ll << 31) ? nst : -1);
            subs[k].reset = was_reset ? 1 : 0;
        }
    } else if (k == nunits - 1 || independent || cutmode) {
        flush();                                              // (cutmode: the next piece opens with a flush, mss_light_kernel)
    } else if (nst != 0 || peak != MSS_NEG) {
        if (lane == 0) atomicOr((unsigned long long *)&grand[2], 1ull);   // forced-reset argument failed: caller falls back
    }
    if (lane == 0) {
        segcnt[k] = (uint64_t)nseg;
        const double want = sub ? subs[k + 1].L : exit_prev[k];
        if (!independent && (pass == 0 || __double_as_longlong(want) != __double_as_longlong(cur)))
            atomicOr((unsigned long long *)&grand[1], 1ull);
        exit_cur[k] = cur;
    }
}

// The stitch of a piece that was scanned in parts (one wave per piece head).  The head's stack is the reference's own; every later
// part left a local stack whose candidates are CLOSED -- their search stopped inside the part: every comparison they made holds
// against the true stack as well (a left-open candidate's true L can only be smaller than its local one, its R is exact) -- or
// LEFT-OPEN: their search ran off the local bottom.  In order: a closed candidate is appended (its `pre` moved by the distance its
// part's bottom-most left-open candidate in front of it moved); a left-open one runs mss.c:66-86 against the true stack.  A
// left-open candidate that grew by absorbing later runs stands for the whole chain of its extensions: R only grew along the chain, so
// what the last extension merges with contains what every earlier one merged with, and the candidates pushed in between were absorbed.
__global__ void __launch_bounds__(64) mss_stitch_kernel(mss_subent *subs, int64_t nunits, mss_cand *stack_all, const mss_cand *dump_all,
                                                        int32_t *__restrict__ segs_all, uint64_t *__restrict__ segcnt, int min_sc,
                                                        uint64_t *__restrict__ grand)
{
    const int lane = threadIdx.x;
    const int64_t k = blockIdx.x;
    if (!subs[k].first || k + 1 >= nunits || subs[k + 1].first) return;           // heads of pieces with more than one part only
    __shared__ mss_cand loc[256];
    volatile mss_cand *T = stack_all + subs[k].run;                                // the true stack, in the piece's own slots
    int64_t nT = subs[k].nst;
    bool bad = nT < 0;
    // the head's stack: slots j - MSS_LCAP hold candidate j >= MSS_LCAP; move them up (from the top down), then the LDS part in front
    if (!bad) {
        for (int64_t hi = nT; hi > MSS_LCAP; hi -= 64) {
            const int64_t j = hi - 1 - lane;                                           // this lane's candidate of the chunk [hi - 64, hi)
            mss_cand c;
            const bool on = j >= MSS_LCAP;
            if (on) { volatile mss_cand *o = T + (j - MSS_LCAP); c.st = o->st; c.en = o->en; c.L = o->L; c.R = o->R; c.pre = o->pre; c.pad = 0; }
            __threadfence_block();
            if (on) { volatile mss_cand *o = T + j; o->st = c.st; o->en = c.en; o->L = c.L; o->R = c.R; o->pre = c.pre; o->pad = 0; }
            __threadfence_block();
        }
        const mss_cand *d = dump_all + (2 * (int64_t)subs[k + 1].vci + 1) * MSS_LCAP;
        for (int64_t j = lane; j < nT && j < MSS_LCAP; j += 64) {
            const mss_cand c = d[j];
            volatile mss_cand *o = T + j; o->st = c.st; o->en = c.en; o->L = c.L; o->R = c.R; o->pre = c.pre; o->pad = 0;
        }
        __threadfence();
    }
    int64_t nseg = (int64_t)segcnt[k];
    int32_t *segs = segs_all + 2 * subs[k].run;
    // mss.c:35-47 over the true stack
    auto flush_T = [&]() {
        for (int64_t base = 0; base < nT; base += 64) {
            const int64_t j = base + lane;
            bool keep = false;
            int32_t a = 0, b = 0;
            if (j < nT) {
                volatile mss_cand *c = T + j;
                keep = c->R - c->L >= min_sc;
                a = c->st;
                b = c->en;
            }
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int64_t slot = nseg + __popcll(m & ((1ull << lane) - 1ull));
                segs[2 * slot] = a;
                segs[2 * slot + 1] = b;
            }
            nseg += __popcll(m);
        }
        nT = 0;
    };
    for (int64_t u = k + 1; !bad && u < nunits && !subs[u].first; ++u) {
        const int64_t nl = subs[u].nst;
        if (nl < 0) { bad = true; break; }
        // the part's candidates pass through a window of 256 in LDS, loaded 64 at a time well ahead of their turn: its deeper ones
        // (index >= MSS_LCAP) lie in the part's own stack slots, which the true stack may grow into -- by at most one slot per
        // candidate processed, from a start at or below the part's first slot (checked)
        const mss_cand *d = dump_all + 2 * (int64_t)subs[u].vci * MSS_LCAP;
        const mss_cand *deep = stack_all + subs[u].run;
        if (nl > MSS_LCAP && subs[k].run + nT > subs[u].run) { bad = true; break; }
        int64_t w = 0;
        auto load_chunk = [&]() {
            const int64_t j = w + lane;
            if (j < nl) loc[j & 255] = j < MSS_LCAP ? d[j] : deep[j - MSS_LCAP];
            w += 64;
        };
        __syncthreads();
        while (w < nl && w < 256) load_chunk();
        __syncthreads();
        int64_t delta = 0;
        for (int64_t i = 0; i < nl && !bad; ++i) {
            if (w < nl && i + 192 >= w) {
                if (subs[k].run + nT > subs[u].run + w - MSS_LCAP) { bad = true; break; }
                __syncthreads();
                load_chunk();
                __syncthreads();
            }
            const mss_cand e = loc[i & 255];
            int32_t tst = e.st, pre;
            double tL = e.L;
            if (e.pad) {
                int64_t j;
                for (;;) {
                    j = nT - 1;
                    double cL = 0.0, cR = 0.0;
                    int32_t cst = 0;
                    while (j >= 0) {
                        volatile mss_cand *c = T + j;
                        cL = c->L;
                        if (cL < tL) { cR = c->R; cst = c->st; break; }
                        const int32_t pr = c->pre;
                        j = pr >= 0 ? pr : j - 1;
                    }
                    if (j >= 0 && cR < e.R) { tst = cst; tL = cL; nT = j; continue; }
                    break;
                }
                // no candidate with a smaller L: mss.c:78-81 moves the stack out -- nothing, if the candidate has absorbed it down to the
                // piece's bottom; else a run that was open across the edge flushes (any other flush would have been a cut)
                if (j < 0 && nT > 0) flush_T();
                pre = (int32_t)j;
                delta = nT - i;
            } else {
                if (i == 0 || e.pre < 0) { bad = true; if (lane == 0) atomicOr((unsigned long long *)&grand[4], 4ull); break; }   // (a local stack starts with a left-open candidate)
                pre = (int32_t)(e.pre + delta);
            }
            volatile mss_cand *o = T + nT;
            o->st = tst; o->en = e.en; o->L = tL; o->R = e.R; o->pre = pre; o->pad = 0;
            __threadfence();
            ++nT;
        }
    }
    if (bad) {
        if (lane == 0) atomicOr((unsigned long long *)&grand[2], 1ull);
        return;
    }
    // the flush at the piece's end (mss.c:35-47), behind what the head and the stitch flushed on the way
    flush_T();
    if (lane == 0) segcnt[k] = (uint64_t)nseg;
}

// ------------------------------------------------------------------------------------------
// Second level of cutting: FLUSH CUTS.  mss.c:68-81: a closed run t searches the candidate stack for the rightmost candidate
// with L < t.L; if there is none, every candidate is moved out (move_segs) and t starts a new stack with max = t.R.  The
// smallest L on the stack is always that of its bottom candidate -- every later candidate was pushed on top of one with a
// smaller L, and a merge only hands that L on -- so "none" means  t.L <= Lf,  Lf = L of the run that caused the previous
// flush (+inf on an empty stack).  Such a run is a cut: the stack work on either side of it is independent, and what the
// scan needs on the right side is known without the stack: L (the running fold), max, Lf.
// This kernel is the reference loop WITHOUT the stack: one wave per light unit walks it in 64-element chunks tracking
// (L, max, Lf, the open run) with wave scans over the closed runs of the chunk, no per-run serial work -- the running values
// from exact prefix sums where the chunk is certified (see (3) above), else from the fold itself, lane after lane (mss_fold);
// only chunks with subnormal / non-finite values go element by element -- and files the cuts (at most one per aligned
// 64-block), each with its entry L and the slot of its run.  Input that never resets by x-drop (noise-like scores: one
// stretch of millions of runs, ~270 ns of serial stack work per run) thus splits into thousands of independent pieces.
// Passes: a unit starts from the END STATE (mss_light_state) the unit in front of it reached in the previous pass; when no
// end state changes, every unit started from its predecessor's true state (induction from unit 0).  Every walk files its
// cuts by 64-block; mss_pieces_kernel lines up those of the converged walk.
// ------------------------------------------------------------------------------------------
#define MSS_LIGHT_SUB 2048         // blocks per light unit (131 072 scores: a pass costs a few ms whatever the record's length)
// what a walk hands to the unit behind it
struct mss_light_state {
    double L, peak, Lf, run_L;          // running value; max and smallest start value since the last flush; the open run's start value
    int64_t run_st, run_slot;           // the open run: start position, stack/segment slot (= index among all runs of the record)
    int64_t last_blk, open;             // 64-block of the last filed cut; is a run open
};

// light units: the stretches (forced resets, mss_is_boundary) cut further every MSS_LIGHT_SUB blocks.  Behind such a SPECULATIVE
// edge nothing is empty: the unit starts from the state the unit in front of it ended with in the previous pass, and the passes
// repeat until no unit's end state changes (then every unit started from its predecessor's true end state: induction from unit 0).
// Convergence is fast wherever the input resets by x-drop now and then: two walks that differ only in their start state agree
// (up to a constant shift of L, max and Lf, which no decision sees while sums are exact) from the first flush behind the edge,
// and bit for bit from the first x-drop reset behind that.  Input that never resets needs as many passes as it has units in a
// row -- no slower than the one-wave walk it replaces.
__global__ void __launch_bounds__(256) mss_lightflag_kernel(const uint8_t *__restrict__ flags, const double *__restrict__ blk_sum,
                                                            int64_t nblk, int64_t n, double thr, int sub, int use_forced,
                                                            uint64_t *__restrict__ lblk)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const bool forced = use_forced && mss_is_boundary(flags, blk_sum, b, nblk, n, thr);
    const bool spec = ((b + 1) % sub) == 0 && (b + 1) * 64 < n;
    lblk[b] = forced || spec ? 1 : 0;
}

// lblk scanned (exclusive); runs_before[b] = low word of the scanned blk (run starts in front of block b), total in runs_total
__global__ void __launch_bounds__(256) mss_lightunits_kernel(const uint64_t *__restrict__ lblk, const uint64_t *__restrict__ lcount,
                                                             const uint8_t *__restrict__ flags, const double *__restrict__ blk_sum,
                                                             const uint64_t *__restrict__ runs_before, const uint64_t *__restrict__ runs_total,
                                                             int64_t nblk, int64_t n, double thr, int sub, int use_forced,
                                                             int64_t *__restrict__ lstart, int64_t *__restrict__ lrun,
                                                             uint8_t *__restrict__ lforced)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) {
        lstart[0] = 0; lrun[0] = 0; lforced[0] = 1;
        const int64_t nl = (int64_t)*lcount + 1;
        lstart[nl] = n; lrun[nl] = (int64_t)(*runs_total & 0xffffffffull); lforced[nl] = 1;
    }
    if (b >= nblk) return;
    const bool forced = use_forced && mss_is_boundary(flags, blk_sum, b, nblk, n, thr);
    const bool spec = ((b + 1) % sub) == 0 && (b + 1) * 64 < n;
    if (forced || spec) {
        const int64_t j = (int64_t)lblk[b] + 1;
        lstart[j] = (b + 1) * 64;
        lrun[j] = (int64_t)((b + 1 < nblk ? runs_before[b + 1] : *runs_total) & 0xffffffffull);
        lforced[j] = forced ? 1 : 0;
    }
}

__global__ void __launch_bounds__(64) mss_light_kernel(const double *__restrict__ S, const int64_t *__restrict__ ustart,
                                                       const int64_t *__restrict__ urun, const uint8_t *__restrict__ uforced,
                                                       int64_t nunits, const mss_light_state *__restrict__ st_prev,
                                                       mss_light_state *__restrict__ st_cur,
                                                       double xdrop, uint64_t *__restrict__ grand, int pass,
                                                       const double *__restrict__ blk_sum, const double *__restrict__ blk_abs,
                                                       const int32_t *__restrict__ blk_q, const uint8_t *__restrict__ flags,
                                                       const double *__restrict__ sup_sum, const double *__restrict__ sup_abs,
                                                       const int32_t *__restrict__ sup_q, const uint8_t *__restrict__ sup_flags,
                                                       uint64_t *__restrict__ cut_cnt, int64_t *__restrict__ cut_st,
                                                       int64_t *__restrict__ cut_run, double *__restrict__ cut_L,
                                                       int2 *__restrict__ cut_kj)
{
    const int lane = threadIdx.x;
    const int64_t k = blockIdx.x;
    const int64_t begin = ustart[k], end = ustart[k + 1];
    const int64_t slot0 = urun[k];                            // slot of the first run that starts in this unit
    const bool forced = uforced[k] != 0;
    // entry state: empty for the first unit and in the first pass (a guess), else what the unit in front ended with last pass
    double cur = 0.0, peak = MSS_NEG, Lf = INFINITY, run_L = 0.0;
    bool run_open = false;
    int64_t run_st = 0, run_slot = 0;                         // the open run: its start and its slot
    int64_t last_blk = begin >> 6;                            // 64-block of the last reported piece start
    if (k > 0 && pass > 0) {
        const mss_light_state e = st_prev[k - 1];
        cur = e.L; peak = e.peak; Lf = e.Lf; run_L = e.run_L;
        run_st = e.run_st; run_slot = e.run_slot; run_open = e.open != 0;
        if (!forced) last_blk = e.last_blk;
    }
    int64_t nruns = 0;                                        // run starts seen so far in this unit
    int64_t ncut = forced ? 1 : 0;                            // pieces so far (a forced unit's own start is piece 0)
    // an interior piece is filed under the 64-block of its start (one per block; mss_pieces_kernel lines them up)
    auto file_cut = [&](int64_t st, int64_t slot, double Ls) {
        if (lane == 0) {
            const int64_t b = st >> 6;
            cut_st[b] = st; cut_run[b] = slot; cut_L[b] = Ls; cut_kj[b] = make_int2((int)k, (int)ncut);
        }
    };

    // a closed run (start value Ls, end value R, start position st, slot): the reference's bookkeeping minus the stack
    auto close_scalar = [&](double Ls, double R, int64_t st, int64_t slot) {
        if (R > peak) peak = R;
        if (Ls <= Lf) {                                       // no candidate with L < t.L: flush, this run starts a new stack
            Lf = Ls;
            peak = R;
            if ((st >> 6) != last_blk) {
                file_cut(st, slot, Ls);
                ++ncut;
                last_blk = st >> 6;
            }
        }
    };

    int64_t pos = begin;
    // chunks end on 64-block edges (the first one is short): whole blocks take |s| and the quantum from the block statistics,
    // and the next chunk is requested before this one is worked on (the walk is one wave: an unhidden load is ~1 us of nothing)
    int64_t pf_pos = -1;
    double pf_s = 0.0, pf_abs = 0.0;
    int pf_q = 0;
    while (pos < end) {
        const int nvalid = (int)min((int64_t)(64 - (pos & 63)), end - pos);
        // ---- skip whole all-non-positive blocks in the history-free state (as mss_scan_kernel does)
        if (!run_open && peak == MSS_NEG && (pos & 63) == 0) {
            if ((pos & 4095) == 0) {
                const int64_t gj = (pos >> 12) + lane;
                const bool stop2 = gj * 4096 + 4096 > end || (sup_flags[gj] & 1);
                const unsigned long long ms2 = __ballot(stop2);
                const int m2 = ms2 ? __builtin_ctzll(ms2) : 64;
                if (m2 > 0) {
                    const bool in = lane < m2;
                    const double gs = wave_sum(in ? sup_sum[gj] : 0.0);
                    const double ga = wave_sum(in ? sup_abs[gj] : 0.0);
                    const int gq = wave_max(in ? sup_q[gj] : MSS_QNONE);
                    if (mss_certified(cur, ga, gq)) { cur += gs; pos += (int64_t)m2 * 4096; continue; }
                }
            }
            const int64_t b0 = pos >> 6, bend = (end + 63) >> 6;
            const int64_t bj = b0 + lane;
            const bool stop = bj >= bend || (flags[bj] & 1) || (bj * 64 + 64 > end) || (lane > 0 && (bj & 63) == 0);
            const unsigned long long mstop = __ballot(stop);
            const int m = mstop ? __builtin_ctzll(mstop) : 64;
            if (m > 0) {
                const bool in = lane < m;
                const double gs = wave_sum(in ? blk_sum[bj] : 0.0);
                const double ga = wave_sum(in ? blk_abs[bj] : 0.0);
                const int gq = wave_max(in ? blk_q[bj] : MSS_QNONE);
                if (mss_certified(cur, ga, gq)) { cur += gs; pos += (int64_t)m * 64; continue; }
            }
        }
        const bool whole = nvalid == 64;                      // then pos is a block edge
        double s, sabs;
        int q;
        if (pf_pos == pos) { s = pf_s; sabs = pf_abs; q = pf_q; }
        else {
            s = lane < nvalid ? S[pos + lane] : 0.0;
            if (whole) { sabs = blk_abs[pos >> 6]; q = blk_q[pos >> 6]; }
        }
        {
            const int64_t np = pos + nvalid;                  // a block edge
            pf_pos = np;
            pf_s = np + lane < end ? S[np + lane] : 0.0;
            if (np + 64 <= end) { pf_abs = blk_abs[np >> 6]; pf_q = blk_q[np >> 6]; }
        }
        const bool ispos = lane < nvalid && s > 0;
        const unsigned long long mpos = __ballot(ispos);
        if (!whole) {
            sabs = lane_value(scan_add_d(fabs(s)), 63);
            q = lane_value_i(scan_max_i(quantum_exp(s) + 8192), 63) - 8192;            // (scan_max_i's identity is -1: bias the exponents)
        }
        const bool cert = mss_certified(cur, sabs, q);
        if (cert || (q < MSS_QBAD && quantum_exp(cur) < MSS_QBAD)) {
            // the reference's running value after every element, one per lane: certified chunks from exact prefix sums
            // (V_i = base + pre_i); others by the fold itself, lane after lane (mss_fold: roundings as in the reference's loop)
            const double pre = cert ? scan_add_d(s) : 0.0;
            double base = cur;
            double V = cert ? base + pre : mss_fold(s, cur, 0, nvalid, lane);
            int p0 = 0;                                       // lanes < p0 are done (behind an x-drop reset)
            for (;;) {
                const bool act = lane >= p0 && lane < nvalid;
                const double Vb = lane_below_d(V, base);      // value in front of this lane's element (lane 0: the chunk's entry value)
                const bool prevpos = lane > p0 && ((mpos >> (lane - 1)) & 1ull);
                // run starts in [p0, nvalid): a positive lane whose predecessor is not positive (lane p0: unless a run is open)
                const bool isstart = act && ispos && !prevpos && !(lane == p0 && run_open);
                // closing lanes: the first non-positive lane behind a run (lane p0: closes the run carried into the chunk)
                const bool closes = act && !ispos && (lane == p0 ? run_open : prevpos);
                const unsigned long long mstart = __ballot(isstart), mclose = __ballot(closes);
                // most recent start lane at or before each lane, and strictly before it (the run a closing lane closes)
                const int ms = scan_max_i(isstart ? lane : -1);
                const int msp = lane_below_i(ms, -1);
                const bool carried = closes && msp < p0;      // the run opened in an earlier chunk
                const double Ls_g = __shfl(Vb, carried || !closes ? 0 : msp); // value in front of that run's start lane
                const double Ls = carried ? run_L : Ls_g;
                const double R = Vb;                          // running value in front of the closing lane = the run's end value
                // Lf in front of each closing run: min(Lf, start values of the runs closed earlier in the chunk)
                const double mn = scan_min_d(closes ? Ls : INFINITY);
                const double mnb = lane_below_d(mn, INFINITY);
                const double Lf_before = mnb < Lf ? mnb : Lf;
                const bool cut = closes && Ls <= Lf_before;
                // max behind each lane: a cut restarts it at R, any other closed run raises it to R (segmented running max)
                int fl = cut ? 1 : 0;
                double pk = closes ? R : -INFINITY;
                scan_segmax_d(fl, pk);
                const double peak_here = fl ? pk : (pk > peak ? pk : peak);    // after every run closed at or before this lane
                // x-drop (mss.c:89) on the non-positive elements
                const bool trig = act && !ispos && xdrop > 0.0 && peak_here != MSS_NEG && (V + xdrop < peak_here);
                const unsigned long long mt = __ballot(trig);
                const int t = mt ? __builtin_ctzll(mt) : nvalid;          // events at lanes <= t count (t = nvalid: all)
                const unsigned long long upto = t >= 63 ? ~0ull : ((2ull << t) - 1ull);
                // ---- commit the closed runs at lanes <= t
                const unsigned long long mcl = mclose & upto;
                if (mcl) {
                    const int lastc = 63 - __builtin_clzll(mcl);
                    // report one cut: the first whose start lies in another 64-block than the last reported start
                    const int64_t st_pos = carried ? run_st : pos + msp;
                    const unsigned long long mrep = __ballot(cut && lane <= t && (st_pos >> 6) != last_blk);
                    if (mrep) {
                        const int f = __builtin_ctzll(mrep);
                        const int msp_f = lane_value_i(msp, f);
                        const bool car_f = (bool)lane_value_i((int)carried, f);
                        const int64_t st_f = car_f ? run_st : pos + msp_f;
                        // slot of that run
                        const int64_t sl = car_f ? run_slot : slot0 + nruns + __popcll(mstart & ((1ull << msp_f) - 1ull));
                        const double Ls_f = lane_value(Ls, f);
                        file_cut(st_f, sl, Ls_f);
                        ++ncut;
                        last_blk = st_f >> 6;
                    }
                    // state behind the last committed run
                    const double mn_l = lane_value(mn, lastc);
                    Lf = mn_l < Lf ? mn_l : Lf;
                    peak = lane_value(peak_here, lastc);
                    run_open = false;
                }
                if (t < nvalid) {
                    // x-drop reset at lane t: everything is moved out, L restarts at 0 in front of S[t]
                    // (a run that starts at or before t was closed at or before t: starts <= t are counted)
                    nruns += __popcll(mstart & upto);
                    Lf = INFINITY;
                    peak = MSS_NEG;
                    run_open = false;
                    if (cert) {
                        base = t == 0 ? 0.0 : -lane_value(pre, t - 1);
                        V = base + pre;
                    } else {
                        base = 0.0;
                        V = mss_fold(s, 0.0, t, nvalid, lane);
                    }
                    // lane t itself is done: L = 0 + S[t] = V_t
                    p0 = t + 1;
                    if (p0 >= nvalid) { cur = lane_value(V, nvalid - 1); break; }
                    continue;
                }
                // ---- no (further) reset: the chunk is done
                cur = lane_value(V, nvalid - 1);
                const bool lastpos = (mpos >> (nvalid - 1)) & 1ull;
                if (lastpos) {
                    const int msl = lane_value_i(ms, nvalid - 1);         // start lane of the run that stays open (-1: carried on)
                    if (msl >= p0) {
                        run_st = pos + msl;
                        run_L = lane_value(Vb, msl);
                        run_slot = slot0 + nruns + __popcll(mstart & ((1ull << msl) - 1ull));
                    }
                    run_open = true;
                }
                nruns += __popcll(mstart);
                break;
            }
        } else {
            // ---- certificate failed: the reference loop, element by element
            for (int i = 0; i < nvalid; ++i) {
                const double v = lane_value(s, i);
                if (v > 0) {
                    if (!run_open) { run_open = true; run_st = pos + i; run_L = cur; run_slot = slot0 + nruns; ++nruns; }
                    cur = cur + v;
                } else {
                    if (run_open) { close_scalar(run_L, cur, run_st, run_slot); run_open = false; }
                    if (xdrop > 0.0 && cur + v + xdrop < peak) { cur = 0.0; peak = MSS_NEG; Lf = INFINITY; }
                    cur += v;
                }
            }
        }
        pos += nvalid;
    }
    if (run_open && k == nunits - 1) { close_scalar(run_L, cur, run_st, run_slot); run_open = false; }   // end of the sequence
    if (lane == 0) {
        // a stretch boundary was placed where a reset is forced: behind it the reference's state must be empty
        if (k != nunits - 1 && uforced[k + 1] && (Lf != INFINITY || peak != MSS_NEG || run_open))
            atomicOr((unsigned long long *)&grand[2], 1ull);
        mss_light_state x;
        x.L = cur; x.peak = peak; x.Lf = Lf; x.run_L = run_open ? run_L : 0.0;
        x.run_st = run_open ? run_st : 0; x.run_slot = run_open ? run_slot : 0; x.last_blk = last_blk; x.open = run_open ? 1 : 0;
        // changed against the previous pass's end state of this unit?
        bool same = pass > 0;
        if (pass > 0) {
            const long long *o = (const long long *)&st_prev[k], *w = (const long long *)&x;
            for (int i = 0; i < 8; ++i) same = same && o[i] == w[i];
        }
        if (!same) atomicOr((unsigned long long *)&grand[1], 1ull);
        st_cur[k] = x;
        cut_cnt[k] = (uint64_t)ncut;
    }
}

// the pieces in order: piece 0 of a forced unit sits at cut_off[k], the cuts filed by 64-block behind it
__global__ void __launch_bounds__(256) mss_pieces_kernel(const int64_t *__restrict__ ustart, const int64_t *__restrict__ urun,
                                                         const uint8_t *__restrict__ uforced, int64_t nunits,
                                                         const mss_light_state *__restrict__ st, const uint64_t *__restrict__ cut_off,
                                                         const int64_t *__restrict__ cut_st, const int64_t *__restrict__ cut_run,
                                                         const double *__restrict__ cut_L, const int2 *__restrict__ cut_kj, int64_t nblk,
                                                         int64_t cap, int64_t *__restrict__ ustart2, int64_t *__restrict__ urun2,
                                                         double *__restrict__ entry2)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nunits && uforced[i]) {
        const uint64_t at = cut_off[i];
        if ((int64_t)at < cap) { ustart2[at] = ustart[i]; urun2[at] = urun[i]; entry2[at] = i == 0 ? 0.0 : st[i - 1].L; }
    }
    if (i < nblk && cut_st[i] >= 0) {
        const int2 kj = cut_kj[i];
        const uint64_t at = cut_off[kj.x] + (uint64_t)kj.y;
        if ((int64_t)at < cap) { ustart2[at] = cut_st[i]; urun2[at] = cut_run[i]; entry2[at] = cut_L[i]; }
    }
}

// ---- the pieces cut again at the speculative light edges inside them (mss_subent) ------------------------------------------------
__device__ __forceinline__ int64_t mss_lower_bound(const int64_t *a, int64_t n, int64_t x)       // first i with a[i] >= x
{
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}
// valid[j] = light unit j starts at a speculative edge that is not a piece start (then scanned: compact index of the edge)
__global__ void __launch_bounds__(256) mss_specvalid_kernel(const int64_t *__restrict__ lstart, const uint8_t *__restrict__ lforced, int64_t nl,
                                                            const int64_t *__restrict__ pstart, int64_t np, uint64_t *__restrict__ valid)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nl) return;
    bool v = j >= 1 && j < nl && !lforced[j];
    if (v) {
        const int64_t at = mss_lower_bound(pstart, np, lstart[j]);
        v = !(at < np && pstart[at] == lstart[j]);
    }
    valid[j] = v ? 1 : 0;
}
// pieces and valid edges merged by position; subs[nsub] = the end sentinel
__global__ void __launch_bounds__(256) mss_subtable_kernel(const int64_t *__restrict__ lstart, const int64_t *__restrict__ lrun,
                                                           const uint8_t *__restrict__ lforced, int64_t nl, const mss_light_state *__restrict__ lst,
                                                           const uint64_t *__restrict__ vc, const uint64_t *__restrict__ nv_p,
                                                           const int64_t *__restrict__ pstart, const int64_t *__restrict__ prun,
                                                           const double *__restrict__ pentry, int64_t np, int64_t n, mss_subent *__restrict__ subs)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nv = (int64_t)*nv_p;
    if (i < np) {
        const int64_t jx = mss_lower_bound(lstart, nl + 1, pstart[i]);            // edges in front of the piece: units j < jx
        mss_subent e = {};
        e.start = pstart[i]; e.run = prun[i]; e.L = pentry[i]; e.peak = MSS_NEG; e.first = 1; e.vci = -1; e.nst = 0;
        subs[i + (int64_t)vc[jx]] = e;
    }
    if (i >= 1 && i < nl && !lforced[i]) {
        const bool valid = vc[i + 1] != vc[i];
        if (valid) {
            const int64_t pb = mss_lower_bound(pstart, np, lstart[i] + 1);        // pieces that start at or in front of the edge
            const mss_light_state st = lst[i - 1];
            mss_subent e = {};
            e.start = lstart[i]; e.run = lrun[i]; e.L = st.L; e.peak = st.peak; e.run_L = st.run_L; e.run_st = st.run_st;
            e.open = st.open != 0; e.first = 0; e.vci = (int32_t)vc[i]; e.nst = 0; e.Lf = st.Lf;
            subs[pb + (int64_t)vc[i]] = e;
        }
    }
    if (i == 0) {
        mss_subent e = {};
        e.start = n; e.first = 1; e.L = pentry[np]; e.vci = -1;
        subs[np + nv] = e;
    }
}

__global__ void __launch_bounds__(256) mss_compact_kernel(const int32_t *__restrict__ segs, const int64_t *__restrict__ urun,
                                                          const uint64_t *__restrict__ segoff, const uint64_t *__restrict__ segcnt_total,
                                                          int64_t nunits, int32_t *__restrict__ out, const mss_subent *__restrict__ subs)
{
    // one thread per stretch copies its kept segments to their global slots (order preserved)
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nunits) return;
    const int64_t from = subs ? subs[k].run : urun[k];
    const uint64_t to = segoff[k];
    const uint64_t cnt = (k + 1 < nunits ? segoff[k + 1] : *segcnt_total) - to;
    for (uint64_t j = 0; j < cnt; ++j) {
        out[2 * (to + j)] = segs[2 * (from + j)];
        out[2 * (to + j) + 1] = segs[2 * (from + j) + 1];
    }
}

// A9  deepgrp/_mss/pymss.pyx:57-77: one wave per kept segment: majority label over 1..C-1 (first
// maximum wins, all-zero segment -> 1), zeros inside the segment take it.
#define MSS_VOTE_LONG 65536       // a longer segment is voted on by the whole grid (mss_vote_long_kernel): one wave walked a 10 M-base
#define MSS_VOTE_CHUNK 16384      // segment for 73 ms
__global__ void __launch_bounds__(256) mss_vote_kernel(const int32_t *__restrict__ segs, const uint64_t *__restrict__ nseg_p,
                                                       const int8_t *__restrict__ cls, int nof_labels,
                                                       int8_t *__restrict__ out, unsigned long long *__restrict__ nlong,
                                                       int32_t *__restrict__ longlist, int64_t longcap)
{
    const int lane = threadIdx.x & 63;
    const int64_t nseg = (int64_t)*nseg_p;
    for (int64_t sidx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); sidx < nseg; sidx += (int64_t)gridDim.x * 4) {
        const int64_t st = segs[2 * sidx], en = segs[2 * sidx + 1];
        if (nlong && en - st > MSS_VOTE_LONG) {
            // filed for the whole grid: [0, longcap) segment indices, behind them 16 counters per filed segment
            unsigned long long q = 0;
            if (lane == 0) q = atomicAdd(nlong, 1ull);
            q = __shfl(q, 0);
            if ((int64_t)q < longcap) {
                if (lane == 0) longlist[q] = (int32_t)sidx;
                if (lane < DGRP_MAXC) longlist[longcap + DGRP_MAXC * q + lane] = 0;
                continue;
            }
        }
        int best = 1, bv = 0;
        for (int c0 = 0; c0 < nof_labels; c0 += 16) {          // (16 classes per pass over the segment: one pass for the usual 5)
            int cnt[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) cnt[c] = 0;
            for (int64_t j = st + lane; j < en; j += 64) {
                const int l = cls[j] - c0;
#pragma unroll
                for (int c = 0; c < 16; ++c) cnt[c] += (l == c);
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                int v = cnt[c];
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                if (c0 + c == 1) bv = v;
                else if (c0 + c > 1 && c0 + c < nof_labels && bv < v) { best = c0 + c; bv = v; }
            }
        }
        for (int64_t j = st + lane; j < en; j += 64)
            if (cls[j] == 0) out[j] = (int8_t)best;
    }
}

// the long segments: PHASE 0 counts the classes chunk by chunk over the whole grid, PHASE 1 picks the label (pymss.pyx:57-77: first
// maximum over 1..C-1, all-zero -> 1) and fills the zeros
template <int PHASE>
__global__ void __launch_bounds__(256) mss_vote_long_kernel(const int32_t *__restrict__ segs, const int8_t *__restrict__ cls, int nof_labels,
                                                            int8_t *__restrict__ out, const unsigned long long *__restrict__ nlong,
                                                            int32_t *__restrict__ longlist, int64_t longcap)
{
    const int64_t nq = (int64_t)min((unsigned long long)longcap, *nlong);
    for (int64_t q = 0; q < nq; ++q) {
        const int64_t sidx = longlist[q];
        const int64_t st = segs[2 * sidx], en = segs[2 * sidx + 1];
        int32_t *hist = longlist + longcap + DGRP_MAXC * q;
        int best = 1;
        if (PHASE == 1) {
            int bv = hist[1];
            for (int c = 2; c < nof_labels; ++c)
                if (bv < hist[c]) { best = c; bv = hist[c]; }
        }
        for (int64_t c0 = st + (int64_t)blockIdx.x * MSS_VOTE_CHUNK; c0 < en; c0 += (int64_t)gridDim.x * MSS_VOTE_CHUNK) {
            const int64_t c1 = min(c0 + (int64_t)MSS_VOTE_CHUNK, en);
            if (PHASE == 0) {
                __shared__ int sh[DGRP_MAXC];
                if (threadIdx.x < DGRP_MAXC) sh[threadIdx.x] = 0;
                __syncthreads();
                for (int b0 = 0; b0 < nof_labels; b0 += 16) {
                    int cnt[16];
#pragma unroll
                    for (int c = 0; c < 16; ++c) cnt[c] = 0;
                    for (int64_t j = c0 + threadIdx.x; j < c1; j += 256) {
                        const int l = cls[j] - b0;
#pragma unroll
                        for (int c = 0; c < 16; ++c) cnt[c] += (l == c);
                    }
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        int v = cnt[c];
                        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                        if ((threadIdx.x & 63) == 0 && v && b0 + c >= 1 && b0 + c < DGRP_MAXC) atomicAdd(&sh[b0 + c], v);
                    }
                }
                __syncthreads();
                if (threadIdx.x >= 1 && threadIdx.x < DGRP_MAXC && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
                __syncthreads();
            } else {
                for (int64_t j = c0 + threadIdx.x; j < c1; j += 256)
                    if (cls[j] == 0) out[j] = (int8_t)best;
            }
        }
    }
}

// A9 over the kept segments (l.segs_out, count in grand[3]); l.segs is free by now: the list of the long ones and their counters
static int mss_vote_all(const mss_layout &l, const int8_t *d_cls, int nof_labels, int8_t *d_labels_out, int64_t n, hipStream_t stream)
{
    const int64_t longcap = n / MSS_VOTE_LONG + 1;                 // ((1 + DGRP_MAXC) longcap ints <= the 2 (n / 2 + 2) of l.segs)
    unsigned long long *nlong = (unsigned long long *)l.grand;     // grand[0]: the scans are done with it
    DGRP_HIP(hipMemsetAsync(nlong, 0, 8, stream));
    hipLaunchKernelGGL(mss_vote_kernel, dim3(1024), dim3(256), 0, stream, l.segs_out, l.grand + 3, d_cls, nof_labels, d_labels_out, nlong,
                       l.segs, longcap);
    DGRP_LAUNCH_CHECK();
    hipLaunchKernelGGL(mss_vote_long_kernel<0>, dim3(1024), dim3(256), 0, stream, l.segs_out, d_cls, nof_labels, d_labels_out, nlong, l.segs,
                       longcap);
    DGRP_LAUNCH_CHECK();
    hipLaunchKernelGGL(mss_vote_long_kernel<1>, dim3(1024), dim3(256), 0, stream, l.segs_out, d_cls, nof_labels, d_labels_out, nlong, l.segs,
                       longcap);
    DGRP_LAUNCH_CHECK();
    return DGRP_OK;
}

DGRP_EXPORT int dgrp_mss_labels(const double *d_scores, const int8_t *d_cls, int64_t n, int nof_labels,
                                int min_mss_len, int xdrop_len, int8_t *d_labels_out, int64_t *d_nseg,
                                void *d_work, int64_t work_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    DGRP_REQUIRE(n >= 0 && n < (1ll << 31), "dgrp_mss_labels: n=%lld out of range (the reference indexes with int)", (long long)n);
    DGRP_REQUIRE(nof_labels >= 2 && nof_labels <= DGRP_MAXC, "dgrp_mss_labels: nof_labels must be in 2..64");
    if (n == 0) {
        if (d_nseg) DGRP_HIP(hipMemsetAsync(d_nseg, 0, sizeof(int64_t), stream));
        return DGRP_OK;
    }
    DGRP_REQUIRE(d_scores && d_cls && d_labels_out && d_work, "dgrp_mss_labels: NULL pointer");
    mss_layout l = mss_carve(d_work, n);
    if (work_bytes < l.bytes) {
        dgrp_set_error("dgrp_mss_labels: workspace %lld < %lld bytes", (long long)work_bytes, (long long)l.bytes);
        return DGRP_ENOMEM;
    }
    // pymss.pyx:46-53 and mss.c:35 (int truncation of the threshold)
    const double s0 = log(0.99 / (1.0 - 0.99));
    const double xdrop = xdrop_len > 0 ? s0 * xdrop_len * 10.0 : -1;
    const int min_sc = (int)(s0 * min_mss_len);
    const double thr = xdrop > 0.0 ? xdrop + 1.0 : -1.0;

    DGRP_HIP(hipMemsetAsync(l.blk, 0, (l.nblk + 1) * 8, stream));
    hipLaunchKernelGGL(mss_blockstat_kernel, dim3((unsigned)((l.nblk + 3) / 4)), dim3(256), 0, stream, d_scores, n, l.blk,
                       l.blk_sum, l.blk_abs, l.blk_q, l.flags);
    DGRP_LAUNCH_CHECK();
    hipLaunchKernelGGL(mss_superstat_kernel, dim3((unsigned)((l.nsup + 3) / 4)), dim3(256), 0, stream, l.blk_sum, l.blk_abs,
                       l.blk_q, l.flags, l.nblk, l.sup_sum, l.sup_abs, l.sup_q, l.sup_flags);
    DGRP_LAUNCH_CHECK();

    bool single = false;
    for (int attempt = 0; attempt < 2; ++attempt) {
        int64_t nunits = 1;
        DGRP_HIP(hipMemsetAsync(l.grand, 0, 64, stream));
        if (!single && xdrop > 0.0) {
            if (attempt == 0) {
                hipLaunchKernelGGL(mss_boundary_kernel, dim3((unsigned)((l.nblk + 255) / 256)), dim3(256), 0, stream,
                                   l.flags, l.blk_sum, l.nblk, n, thr, l.blk);
                DGRP_LAUNCH_CHECK();
                int rc = device_exclusive_scan(l.blk, l.blk, l.nblk, l.tiles, l.grand, stream);
                if (rc) return rc;
            }
            hipLaunchKernelGGL(mss_units_kernel, dim3((unsigned)((l.nblk + 255) / 256)), dim3(256), 0, stream, l.blk, l.flags,
                               l.blk_sum, l.nblk, n, thr, l.ustart, l.urun, l.grand);
            DGRP_LAUNCH_CHECK();
            uint64_t g = 0;
            DGRP_HIP(hipMemcpyAsync(&g, l.grand, 8, hipMemcpyDeviceToHost, stream));
            DGRP_HIP(hipStreamSynchronize(stream));
            nunits = (int64_t)(g >> 32) + 1;
        } else {
            // one stretch: the plain sequential scan (still chunked and certified)
            int64_t h[2] = { 0, n };
            DGRP_HIP(hipMemcpyAsync(l.ustart, h, 16, hipMemcpyHostToDevice, stream));
            int64_t r[2] = { 0, 0 };
            DGRP_HIP(hipMemcpyAsync(l.urun, r, 16, hipMemcpyHostToDevice, stream));
            DGRP_HIP(hipStreamSynchronize(stream));
        }
        DGRP_REQUIRE(nunits < (1ll << 31), "dgrp_mss_labels: too many stretches");
        DGRP_HIP(hipMemsetAsync(l.exitL[0], 0, nunits * 8, stream));
        DGRP_HIP(hipMemsetAsync(l.exitL[1], 0, nunits * 8, stream));
        bool failed = false;
        // ---- second level: flush cuts (mss_light_kernel).  The light walk finds every stretch's entry L by the same fixed point
        // the scan used to run with the full stack machinery, and cuts the stretches where the reference's stack is flushed;
        // then ONE pass of the stack scan over the pieces, each from its known entry L.  Any inconsistency (a piece whose exit L
        // differs from the next piece's entry, bit for bit) sends the record through the uncut scan below.
        bool done = false;
        if (!single && !getenv("DGRP_MSS_NO_CUTS")) {
            // light units: the stretches, cut further every `sub` blocks where the input can reset by x-drop
            int64_t nl = 1;
            if (xdrop > 0.0) {
                const char *e = getenv("DGRP_MSS_SUB");
                const int sub = e && atoi(e) > 0 ? atoi(e) : MSS_LIGHT_SUB;
                hipLaunchKernelGGL(mss_lightflag_kernel, dim3((unsigned)((l.nblk + 255) / 256)), dim3(256), 0, stream, l.flags, l.blk_sum,
                                   l.nblk, n, thr, sub, 1, l.lblk);
                DGRP_LAUNCH_CHECK();
                int rc1 = device_exclusive_scan(l.lblk, l.lblk, l.nblk, l.tiles, l.grand + 6, stream);
                if (rc1) return rc1;
                hipLaunchKernelGGL(mss_lightunits_kernel, dim3((unsigned)((l.nblk + 255) / 256)), dim3(256), 0, stream, l.lblk, l.grand + 6,
                                   l.flags, l.blk_sum, l.blk, l.grand, l.nblk, n, thr, sub, 1, l.lstart, l.lrun, l.lforced);
                DGRP_LAUNCH_CHECK();
                uint64_t g6 = 0;
                DGRP_HIP(hipMemcpyAsync(&g6, l.grand + 6, 8, hipMemcpyDeviceToHost, stream));
                DGRP_HIP(hipStreamSynchronize(stream));
                nl = (int64_t)g6 + 1;
            } else {
                const int64_t h[2] = { 0, n }, r[2] = { 0, 0 };
                const uint8_t f[2] = { 1, 1 };
                DGRP_HIP(hipMemcpyAsync(l.lstart, h, 16, hipMemcpyHostToDevice, stream));
                DGRP_HIP(hipMemcpyAsync(l.lrun, r, 16, hipMemcpyHostToDevice, stream));
                DGRP_HIP(hipMemcpyAsync(l.lforced, f, 2, hipMemcpyHostToDevice, stream));
                DGRP_HIP(hipStreamSynchronize(stream));
            }
            int pass = 0;
            for (;; ++pass) {
                DGRP_HIP(hipMemsetAsync(l.grand + 1, 0, 8, stream));
                DGRP_HIP(hipMemsetAsync(l.cut_st, 0xff, l.nblk * 8, stream));
                hipLaunchKernelGGL(mss_light_kernel, dim3((unsigned)nl), dim3(64), 0, stream, d_scores, l.lstart, l.lrun, l.lforced, nl,
                                   (const mss_light_state *)l.lstate[(pass + 1) & 1], (mss_light_state *)l.lstate[pass & 1], xdrop, l.grand,
                                   pass, l.blk_sum, l.blk_abs, l.blk_q, l.flags, l.sup_sum, l.sup_abs, l.sup_q, l.sup_flags, l.cutcnt,
                                   l.cut_st, l.cut_run, l.cut_L, l.cut_kj);
                DGRP_LAUNCH_CHECK();
                uint64_t g[3] = { 0, 0, 0 };
                DGRP_HIP(hipMemcpyAsync(g, l.grand, 24, hipMemcpyDeviceToHost, stream));
                DGRP_HIP(hipStreamSynchronize(stream));
                if (g[2]) { failed = true; break; }
                if (nl == 1 || (pass > 0 && g[1] == 0)) break;
                if (pass > nl + 2) { failed = true; break; }
            }
            if (getenv("DGRP_MSS_TRACE")) fprintf(stderr, "dgrp_mss_labels: n=%lld stretches=%lld light units=%lld passes=%d%s\n", (long long)n,
                                                  (long long)nunits, (long long)nl, pass + 1, failed ? " FAILED" : "");
            if (failed) {
                if (single) { dgrp_set_error("dgrp_mss_labels: sequential scan reported an inconsistent state"); return DGRP_EHIP; }
                single = true;
                continue;
            }
            // pieces per unit -> offsets, then every piece to its slot (the last walk's table is the converged one)
            int rc2 = device_exclusive_scan(l.cutcnt, l.cutcnt, nl, l.tiles, l.grand + 5, stream);
            if (rc2) return rc2;
            const int64_t piece_cap = 2 * l.nblk + 2;
            {
                const int64_t threads = nl > l.nblk ? nl : l.nblk;
                hipLaunchKernelGGL(mss_pieces_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, l.lstart, l.lrun, l.lforced,
                                   nl, (const mss_light_state *)l.lstate[pass & 1], l.cutcnt, l.cut_st, l.cut_run, l.cut_L, l.cut_kj, l.nblk,
                                   piece_cap, l.ustart2, l.urun2, l.entry2);
                DGRP_LAUNCH_CHECK();
            }
            uint64_t np = 0;
            DGRP_HIP(hipMemcpyAsync(&np, l.grand + 5, 8, hipMemcpyDeviceToHost, stream));
            DGRP_HIP(hipStreamSynchronize(stream));
            const int64_t npieces = (int64_t)np;
            if (npieces >= 1 && npieces <= piece_cap) {
                DGRP_HIP(hipMemcpyAsync(l.ustart2 + npieces, &n, 8, hipMemcpyHostToDevice, stream));
                DGRP_HIP(hipMemcpyAsync(l.entry2 + npieces, (const mss_light_state *)l.lstate[pass & 1] + (nl - 1), 8, hipMemcpyDeviceToDevice,
                                        stream));   // .L is the first field
                // ---- third level: a piece that spans speculative light edges is scanned in parts, all at once, each on a local stack
                // from the edge's converged state, and stitched (mss_subent, mss_stitch_kernel).  Anything the stitch does not take -- a
                // local stack deeper than the LDS part, an impossible event -- sends the pieces through the scan whole, as before.
                bool sub_done = false;
                mss_subent *subs = (mss_subent *)l.subs;
                if (nl > 1 && xdrop > 0.0 && !getenv("DGRP_MSS_NO_SUB")) {
                    const char *e = getenv("DGRP_MSS_SUB");
                    const int sub = e && atoi(e) > 0 ? atoi(e) : MSS_LIGHT_SUB;
                    if (sub >= MSS_SUB_MIN) {
                        hipLaunchKernelGGL(mss_specvalid_kernel, dim3((unsigned)((nl + 1 + 255) / 256)), dim3(256), 0, stream, l.lstart, l.lforced,
                                           nl, l.ustart2, npieces, l.lblk);
                        DGRP_LAUNCH_CHECK();
                        int rcv = device_exclusive_scan(l.lblk, l.lblk, nl + 1, l.tiles, l.grand + 7, stream);
                        if (rcv) return rcv;
                        uint64_t nv = 0;
                        DGRP_HIP(hipMemcpyAsync(&nv, l.grand + 7, 8, hipMemcpyDeviceToHost, stream));
                        DGRP_HIP(hipStreamSynchronize(stream));
                        if (nv > 0 && 2 * (int64_t)nv + 2 <= l.dump_cap && npieces + (int64_t)nv <= l.nblk) {   // (segcnt / exitL hold nblk + 1 units)
                            const int64_t nsub = npieces + (int64_t)nv;
                            const int64_t threads = (npieces > nl ? npieces : nl) + 1;
                            hipLaunchKernelGGL(mss_subtable_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, l.lstart, l.lrun,
                                               l.lforced, nl, (const mss_light_state *)l.lstate[pass & 1], l.lblk, l.grand + 7, l.ustart2,
                                               l.urun2, l.entry2, npieces, n, subs);
                            DGRP_LAUNCH_CHECK();
                            DGRP_HIP(hipMemsetAsync(l.grand + 1, 0, 16, stream));
                            DGRP_HIP(hipMemsetAsync(l.grand + 4, 0, 8, stream));
                            hipLaunchKernelGGL(mss_scan_kernel, dim3((unsigned)nsub), dim3(64), 0, stream, d_scores, l.ustart2, l.urun2, nsub,
                                               l.entry2 + 1, l.exitL[(pass + 1) & 1], l.stack, l.segs, l.segcnt, min_sc, xdrop, l.grand, 1,
                                               l.blk_sum, l.blk_abs, l.blk_q, l.flags, l.sup_sum, l.sup_abs, l.sup_q, l.sup_flags, 1, 0, 1, subs,
                                               l.dump);
                            DGRP_LAUNCH_CHECK();
                            hipLaunchKernelGGL(mss_stitch_kernel, dim3((unsigned)nsub), dim3(64), 0, stream, subs, nsub, l.stack, l.dump, l.segs,
                                               l.segcnt, min_sc, l.grand);
                            DGRP_LAUNCH_CHECK();
                            uint64_t g[5] = { 0, 0, 0, 0, 0 };
                            DGRP_HIP(hipMemcpyAsync(g, l.grand, 40, hipMemcpyDeviceToHost, stream));
                            DGRP_HIP(hipStreamSynchronize(stream));
                            if (getenv("DGRP_MSS_TRACE"))
                                fprintf(stderr, "dgrp_mss_labels: %lld pieces in %lld parts (%llu speculative edges inside pieces)%s (exit %llu, error %llu, why %llu)\n",
                                        (long long)npieces, (long long)nsub, (unsigned long long)nv, g[1] || g[2] ? ": not stitched" : "",
                                        (unsigned long long)g[1], (unsigned long long)g[2], (unsigned long long)g[4]);
                            if (!g[1] && !g[2]) {
                                int rc3 = device_exclusive_scan(l.segcnt, l.segcnt, nsub, l.tiles, l.grand + 3, stream);
                                if (rc3) return rc3;
                                hipLaunchKernelGGL(mss_compact_kernel, dim3((unsigned)((nsub + 255) / 256)), dim3(256), 0, stream, l.segs,
                                                   l.urun2, l.segcnt, l.grand + 3, nsub, l.segs_out, (const mss_subent *)subs);
                                DGRP_LAUNCH_CHECK();
                                done = sub_done = true;
                            }
                        }
                    }
                }
                if (!sub_done) {
                DGRP_HIP(hipMemsetAsync(l.grand + 1, 0, 16, stream));
                hipLaunchKernelGGL(mss_scan_kernel, dim3((unsigned)npieces), dim3(64), 0, stream, d_scores, l.ustart2, l.urun2, npieces,
                                   l.entry2 + 1, l.exitL[(pass + 1) & 1], l.stack, l.segs, l.segcnt, min_sc, xdrop, l.grand, 1, l.blk_sum,
                                   l.blk_abs, l.blk_q, l.flags, l.sup_sum, l.sup_abs, l.sup_q, l.sup_flags, 1, 0, 1, (mss_subent *)nullptr,
                                   (mss_cand *)nullptr);
                DGRP_LAUNCH_CHECK();
                uint64_t g[3] = { 0, 0, 0 };
                DGRP_HIP(hipMemcpyAsync(g, l.grand, 24, hipMemcpyDeviceToHost, stream));
                DGRP_HIP(hipStreamSynchronize(stream));
                if (!g[1] && !g[2]) {
                    int rc3 = device_exclusive_scan(l.segcnt, l.segcnt, npieces, l.tiles, l.grand + 3, stream);
                    if (rc3) return rc3;
                    hipLaunchKernelGGL(mss_compact_kernel, dim3((unsigned)((npieces + 255) / 256)), dim3(256), 0, stream, l.segs, l.urun2,
                                       l.segcnt, l.grand + 3, npieces, l.segs_out, (const mss_subent *)nullptr);
                    DGRP_LAUNCH_CHECK();
                    done = true;
                }
                }
            }
            if (done) break;
            // inconsistent: fall through to the uncut scan (exits start from zero again)
            DGRP_HIP(hipMemsetAsync(l.exitL[0], 0, nunits * 8, stream));
            DGRP_HIP(hipMemsetAsync(l.exitL[1], 0, nunits * 8, stream));
            DGRP_HIP(hipMemsetAsync(l.grand + 1, 0, 16, stream));
        }
        for (int pass = 0;; ++pass) {
            // pass p reads the exits of pass p-1 from exitL[(p+1)&1] and writes exitL[p&1]
            DGRP_HIP(hipMemsetAsync(l.grand + 1, 0, 8, stream));
            hipLaunchKernelGGL(mss_scan_kernel, dim3((unsigned)nunits), dim3(64), 0, stream, d_scores, l.ustart, l.urun,
                               nunits, l.exitL[(pass + 1) & 1], l.exitL[pass & 1], l.stack, l.segs, l.segcnt, min_sc, xdrop,
                               l.grand, pass, l.blk_sum, l.blk_abs, l.blk_q, l.flags, l.sup_sum, l.sup_abs, l.sup_q,
                               l.sup_flags, 1, 0, 0, (mss_subent *)nullptr, (mss_cand *)nullptr);
            DGRP_LAUNCH_CHECK();
            uint64_t g[3] = { 0, 0, 0 };
            DGRP_HIP(hipMemcpyAsync(g, l.grand, 24, hipMemcpyDeviceToHost, stream));
            DGRP_HIP(hipStreamSynchronize(stream));
            if (g[2]) { failed = true; break; }
            if (nunits == 1 || (pass > 0 && g[1] == 0)) break;
            if (pass > nunits + 2) { failed = true; break; }
        }
        if (failed) {
            if (single) { dgrp_set_error("dgrp_mss_labels: sequential scan reported an inconsistent state"); return DGRP_EHIP; }
            single = true;
            continue;
        }
        // kept segments -> one ordered list
        int rc = device_exclusive_scan(l.segcnt, l.segcnt, nunits, l.tiles, l.grand + 3, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(mss_compact_kernel, dim3((unsigned)((nunits + 255) / 256)), dim3(256), 0, stream, l.segs, l.urun,
                           l.segcnt, l.grand + 3, nunits, l.segs_out, (const mss_subent *)nullptr);
        DGRP_LAUNCH_CHECK();
        break;
    }
    DGRP_HIP(hipMemcpyAsync(d_labels_out, d_cls, n, hipMemcpyDeviceToDevice, stream));
    {
        const int rcv = mss_vote_all(l, d_cls, nof_labels, d_labels_out, n, stream);
        if (rcv) return rcv;
    }
    if (d_nseg) DGRP_HIP(hipMemcpyAsync(d_nseg, l.grand + 3, 8, hipMemcpyDeviceToDevice, stream));
    return DGRP_OK;
}

DGRP_EXPORT int dgrp_mss_segments_host(const void *d_work, int64_t work_bytes, int32_t *h_st_en, int64_t cap, int64_t *n_seg)
{
    DGRP_REQUIRE(d_work && n_seg, "dgrp_mss_segments_host: NULL pointer");
    // the layout depends only on n, which the caller encodes through work_bytes = dgrp_mss_workspace_bytes(n):
    // recover n by bisection (the carve is monotone in n)
    int64_t lo = 0, hi = (1ll << 31) - 1;
    while (lo < hi) {
        int64_t mid = lo + (hi - lo + 1) / 2;
        if (mss_carve(nullptr, mid).bytes <= work_bytes) lo = mid; else hi = mid - 1;
    }
    mss_layout l = mss_carve((void *)d_work, lo);
    uint64_t cnt = 0;
    DGRP_HIP(hipMemcpy(&cnt, l.grand + 3, 8, hipMemcpyDeviceToHost));
    *n_seg = (int64_t)cnt;
    const int64_t take = (int64_t)cnt < cap ? (int64_t)cnt : cap;
    if (take > 0 && h_st_en) DGRP_HIP(hipMemcpy(h_st_en, l.segs_out, take * 8, hipMemcpyDeviceToHost));
    return DGRP_OK;
}

// ---- A9+A10 for MANY records side by side: record r occupies [h_start[r], h_start[r+1]) of the score / class arrays,
// every start a multiple of 64; positions between a record's last base and the next start must hold score 0 and
// class 0 (a non-positive score there closes an open run, may fire an x-drop flush and adds 0 to L: the kept
// segments are those of the record alone, mss.c:96).  One wave scans one record with the same exact arithmetic as
// dgrp_mss_labels' single-stretch mode (certified 64-chunks, element-by-element otherwise); one launch of each
// kernel for all records, one synchronisation.  Meant for records of up to a few 100 kbp.
DGRP_EXPORT int64_t dgrp_mss_batch_workspace_bytes(int64_t total_n, int64_t nrec)
{
    if (total_n < 0 || nrec < 0) return 0;
    return mss_carve(nullptr, total_n + 4 * nrec + 64).bytes;
}

DGRP_EXPORT int dgrp_mss_labels_batch(const double *d_scores, const int8_t *d_cls, int64_t total_n, int64_t nrec,
                                      const int64_t *h_start, int nof_labels, int min_mss_len, int xdrop_len,
                                      int8_t *d_labels_out, void *d_work, int64_t work_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    DGRP_REQUIRE(total_n >= 0 && total_n < (1ll << 31) && nrec >= 0 && (nrec == 0 || h_start), "dgrp_mss_labels_batch: bad arguments");
    DGRP_REQUIRE(nof_labels >= 2 && nof_labels <= DGRP_MAXC, "dgrp_mss_labels_batch: nof_labels must be in 2..64");
    if (nrec == 0 || total_n == 0) return DGRP_OK;
    DGRP_REQUIRE(d_scores && d_cls && d_labels_out && d_work, "dgrp_mss_labels_batch: NULL pointer");
    DGRP_REQUIRE(h_start[0] == 0 && h_start[nrec] == total_n, "dgrp_mss_labels_batch: starts must run from 0 to total_n");
    for (int64_t r = 0; r < nrec; ++r)
        DGRP_REQUIRE((h_start[r] & 63) == 0 && h_start[r + 1] > h_start[r], "dgrp_mss_labels_batch: record %lld: starts must be increasing multiples of 64", (long long)r);
    mss_layout l = mss_carve(d_work, total_n + 4 * nrec + 64);
    if (work_bytes < l.bytes) {
        dgrp_set_error("dgrp_mss_labels_batch: workspace %lld < %lld bytes", (long long)work_bytes, (long long)l.bytes);
        return DGRP_ENOMEM;
    }
    l.nblk = (total_n + 63) / 64;
    l.nsup = (l.nblk + 63) / 64;
    const double s0 = log(0.99 / (1.0 - 0.99));
    const double xdrop = xdrop_len > 0 ? s0 * xdrop_len * 10.0 : -1;
    const int min_sc = (int)(s0 * min_mss_len);
    hipLaunchKernelGGL(mss_blockstat_kernel, dim3((unsigned)((l.nblk + 3) / 4)), dim3(256), 0, stream, d_scores, total_n, l.blk,
                       l.blk_sum, l.blk_abs, l.blk_q, l.flags);
    hipLaunchKernelGGL(mss_superstat_kernel, dim3((unsigned)((l.nsup + 3) / 4)), dim3(256), 0, stream, l.blk_sum, l.blk_abs,
                       l.blk_q, l.flags, l.nblk, l.sup_sum, l.sup_abs, l.sup_q, l.sup_flags);
    DGRP_LAUNCH_CHECK();
    std::vector<int64_t> urun((size_t)nrec + 1);
    for (int64_t r = 0; r <= nrec; ++r) urun[(size_t)r] = h_start[r] / 2 + 2 * r;      // a record of n scores has at most n/2 + 1 runs
    DGRP_HIP(hipMemcpyAsync(l.ustart, h_start, (size_t)(nrec + 1) * 8, hipMemcpyHostToDevice, stream));
    DGRP_HIP(hipMemcpyAsync(l.urun, urun.data(), (size_t)(nrec + 1) * 8, hipMemcpyHostToDevice, stream));
    DGRP_HIP(hipMemsetAsync(l.grand, 0, 64, stream));
    hipLaunchKernelGGL(mss_scan_kernel, dim3((unsigned)nrec), dim3(64), 0, stream, d_scores, l.ustart, l.urun, nrec, l.exitL[1],
                       l.exitL[0], l.stack, l.segs, l.segcnt, min_sc, xdrop, l.grand, 0, l.blk_sum, l.blk_abs, l.blk_q, l.flags,
                       l.sup_sum, l.sup_abs, l.sup_q, l.sup_flags, 1, 1, 0, (mss_subent *)nullptr, (mss_cand *)nullptr);
    DGRP_LAUNCH_CHECK();
    int rc = device_exclusive_scan(l.segcnt, l.segcnt, nrec, l.tiles, l.grand + 3, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(mss_compact_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, stream, l.segs, l.urun, l.segcnt,
                       l.grand + 3, nrec, l.segs_out, (const mss_subent *)nullptr);
    DGRP_LAUNCH_CHECK();
    DGRP_HIP(hipMemcpyAsync(d_labels_out, d_cls, total_n, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(mss_vote_kernel, dim3(1024), dim3(256), 0, stream, l.segs_out, l.grand + 3, d_cls, nof_labels, d_labels_out,
                       (unsigned long long *)nullptr, (int32_t *)nullptr, (int64_t)0);
    DGRP_LAUNCH_CHECK();
    // the urun vector must outlive the asynchronous copy
    DGRP_HIP(hipStreamSynchronize(stream));
    return DGRP_OK;
}


/*
 * dgrp_oracle.c -- CPU restatement of DeepGRP's prediction hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker and the timed
 * "cpu_baseline" of bench.py.  Nothing under deepgrp_amd/ (the product) may
 * import, link or call it; the product path is the HIP library and fails
 * loudly without it.
 *
 * Every function cites the reference location it restates (paths relative to
 * the upstream repository root).  Integer / byte / index stages are meant to be
 * bit-exact with the reference's compiled C/Cython (pinned by the fixtures in
 * tests/golden/, generated from the reference itself by oracle/make_golden.py).  The neural
 * network forward (orc_nn_forward_*) restates Keras semantics that live in
 * TensorFlow, which is not available offline: its numerics are
 * "parity unpinned" by reference outputs and are cross-checked against
 * torch.nn.GRU in tests instead.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* A2  one-hot encoder: deepgrp/sequence.pyx:11-36                           */
/* ------------------------------------------------------------------------- */

/* Class index of one byte.  The reference table (sequence.pyx:11-17) has 128
 * entries: A/a=0 C/c=1 G/g=2 T/t=3, everything else 4.  Bytes >= 128 index out
 * of bounds in the reference (undefined); we define them as 4. */
static inline uint8_t orc_class_of(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

/* Leading/trailing exact 'N' bytes are dropped (sequence.pyx:27-30).
 * Returns the kept length (may be negative for an all-N record, which makes
 * the reference raise ValueError: negative dimensions; callers mirror that). */
ORC_API int64_t orc_strip_n(const uint8_t *seq, int64_t len, int64_t *startpos)
{
    int64_t st = 0, en = len;
    while (st < en && seq[st] == 'N') ++st;
    /* the reference's second loop is independent of the first */
    en = len;
    while (en > 0 && seq[en - 1] == 'N') --en;
    *startpos = st;
    return en - st;
}

ORC_API void orc_encode_idx(const uint8_t *seq, int64_t n, uint8_t *idx)
{
    for (int64_t i = 0; i < n; ++i) idx[i] = orc_class_of(seq[i]);
}

/* int8 [5, n] C-order exactly as sequence.pyx:32-36 returns it */
ORC_API void orc_onehot_int8(const uint8_t *seq, int64_t n, int8_t *out)
{
    memset(out, 0, (size_t)(5 * n));
    for (int64_t i = 0; i < n; ++i) out[(int64_t)orc_class_of(seq[i]) * n + i] = 1;
}

/* ------------------------------------------------------------------------- */
/* A3  window enumeration: deepgrp/prediction.py:28-32                       */
/* ------------------------------------------------------------------------- */

/* len(range(0, N - T, s)) */
ORC_API int64_t orc_window_count(int64_t n, int64_t T, int64_t s)
{
    if (n - T <= 0) return 0;
    return (n - T + s - 1) / s;
}

/* f32 [nw, T, 5] windows w0..w0+nw-1 from class indices */
ORC_API void orc_windows_f32(const uint8_t *idx, int64_t T, int64_t s,
                             int64_t w0, int64_t nw, float *out)
{
    memset(out, 0, sizeof(float) * (size_t)(nw * T * 5));
    for (int64_t w = 0; w < nw; ++w)
        for (int64_t t = 0; t < T; ++t)
            out[(w * T + t) * 5 + idx[(w0 + w) * s + t]] = 1.0f;
}

/* ------------------------------------------------------------------------- */
/* A5  placement of window w in the per-base array: prediction.py:104-105    */
/* ------------------------------------------------------------------------- */

/* Row of the [N, C] array where window w's first position is max-merged.
 * index = i * batch.shape[0] * step uses the CURRENT batch's size, so windows
 * of a short last batch are misplaced.  Returned in rows. */
ORC_API int64_t orc_place_row(int64_t w, int64_t nwin, int64_t B, int64_t s)
{
    int64_t nfull = nwin / B, r = nwin % B;
    if (w < nfull * B) return w * s;
    return (nfull * r + (w - nfull * B)) * s;
}

/* ------------------------------------------------------------------------- */
/* A6  overlap max-merge: deepgrp/maxcalc.c:10-24                            */
/* ------------------------------------------------------------------------- */

ORC_API void orc_get_max(float *out, const float *in, int64_t dim0, int64_t dim1,
                         int64_t stride, int64_t batch)
{
    const int64_t per = dim0 * dim1, hop = stride * dim1;
    for (int64_t b = 0; b < batch; ++b) {
        float *o = out + b * hop;
        const float *x = in + b * per;
        for (int64_t i = 0; i < per; ++i)
            if (x[i] > o[i]) o[i] = x[i];   /* MAX(a,b) = a > b ? a : b with a = out */
    }
}

/* probs [nwin, T, C] of ALL windows -> merged [N, C] with the Q2 placement.
 * The caller zero-fills `out` (np.zeros, prediction.py:103).  A window whose
 * placement would run past row N is clipped exactly like the reference's
 * unchecked pointer walk would NOT be -- the reference never gets there because
 * placements only move windows to the left. */
ORC_API void orc_merge_all(float *out, int64_t N, const float *probs, int64_t nwin,
                           int64_t T, int64_t C, int64_t s, int64_t B)
{
    for (int64_t w = 0; w < nwin; ++w) {
        int64_t row = orc_place_row(w, nwin, B, s);
        const float *x = probs + w * T * C;
        for (int64_t t = 0; t < T && row + t < N; ++t)
            for (int64_t c = 0; c < C; ++c) {
                float *o = out + (row + t) * C + c;
                if (x[t * C + c] > *o) *o = x[t * C + c];
            }
    }
}

/* ------------------------------------------------------------------------- */
/* numpy float32 log / exp                                                   */
/* ------------------------------------------------------------------------- */
/*
 * prediction.py:55 calls np.log on a float32 array and prediction.py:64 np.exp.
 * numpy (>= 1.17, incl. the reference's pinned 1.19.5 and this image's 2.2.6) on
 * any AVX2/AVX512F x86 host computes those with its own SIMD routines, which are
 * NOT correctly rounded (log: max 3.83 ulp, exp: 2.52 ulp) but are deterministic
 * fma chains.  The two functions below restate those published algorithms
 * (numpy/core/src/umath/loops_exponent_log.dispatch.c.src; numpy is a
 * dependency of the reference, pyproject.toml, not part of its tree).  They were
 * checked bit-for-bit against np.log / np.exp of the installed numpy on 6e6
 * random float32 values each (tests/test_oracle_numpy_math.py repeats that).
 */
static const float LP1 = 9.999999999999998702752e-01f, LP2 = 2.112677543073053063722e+00f,
                   LP3 = 1.480000633576506585156e+00f, LP4 = 3.808837741388407920751e-01f,
                   LP5 = 2.589979117907922693523e-02f, LQ1 = 2.612677543073109236779e+00f,
                   LQ2 = 2.453006071784736363091e+00f, LQ3 = 9.864942958519418960339e-01f,
                   LQ4 = 1.546476374983906719538e-01f, LQ5 = 5.875095403124574342950e-03f;

ORC_API float orc_np_logf(float xin)
{
    if (!(xin > 0.0f)) return xin == 0.0f ? -INFINITY : NAN;
    if (isinf(xin)) return xin;
    int e;
    float m = frexpf(xin, &e);            /* m in [0.5, 1) */
    float k = (float)e;
    if (m <= 0.70710678118654752440f) { m = m + m; k = k - 1.0f; }
    float x = m - 1.0f;
    float den = fmaf(LQ5, x, LQ4);
    den = fmaf(den, x, LQ3); den = fmaf(den, x, LQ2); den = fmaf(den, x, LQ1); den = fmaf(den, x, 1.0f);
    float num = fmaf(LP5, x, LP4);
    num = fmaf(num, x, LP3); num = fmaf(num, x, LP2); num = fmaf(num, x, LP1); num = fmaf(num, x, 0.0f);
    return fmaf(k, 0.693147180559945309417232121458176568f, num / den);
}

static const float EP0 = 9.999999999980870924916e-01f, EP1 = 7.257664613233124478488e-01f,
                   EP2 = 2.473615434895520810817e-01f, EP3 = 5.114512081637298353406e-02f,
                   EP4 = 6.757896990527504603057e-03f, EP5 = 5.082762527590693718096e-04f,
                   EQ1 = -2.742335390411667452936e-01f, EQ2 = 2.159509375685829852307e-02f;

ORC_API float orc_np_expf(float x)
{
    if (isnan(x)) return x;
    if (x >= 88.72283905206835f) return INFINITY;
    if (x <= -103.97208f) return 0.0f;
    float q = rintf(x * 1.44269504088896340736f);
    float r = fmaf(q, -6.93145752e-1f, x);
    r = fmaf(q, -1.42860677e-6f, r);
    float num = fmaf(EP5, r, EP4);
    num = fmaf(num, r, EP3); num = fmaf(num, r, EP2); num = fmaf(num, r, EP1); num = fmaf(num, r, EP0);
    float den = fmaf(EQ2, r, EQ1);
    den = fmaf(den, r, 1.0f);
    return ldexpf(num / den, (int)q);
}

/* ------------------------------------------------------------------------- */
/* A7  MSS score transform: deepgrp/prediction.py:51-57                      */
/* ------------------------------------------------------------------------- */

/* probs f32 [N, C] -> scores f64 [N], classes i64 [N].  All arithmetic in
 * float32 like numpy does for a float32 array with Python-float scalars. */
ORC_API void orc_scores(const float *probs, int64_t N, int64_t C, double *scores, int64_t *cls)
{
    const float eps = 1e-6f, cap = 0.99f;
    for (int64_t i = 0; i < N; ++i) {
        const float *p = probs + i * C;
        int64_t a = 0;
        float mx = p[0];
        for (int64_t c = 1; c < C; ++c)
            if (p[c] > mx) { mx = p[c]; a = c; }          /* argmax: first maximum */
        float m = mx + eps;
        if (m > cap) m = cap;
        float one_minus = 1.0f - m;
        float t = orc_np_logf(m / one_minus);
        float sc = a > 0 ? t : -10.0f * t;
        scores[i] = (double)sc;
        cls[i] = a;
    }
}

/* A8  prediction.py:62-65 followed by __main__.py:83 (argmax).  out = softmaxed
 * [N, C] (may be NULL), cls = argmax over axis 1. */
ORC_API void orc_softmax_argmax(const float *a, int64_t N, int64_t C, float *out, int64_t *cls)
{
    float gmax = -INFINITY;
    for (int64_t i = 0; i < N * C; ++i)
        if (a[i] > gmax) gmax = a[i];
    float *row = (float *)malloc(sizeof(float) * (size_t)(C > 0 ? C : 1));
    for (int64_t i = 0; i < N; ++i) {
        /* numpy sums a length-C row pairwise; for C < 8 that is a plain left fold */
        float sum = 0.0f;
        for (int64_t c = 0; c < C; ++c) { row[c] = orc_np_expf(a[i * C + c] - gmax); sum += row[c]; }
        int64_t best = 0;
        float bv = row[0] / sum;
        if (out) out[i * C] = bv;
        for (int64_t c = 1; c < C; ++c) {
            float v = row[c] / sum;
            if (out) out[i * C + c] = v;
            if (v > bv) { bv = v; best = c; }
        }
        cls[i] = best;
    }
    free(row);
}

/* ------------------------------------------------------------------------- */
/* A10  all maximal scoring segments with x-drop: deepgrp/_mss/mss.c:50-101   */
/* ------------------------------------------------------------------------- */

typedef struct { int32_t st, en; double sc; } orc_seg_t;          /* mss.h:11-14 */
typedef struct { int32_t st, en; double L, R; int32_t pre; } cand_t;   /* mss.c:24-28 */

typedef struct { orc_seg_t *a; size_t n, cap; } segvec_t;
typedef struct { cand_t *a; size_t n, cap; } candvec_t;

static void segvec_push(segvec_t *v, orc_seg_t x)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 16;
        v->a = (orc_seg_t *)realloc(v->a, v->cap * sizeof(orc_seg_t));
    }
    v->a[v->n++] = x;
}

static void candvec_push(candvec_t *v, cand_t x)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 16;
        v->a = (cand_t *)realloc(v->a, v->cap * sizeof(cand_t));
    }
    v->a[v->n++] = x;
}

/* mss.c:35-47: emit every stacked candidate whose score reaches the (integer,
 * truncated) threshold, then clear the stack. */
static void flush_stack(segvec_t *out, candvec_t *st, int min_sc)
{
    for (size_t i = 0; i < st->n; ++i) {
        double sc = st->a[i].R - st->a[i].L;
        if (sc >= min_sc) {
            orc_seg_t s = { st->a[i].st, st->a[i].en, sc };
            segvec_push(out, s);
        }
    }
    st->n = 0;
}

/* Returns a malloc'd array (caller frees with orc_free) and its length. */
ORC_API orc_seg_t *orc_mss_find_all(int32_t n, const double *S, double min_sc_f, double xdrop,
                                    int32_t *n_seg)
{
    const int min_sc = (int)min_sc_f;       /* mss.c:35 takes an int: truncation */
    segvec_t out = { 0, 0, 0 };
    candvec_t st = { 0, 0, 0 };
    double L = 0.0, peak = -1e30;           /* NEG_INF, mss.c:33 */
    int32_t i = 0;
    while (i < n) {
        if (S[i] > 0) {
            /* a maximal run of positive scores becomes one candidate (mss.c:59-64) */
            double R = L + S[i];
            int32_t k = i + 1;
            while (k < n && S[k] > 0.) { R += S[k]; ++k; }
            if (R > peak) peak = R;
            cand_t t;
            t.st = i; t.en = k; t.L = L; t.R = R; t.pre = -1;
            for (;;) {
                /* walk left for the nearest candidate that starts lower (mss.c:68-72) */
                int64_t j = (int64_t)st.n - 1;
                while (j >= 0) {
                    const cand_t *p = &st.a[j];
                    if (p->L < t.L) break;
                    j = p->pre >= 0 ? p->pre : j - 1;
                }
                if (j >= 0 && st.a[j].R < t.R) {
                    /* absorb everything from j upward (mss.c:73-76) */
                    t.st = st.a[j].st;
                    t.L = st.a[j].L;
                    st.n = (size_t)j;
                    continue;
                }
                if (j < 0) {                 /* nothing to the left can still grow (mss.c:78-81) */
                    flush_stack(&out, &st, min_sc);
                    peak = R;
                }
                t.pre = (int32_t)j;
                candvec_push(&st, t);
                break;
            }
            L = R;
            i = k;
        } else {
            if (xdrop > 0.0 && L + S[i] + xdrop < peak) {   /* x-drop reset (mss.c:89-92) */
                flush_stack(&out, &st, min_sc);
                L = 0.0;
                peak = -1e30;
            }
            L += S[i];
            ++i;
        }
    }
    flush_stack(&out, &st, min_sc);
    free(st.a);
    *n_seg = (int32_t)out.n;
    return out.a;
}

ORC_API void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------------- */
/* A9  label vote + fill: deepgrp/_mss/pymss.pyx:31-80                        */
/* ------------------------------------------------------------------------- */

/* Writes the label (the argmax of the reference's float64 one-hot rows, which
 * __main__.py:83 takes next) for every position.  Returns the segment count;
 * if segs_out != NULL it receives the malloc'd segment array. */
ORC_API int32_t orc_find_mss_labels(const double *scores, const int64_t *label, int32_t n,
                                    int32_t nof_labels, int32_t min_mss_len, int32_t xdrop_len,
                                    int64_t *out_label, orc_seg_t **segs_out)
{
    const double s0 = log(0.99 / (1.0 - 0.99));                     /* pymss.pyx:46 */
    const double xdrop = xdrop_len > 0 ? s0 * xdrop_len * 10.0 : -1; /* :48-51 */
    const double min_sc = s0 * min_mss_len;                          /* :53 */
    int32_t nseg = 0;
    orc_seg_t *segs = orc_mss_find_all(n, scores, min_sc, xdrop, &nseg);
    int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * (size_t)nof_labels);
    int64_t pos = 0;
    for (int32_t i = 0; i < nseg; ++i) {
        for (int32_t j = 0; j < nof_labels; ++j) cnt[j] = 0;
        for (int64_t j = segs[i].st; j < segs[i].en; ++j) cnt[label[j]]++;
        int32_t best = 1;
        int64_t bv = cnt[1];
        for (int32_t j = 2; j < nof_labels; ++j)
            if (bv < cnt[j]) { best = j; bv = cnt[j]; }              /* strict: first maximum wins */
        for (int64_t j = segs[i].st; j < segs[i].en; ++j)
            out_label[j] = label[j] == 0 ? best : label[j];
        for (int64_t j = pos; j < segs[i].st; ++j) out_label[j] = label[j];
        pos = segs[i].en;
    }
    for (int64_t j = pos; j < n; ++j) out_label[j] = label[j];
    free(cnt);
    if (segs_out) *segs_out = segs; else free(segs);
    return nseg;
}

/* ------------------------------------------------------------------------- */
/* A11  segment extraction: deepgrp/sequence.pyx:38-53, :79-85               */
/* ------------------------------------------------------------------------- */

/* One call of get_segments(classes, startpos). */
ORC_API void orc_get_segment(const int64_t *classes, int64_t size, int64_t startpos,
                             int64_t *st, int64_t *en, int64_t *lab)
{
    const int64_t length = size - 1;          /* the quirk: the last element is never scanned over */
    int64_t cur = classes[startpos];
    while (startpos < length && cur == 0) { ++startpos; cur = classes[startpos]; }
    int64_t end = startpos + 1;
    while (end < length && classes[end] == cur) ++end;
    *st = startpos; *en = end; *lab = cur;
}

/* The whole yield_segments loop, keeping only label > 0 rows (__main__.py:290).
 * rec receives (start+offset, end+offset, label) triples; returns the count
 * (call with rec == NULL to count). */
ORC_API int64_t orc_segments(const int64_t *classes, int64_t size, int64_t offset, int64_t *rec)
{
    int64_t i = 0, n = 0;
    while (i < size) {
        int64_t st, en, lab;
        orc_get_segment(classes, size, i, &st, &en, &lab);
        i = en;
        if (lab > 0) {
            if (rec) { rec[3 * n] = st + offset; rec[3 * n + 1] = en + offset; rec[3 * n + 2] = lab; }
            ++n;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------- */
/* N2  evaluation helpers next to the path: deepgrp/prediction.py:204-260      */
/* ------------------------------------------------------------------------- */
/* confusion_matrix (prediction.py:204-222): n_classes = max over both arrays - min over both + 1, a zero
 * int matrix, then cnf[t, p] += 1 element by element -- indexed with the RAW labels (the minimum is not
 * subtracted), so a label outside [-n_classes, n_classes) raises IndexError in numpy and a negative one wraps.
 * Returns n_classes, or -1 where numpy would raise.  cnf must hold n_classes^2 entries (call with cnf == NULL
 * to get n_classes first). */
ORC_API int64_t orc_confusion_matrix(const int64_t *truelbl, const int64_t *predlbl, int64_t n, int64_t *cnf)
{
    if (n <= 0) return -1;                                   /* max() of an empty array raises */
    int64_t lo = truelbl[0], hi = truelbl[0];
    for (int64_t i = 0; i < n; ++i) {
        if (truelbl[i] < lo) lo = truelbl[i];
        if (truelbl[i] > hi) hi = truelbl[i];
        if (predlbl[i] < lo) lo = predlbl[i];
        if (predlbl[i] > hi) hi = predlbl[i];
    }
    const int64_t k = hi - lo + 1;
    if (!cnf) return k;
    for (int64_t i = 0; i < k * k; ++i) cnf[i] = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t t = truelbl[i], q = predlbl[i];
        if (t < -k || t >= k || q < -k || q >= k) return -1;
        if (t < 0) t += k;
        if (q < 0) q += k;
        cnf[t * k + q] += 1;
    }
    return k;
}

/* filter_segments (prediction.py:244-260), in place: walk the positive positions; a run of equal labels that
 * starts at one of them and is shorter than min_len is zeroed. */
ORC_API void orc_filter_segments(int64_t *array, int64_t n, int64_t min_len)
{
    int64_t next_idx = 0;
    for (int64_t idx = 0; idx < n; ++idx) {
        if (!(array[idx] > 0) || next_idx > idx) continue;
        next_idx = idx + 1;
        int64_t found = 1;
        while (next_idx < n && array[next_idx] == array[idx]) { ++found; ++next_idx; }
        if (found < min_len)
            for (int64_t k = idx; k < next_idx; ++k) array[k] = 0;
    }
}

/* ------------------------------------------------------------------------- */
/* A4  model forward: deepgrp/model.py:293-336 (graph), Keras 2.5 layer math  */
/* ------------------------------------------------------------------------- */
/*
 * Keras GRU(reset_after=True, activation=tanh, recurrent_activation=sigmoid,
 * use_bias=True) -- defaults pinned by the reference's tests/test_model.json --
 * with kernel [5,3u], recurrent_kernel [u,3u], bias [2,3u], gate columns z|r|h:
 *     g  = h_prev . U + b_rec
 *     z  = sigmoid(x.W_z + b_in_z + g_z)      r = sigmoid(x.W_r + b_in_r + g_r)
 *     hh = tanh(x.W_h + b_in_h + r * g_h)     h = z * h_prev + (1 - z) * hh
 * The same layer runs on the window and on its reverse complement
 * (model.py:266-279, complement table [3,2,1,0,4] model.py:233-237); the two
 * output sequences are averaged WITHOUT re-reversing the second (model.py:312,
 * :323).  Optional AdditiveAttention(use_scale=True) with the
 * averaged final states as the single query (model.py:309-319), then Dense and
 * Softmax over classes (model.py:325-329).
 *
 * Templated on the scalar type by macro: real_t = float gives the timed CPU
 * baseline, real_t = double the tolerance reference.
 */
static const int ORC_COMP[5] = { 3, 2, 1, 0, 4 };

#define ORC_NN_IMPL(NAME, real_t, EXP, TANH)                                                     \
    static void NAME##_one(const uint8_t *base, int T, int u, int C, int attention,              \
                           const real_t *Wx, const real_t *U, const real_t *bi, const real_t *br, \
                           const real_t *scale, const real_t *Wd, const real_t *bd,              \
                           real_t *work, real_t *probs)                                          \
    {                                                                                            \
        real_t *hf = work, *hr = work + u, *g = work + 2 * u, *avg = work + 5 * u;               \
        real_t *ctx = avg + (size_t)T * u, *e = ctx + u;                                         \
        for (int k = 0; k < u; ++k) hf[k] = hr[k] = 0;                                           \
        for (int t = 0; t < T; ++t) {                                                            \
            for (int dir = 0; dir < 2; ++dir) {                                                  \
                real_t *h = dir ? hr : hf;                                                       \
                int b = dir ? ORC_COMP[base[T - 1 - t]] : base[t];                               \
                const real_t *xw = Wx + (size_t)b * 3 * u;                                       \
                for (int j = 0; j < 3 * u; ++j) g[j] = br[j];                                    \
                for (int k = 0; k < u; ++k) {                                                    \
                    const real_t hk = h[k];                                                      \
                    const real_t *Uk = U + (size_t)k * 3 * u;                                    \
                    for (int j = 0; j < 3 * u; ++j) g[j] += hk * Uk[j];                          \
                }                                                                                \
                for (int k = 0; k < u; ++k) {                                                    \
                    real_t z = 1 / (1 + EXP(-(xw[k] + bi[k] + g[k])));                           \
                    real_t r = 1 / (1 + EXP(-(xw[u + k] + bi[u + k] + g[u + k])));               \
                    real_t hh = TANH(xw[2 * u + k] + bi[2 * u + k] + r * g[2 * u + k]);          \
                    h[k] = z * h[k] + (1 - z) * hh;                                              \
                }                                                                                \
            }                                                                                    \
            for (int k = 0; k < u; ++k) avg[(size_t)t * u + k] = (hf[k] + hr[k]) / 2;            \
        }                                                                                        \
        if (attention) {                                                                         \
            /* query = Average([h_fwd_T, h_rc_T]); scores = sum_k scale*tanh(q + key) */         \
            real_t mx = -INFINITY, den = 0;                                                      \
            for (int t = 0; t < T; ++t) {                                                        \
                real_t acc = 0;                                                                  \
                for (int k = 0; k < u; ++k)                                                      \
                    acc += scale[k] * TANH((hf[k] + hr[k]) / 2 + avg[(size_t)t * u + k]);        \
                e[t] = acc;                                                                      \
                if (acc > mx) mx = acc;                                                          \
            }                                                                                    \
            for (int t = 0; t < T; ++t) { e[t] = EXP(e[t] - mx); den += e[t]; }                  \
            for (int k = 0; k < u; ++k) ctx[k] = 0;                                              \
            for (int t = 0; t < T; ++t) {                                                        \
                real_t a = e[t] / den;                                                           \
                for (int k = 0; k < u; ++k) ctx[k] += a * avg[(size_t)t * u + k];                \
            }                                                                                    \
        }                                                                                        \
        for (int t = 0; t < T; ++t) {                                                            \
            real_t lg[64], mx = -INFINITY, den = 0;                                              \
            for (int c = 0; c < C; ++c) {                                                        \
                real_t acc = bd[c];                                                              \
                if (attention) {                                                                 \
                    for (int k = 0; k < u; ++k) acc += ctx[k] * Wd[(size_t)k * C + c];           \
                    for (int k = 0; k < u; ++k)                                                  \
                        acc += avg[(size_t)t * u + k] * Wd[(size_t)(u + k) * C + c];             \
                } else {                                                                         \
                    for (int k = 0; k < u; ++k) acc += avg[(size_t)t * u + k] * Wd[(size_t)k * C + c]; \
                }                                                                                \
                lg[c] = acc;                                                                     \
                if (acc > mx) mx = acc;                                                          \
            }                                                                                    \
            for (int c = 0; c < C; ++c) { lg[c] = EXP(lg[c] - mx); den += lg[c]; }               \
            for (int c = 0; c < C; ++c) probs[(size_t)t * C + c] = lg[c] / den;                  \
        }                                                                                        \
    }                                                                                            \
    ORC_API int NAME(const uint8_t *idx, int64_t s, int64_t w0, int64_t nw, int T, int u, int C, \
                     int attention, const real_t *Wx, const real_t *U, const real_t *bias,       \
                     const real_t *scale, const real_t *Wd, const real_t *bd, real_t *probs,     \
                     int threads)                                                                \
    {                                                                                            \
        if (C > 64) return -1;                                                                   \
        size_t wsz = (size_t)5 * u + (size_t)T * u + u + T;                                      \
        int nt = threads > 0 ? threads : 1;                                                      \
        real_t *work = (real_t *)malloc(sizeof(real_t) * wsz * (size_t)nt);                      \
        if (!work) return -2;                                                                    \
        _Pragma("omp parallel for num_threads(nt) schedule(dynamic, 4)")                         \
        for (int64_t w = 0; w < nw; ++w) {                                                       \
            int tid = 0;                                                                         \
            ORC_TID(tid);                                                                        \
            NAME##_one(idx + (w0 + w) * s, T, u, C, attention, Wx, U, bias, bias + 3 * u, scale, \
                       Wd, bd, work + wsz * (size_t)tid, probs + (size_t)w * T * C);             \
        }                                                                                        \
        free(work);                                                                              \
        return 0;                                                                                \
    }

#ifdef _OPENMP
#define ORC_TID(t) (t) = omp_get_thread_num()
#else
#define ORC_TID(t) (t) = 0
#endif

/* probs [nw, T, C] for windows w0..w0+nw-1 (window w starts at idx[w*s]) */
ORC_NN_IMPL(orc_nn_forward_f32, float, expf, tanhf)
ORC_NN_IMPL(orc_nn_forward_f64, double, exp, tanh)

/* ------------------------------------------------------------------------- */
/* A4 (rnn = "LSTM"): deepgrp/model.py:219-223, :321-323                      */
/* ------------------------------------------------------------------------- */
/*
 * Keras LSTM(units, activation=tanh, recurrent_activation=sigmoid, use_bias=True) -- the defaults
 * recorded for the BLSTM layer in the reference's tests/test_model.json -- with kernel [5,4u],
 * recurrent_kernel [u,4u], bias [4u], gate columns i|f|c|o:
 *     z = x.W + h_prev.U + b ;  i = sig(z_i)  f = sig(z_f)  o = sig(z_o)
 *     c = f * c_prev + i * tanh(z_c) ;  h = o * tanh(c)
 * No attention with this cell (model.py:308); same reverse-complement / Average / Dense / Softmax.
 */
#define ORC_LSTM_IMPL(NAME, real_t, EXP, TANH)                                                   \
    static void NAME##_one(const uint8_t *base, int T, int u, int C, const real_t *Wx,           \
                           const real_t *U, const real_t *b, const real_t *Wd, const real_t *bd, \
                           real_t *work, real_t *probs)                                          \
    {                                                                                            \
        real_t *hf = work, *hr = work + u, *cf = work + 2 * u, *cr = work + 3 * u;               \
        real_t *g = work + 4 * u, *avg = work + 8 * u;                                           \
        for (int k = 0; k < u; ++k) hf[k] = hr[k] = cf[k] = cr[k] = 0;                           \
        for (int t = 0; t < T; ++t) {                                                            \
            for (int dir = 0; dir < 2; ++dir) {                                                  \
                real_t *h = dir ? hr : hf, *c = dir ? cr : cf;                                   \
                int bs = dir ? ORC_COMP[base[T - 1 - t]] : base[t];                              \
                const real_t *xw = Wx + (size_t)bs * 4 * u;                                      \
                for (int j = 0; j < 4 * u; ++j) g[j] = xw[j] + b[j];                             \
                for (int k = 0; k < u; ++k) {                                                    \
                    const real_t hk = h[k];                                                      \
                    const real_t *Uk = U + (size_t)k * 4 * u;                                    \
                    for (int j = 0; j < 4 * u; ++j) g[j] += hk * Uk[j];                          \
                }                                                                                \
                for (int k = 0; k < u; ++k) {                                                    \
                    real_t ig = 1 / (1 + EXP(-g[k]));                                            \
                    real_t fg = 1 / (1 + EXP(-g[u + k]));                                        \
                    real_t og = 1 / (1 + EXP(-g[3 * u + k]));                                    \
                    c[k] = fg * c[k] + ig * TANH(g[2 * u + k]);                                  \
                    h[k] = og * TANH(c[k]);                                                      \
                }                                                                                \
            }                                                                                    \
            for (int k = 0; k < u; ++k) avg[(size_t)t * u + k] = (hf[k] + hr[k]) / 2;            \
        }                                                                                        \
        for (int t = 0; t < T; ++t) {                                                            \
            real_t lg[64], mx = -INFINITY, den = 0;                                              \
            for (int c = 0; c < C; ++c) {                                                        \
                real_t acc = bd[c];                                                              \
                for (int k = 0; k < u; ++k) acc += avg[(size_t)t * u + k] * Wd[(size_t)k * C + c]; \
                lg[c] = acc;                                                                     \
                if (acc > mx) mx = acc;                                                          \
            }                                                                                    \
            for (int c = 0; c < C; ++c) { lg[c] = EXP(lg[c] - mx); den += lg[c]; }               \
            for (int c = 0; c < C; ++c) probs[(size_t)t * C + c] = lg[c] / den;                  \
        }                                                                                        \
    }                                                                                            \
    ORC_API int NAME(const uint8_t *idx, int64_t s, int64_t w0, int64_t nw, int T, int u, int C, \
                     const real_t *Wx, const real_t *U, const real_t *bias, const real_t *Wd,    \
                     const real_t *bd, real_t *probs, int threads)                               \
    {                                                                                            \
        if (C > 64) return -1;                                                                   \
        size_t wsz = (size_t)8 * u + (size_t)T * u;                                              \
        int nt = threads > 0 ? threads : 1;                                                      \
        real_t *work = (real_t *)malloc(sizeof(real_t) * wsz * (size_t)nt);                      \
        if (!work) return -2;                                                                    \
        _Pragma("omp parallel for num_threads(nt) schedule(dynamic, 4)")                         \
        for (int64_t w = 0; w < nw; ++w) {                                                       \
            int tid = 0;                                                                         \
            ORC_TID(tid);                                                                        \
            NAME##_one(idx + (w0 + w) * s, T, u, C, Wx, U, bias, Wd, bd, work + wsz * (size_t)tid, \
                       probs + (size_t)w * T * C);                                               \
        }                                                                                        \
        free(work);                                                                              \
        return 0;                                                                                \
    }

ORC_LSTM_IMPL(orc_lstm_forward_f32, float, expf, tanhf)
ORC_LSTM_IMPL(orc_lstm_forward_f64, double, exp, tanh)

ORC_API int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

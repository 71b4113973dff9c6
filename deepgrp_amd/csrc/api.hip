// C-ABI glue of libdeepgrp_hip.so: error channel, device query, model packing/upload and the
// forward entry points that sequence the GRU and attention kernels.
#include "dgrp_model.h"

#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <vector>
#include <mutex>

static thread_local char g_err[512] = "";

void dgrp_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

DGRP_EXPORT int dgrp_abi_version(void) { return DGRP_ABI_VERSION; }
DGRP_EXPORT const char *dgrp_last_error(void) { return g_err; }

DGRP_EXPORT int dgrp_device_info(char *name, size_t name_cap, int *cu_count, int64_t *hbm_bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { dgrp_set_error("no HIP device: %s", hipGetErrorString(e)); return DGRP_ENODEV; }
    hipDeviceProp_t prop;
    DGRP_HIP(hipGetDeviceProperties(&prop, dev));
    if (name && name_cap) snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        dgrp_set_error("device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName);
        return DGRP_ENODEV;
    }
    return DGRP_OK;
}

// ---- kernel timer (include/deepgrp_hip.h, "instrumentation") --------------------------------------------------------
struct timed_launch { hipEvent_t start, stop; int64_t windows; };
static thread_local bool g_timer_on = false;
static thread_local std::vector<timed_launch> g_timed;

dgrp_timer_scope::dgrp_timer_scope(hipStream_t s, int64_t nw) : stream(s), windows(nw), start(nullptr), on(g_timer_on)
{
    if (!on) return;
    if (hipEventCreate(&start) != hipSuccess) { on = false; start = nullptr; return; }
    if (hipEventRecord(start, stream) != hipSuccess) { on = false; (void)hipEventDestroy(start); start = nullptr; }
}
dgrp_timer_scope::~dgrp_timer_scope()
{
    if (!on) return;
    hipEvent_t stop = nullptr;
    if (hipEventCreate(&stop) == hipSuccess && hipEventRecord(stop, stream) == hipSuccess) {
        g_timed.push_back({ start, stop, windows });
        return;
    }
    if (stop) (void)hipEventDestroy(stop);                    // no pair, no measurement: neither event may leak
    (void)hipEventDestroy(start);
}

static void timer_clear()
{
    for (auto &t : g_timed) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    g_timed.clear();
}

DGRP_EXPORT int dgrp_kernel_timer_enable(int on)
{
    timer_clear();
    g_timer_on = on != 0;
    return DGRP_OK;
}

DGRP_EXPORT int dgrp_kernel_timer_read(double *h_ms, int64_t *h_launches, int64_t *h_windows)
{
    double ms = 0.0;
    int64_t windows = 0;
    for (auto &t : g_timed) {
        DGRP_HIP(hipEventSynchronize(t.stop));
        float one = 0.0f;
        DGRP_HIP(hipEventElapsedTime(&one, t.start, t.stop));
        ms += one;
        windows += t.windows;
    }
    if (h_ms) *h_ms = ms;
    if (h_launches) *h_launches = (int64_t)g_timed.size();
    if (h_windows) *h_windows = windows;
    timer_clear();
    return DGRP_OK;
}

// deepgrp/sequence.pyx:27-30
DGRP_EXPORT int dgrp_strip_n(const uint8_t *h_seq, int64_t len, int64_t *startpos, int64_t *kept)
{
    DGRP_REQUIRE(len >= 0 && startpos && kept && (len == 0 || h_seq), "dgrp_strip_n: bad arguments");
    int64_t st = 0, en = len;
    while (st < len && h_seq[st] == 'N') ++st;
    while (en > 0 && h_seq[en - 1] == 'N') --en;
    *startpos = st;
    *kept = en - st;
    return DGRP_OK;
}

// ------------------------------------------------------------------------------------------
// Fragment packing (host).  Operand lane maps of gfx950:
//   v_mfma_f32_32x32x16_f16   B: lane l, element j  <->  B[k = 8*(l>>5) + j][col = l & 31]
//   v_mfma_f32_16x16x32_f16   B: lane l, element j  <->  B[k = 8*(l>>4) + j][col = l & 15]
// ------------------------------------------------------------------------------------------
static inline uint16_t f2h(float v)
{
    _Float16 h = (_Float16)v;
    uint16_t b;
    memcpy(&b, &h, 2);
    return b;
}
static inline float h2f(uint16_t b)
{
    _Float16 h;
    memcpy(&h, &b, 2);
    return (float)h;
}

// ---- fragment packing shared by the GRU and the LSTM constructor (layouts: gru_kernel.hip) -----------------------
struct frag_writer {
    std::vector<uint16_t> &pack;
    int NF;
    uint16_t &at(int w, int f, int l, int j) { return pack[(((size_t)w * NF + f) * 64 + l) * 8 + j]; }
};

// k-steps 0..KS-1 of one gate: lane l of wave w holds rows 16 ks + 8 (l >> 5) + j of column `col` (scaled by gs)
static void pack_recurrent(frag_writer fw, int w, int l, int frag0, int KS, const float *rec, int ld, int col, int u, bool uok, float gs)
{
    for (int ks = 0; ks < KS; ++ks)
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * ks + 8 * (l >> 5) + j;
            fw.at(w, frag0 + ks, l, j) = (uok && k < u) ? f2h(gs * rec[(size_t)k * ld + col]) : 0;
        }
}

// the input k-step: 8 values (5 kernel rows, bias, 2 x zero) as fp16 hi parts in lanes 0-31 and lo parts in lanes 32-63
static void pack_input(frag_writer fw, int w, int l, int frag, const float v8[8])
{
    for (int j = 0; j < 8; ++j) {
        const uint16_t hi = f2h(v8[j]);
        fw.at(w, frag, l, j) = (l >> 5) == 0 ? hi : f2h(v8[j] - h2f(hi));
    }
}

// Dense (16x16x32): 0.5 * FF kernel rows of this wave's 32 units (row offset `row0` for the avg half with attention)
static void pack_dense(frag_writer fw, int w, int l, int frag_hi, int frag_lo, const float *ffk, int row0, int u, int C)
{
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * (l >> 4) + j, du = 32 * w + k, c = l & 15;
        float v = 0.0f;
        if (du < u && c < C) v = 0.5f * ffk[(size_t)(row0 + du) * C + c];
        const uint16_t hi = f2h(v);
        fw.at(w, frag_hi, l, j) = hi;
        fw.at(w, frag_lo, l, j) = f2h(v - h2f(hi));
    }
}

// fp32 copies of the tensors as given (ref_kernels.hip reads them); one allocation, offsets in floats
static hipError_t upload_raw(dgrp_model *m, const float *kernel, const float *rec, const float *bias, int64_t nbias,
                             const float *ffk, int64_t nffk, const float *ffb, const float *scale)
{
    const int64_t G = m->cell ? 4 : 3, u = m->u;
    std::vector<float> raw;
    auto put = [&](const float *p, int64_t cnt) {
        const int64_t off = (int64_t)raw.size();
        if (p) raw.insert(raw.end(), p, p + cnt);
        else raw.insert(raw.end(), (size_t)cnt, 0.0f);
        raw.resize((raw.size() + 63) / 64 * 64, 0.0f);
        return off;
    };
    m->raw_kernel = put(kernel, 5 * G * u);
    m->raw_rec = put(rec, u * G * u);
    m->raw_bias = put(bias, nbias);
    m->raw_ffk = put(ffk, nffk);
    m->raw_ffb = put(ffb, m->C);
    m->raw_scale = put(scale, u);
    hipError_t e = hipMalloc((void **)&m->d_raw, raw.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(m->d_raw, raw.data(), raw.size() * sizeof(float), hipMemcpyHostToDevice);
    return e;
}

// recurrent fragments for rnn_split_stream_kernel: [NW][KS][2 G][64][8], hi halves of gates 0..G-1 then their lo halves per k-step
// (lane l of wave w: rows 16 ks + 8 (l >> 5) + j of column g*u + unit, unit = 32 w + (l & 31), scaled by gs[g])
static hipError_t upload_stream(dgrp_model *m, const float *rec, int G, const float *gs)
{
    const int KS = m->KS, u = m->u, ld = G * u;
    std::vector<uint16_t> st((size_t)m->NW * KS * 2 * G * 64 * 8, 0);
    for (int w = 0; w < m->NW; ++w)
        for (int ks = 0; ks < KS; ++ks)
            for (int g = 0; g < G; ++g)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int unit = 32 * w + (l & 31), k = 16 * ks + 8 * (l >> 5) + j;
                        if (unit >= u || k >= u) continue;
                        const float x = gs[g] * rec[(size_t)k * ld + g * u + unit];
                        const uint16_t hi = f2h(x);
                        st[((((size_t)w * KS + ks) * 2 * G + g) * 64 + l) * 8 + j] = hi;
                        st[((((size_t)w * KS + ks) * 2 * G + G + g) * 64 + l) * 8 + j] = f2h(x - h2f(hi));
                    }
    hipError_t e = hipMalloc((void **)&m->d_stream, st.size() * 2);
    if (e == hipSuccess) e = hipMemcpy(m->d_stream, st.data(), st.size() * 2, hipMemcpyHostToDevice);
    return e;
}

// Models beyond the fused kernels' sizes: nothing is packed, the fp32 tensors as given are all the device holds, and every forward
// call goes through ref_kernels.hip (forward_ref below).  The reference builds its RNN layer with any `units`
// (deepgrp/model.py:117,219-229): such a model is slow here (tens of Mbp/s), not refused.
static int create_ref_only(dgrp_model **out, dgrp_model *m, const float *kernel, const float *rec, const float *bias, int64_t nbias,
                           const float *ffk, int64_t nffk, const float *ffb, const float *scale)
{
    m->ref_only = 1;
    m->onercp = 0;
    m->UP = (m->u + 31) / 32 * 32; m->NW = m->UP / 32; m->KS = m->UP / 16; m->nfrag = 0;
    m->d_pack = nullptr; m->d_ffb = nullptr; m->d_scale = nullptr; m->d_wtop = nullptr; m->d_raw = nullptr;
    m->d_pack_lo = nullptr; m->precision = 1; m->d_pack16 = nullptr; m->d_xtab = nullptr; m->d_stream = nullptr;
    m->d_packw = nullptr; m->d_xtabw = nullptr; m->NU16 = 0;
    const hipError_t e = upload_raw(m, kernel, rec, bias, nbias, ffk, nffk, ffb, scale);
    if (e != hipSuccess) {
        dgrp_set_error("dgrp_model_create: %s", hipGetErrorString(e));
        dgrp_model_destroy(m);
        return DGRP_EHIP;
    }
    *out = m;
    return DGRP_OK;
}

DGRP_EXPORT int dgrp_model_create(dgrp_model **out, int T, int u, int C, int attention, const float *kernel,
                                  const float *rec, const float *bias, const float *scale, const float *ffk,
                                  const float *ffb)
{
    DGRP_REQUIRE(out, "dgrp_model_create: NULL out");
    *out = nullptr;
    DGRP_REQUIRE(T >= 1 && T <= 65535, "dgrp_model_create: window size %d out of range", T);
    DGRP_REQUIRE(u >= 1 && u <= 2048, "dgrp_model_create: units=%d not supported (1..2048)", u);
    DGRP_REQUIRE(C >= 2 && C <= DGRP_MAXC, "dgrp_model_create: classes=%d not supported (2..%d)", C, DGRP_MAXC);
    DGRP_REQUIRE(kernel && rec && bias && ffk && ffb && (!attention || scale), "dgrp_model_create: NULL tensor");
    char nm[8];
    int rc = dgrp_device_info(nm, sizeof(nm), nullptr, nullptr);
    if (rc != DGRP_OK) return rc;

    dgrp_model *m = new dgrp_model();
    m->T = T; m->u = u; m->C = C; m->attention = attention ? 1 : 0;
    m->cell = 0;
    m->ref_only = 0; m->is_view = 0;
    if (u > 256 || C > 16) return create_ref_only(out, m, kernel, rec, bias, 2 * 3 * (int64_t)u, ffk, (int64_t)(attention ? 2 : 1) * u * C, ffb,
                                        attention ? scale : nullptr);
    m->UP = (u + 31) / 32 * 32;
    m->NW = m->UP / 32;
    m->KS = m->UP / 16;
    m->nfrag = 3 * (m->KS + 1) + 3;
    m->d_pack = nullptr; m->d_ffb = nullptr; m->d_scale = nullptr; m->d_wtop = nullptr; m->d_raw = nullptr;
    m->d_pack_lo = nullptr; m->precision = 0; m->d_pack16 = nullptr; m->d_xtab = nullptr; m->d_stream = nullptr;
    m->d_packw = nullptr; m->d_xtabw = nullptr; m->NU16 = 0;
    const int KS = m->KS, NF = m->nfrag, u3 = 3 * u;
    // The update z*h + (1-z)*tanh(g) can be written with ONE reciprocal, of (1 + 2^az)(1 + 2^ag), if that
    // product cannot overflow: |h| <= 1 and 0 < r < 1 bound both pre-activations by the weights' absolute
    // column sums.  Checked here once; models that fail it (or DGRP_GRU_SAFE=1) take the two-reciprocal kernel.
    {
        double worst = 0.0;
        for (int j = 0; j < u; ++j) {
            double bz = 1.0, bg = 0.0;                                        // 1.0: the z gate's "+1" fold (below)
            double kz = 0.0, kg = 0.0;
            for (int c = 0; c < 5; ++c) {
                kz = std::max(kz, (double)fabsf(kernel[(size_t)c * u3 + j]));
                kg = std::max(kg, (double)fabsf(kernel[(size_t)c * u3 + 2 * u + j]));
            }
            double sz = kz + fabs((double)bias[j] + (double)bias[u3 + j]);
            double sg = kg + fabsf(bias[2 * u + j]) + fabsf(bias[u3 + 2 * u + j]);
            for (int k = 0; k < u; ++k) {
                sz += fabsf(rec[(size_t)k * u3 + j]);
                sg += fabsf(rec[(size_t)k * u3 + 2 * u + j]);
            }
            bz += 1.4426950408889634 * sz;
            bg += 2.8853900817779268 * sg;
            if (!(bz + bg <= worst)) worst = bz + bg;                         // NaN weights -> not provable
        }
        const char *safe = getenv("DGRP_GRU_SAFE");
        m->onercp = (m->NW <= 4 && worst <= 120.0 && !(safe && safe[0] == '1')) ? 1 : 0;
    }

    std::vector<uint16_t> pack((size_t)m->NW * NF * 64 * 8, 0);
    frag_writer fw{ pack, NF };
    for (int w = 0; w < m->NW; ++w) {
        for (int l = 0; l < 64; ++l) {
            const int unit = 32 * w + (l & 31);
            const bool uok = unit < u;
            for (int g = 0; g < 3; ++g) {
                // exp2-domain scale folded into the weights: sigmoid(x) = 1/(1 + 2^(-x log2 e)) for z and r,
                // tanh(x) = 1 - 2/(1 + 2^(2 x log2 e)) for the candidate (gru_kernel.hip)
                const float gs = g < 2 ? -1.4426950408889634f : 2.8853900817779268f;
                pack_recurrent(fw, w, l, g * (KS + 1), KS, rec, u3, g * u + unit, u, uok, gs);
                // input part (k-step KS): rows 0-4 kernel, 5 bias.  z and r carry the whole input projection and both
                // biases; the h gate's fragment only the RECURRENT bias (it sits inside r * (...)), its input
                // projection is the Bxh fragment.
                float v8[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
                if (uok) {
                    for (int j = 0; j < 5; ++j) v8[j] = g < 2 ? gs * kernel[(size_t)j * u3 + g * u + unit] : 0.0f;
                    v8[5] = gs * (g < 2 ? (float)((double)bias[g * u + unit] + (double)bias[u3 + g * u + unit]) : bias[u3 + 2 * u + unit]);
                    if (m->onercp && g == 0) v8[5] += 1.0f;                    // one-reciprocal blend wants 2 * 2^az
                }
                pack_input(fw, w, l, g * (KS + 1) + KS, v8);
            }
            float x8[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };                          // Bxh
            if (uok) {
                for (int j = 0; j < 5; ++j) x8[j] = 2.8853900817779268f * kernel[(size_t)j * u3 + 2 * u + unit];
                x8[5] = 2.8853900817779268f * bias[2 * u + unit];
            }
            pack_input(fw, w, l, 3 * (KS + 1), x8);
            pack_dense(fw, w, l, 3 * (KS + 1) + 1, 3 * (KS + 1) + 2, ffk, attention ? u : 0, u, C);
        }
    }
    float ffb16[16] = { 0 };
    for (int c = 0; c < C; ++c) ffb16[c] = ffb[c];

#define CREATE_HIP(call)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            dgrp_set_error("%s failed: %s", #call, hipGetErrorString(e_));                 \
            dgrp_model_destroy(m);                                                         \
            return DGRP_EHIP;                                                              \
        }                                                                                  \
    } while (0)
    CREATE_HIP(hipMalloc((void **)&m->d_pack, pack.size() * 2));
    CREATE_HIP(hipMemcpy(m->d_pack, pack.data(), pack.size() * 2, hipMemcpyHostToDevice));
    CREATE_HIP(hipMalloc((void **)&m->d_ffb, sizeof(ffb16)));
    CREATE_HIP(hipMemcpy(m->d_ffb, ffb16, sizeof(ffb16), hipMemcpyHostToDevice));
    if (attention) {
        std::vector<float> sc(m->UP, 0.0f), wtop((size_t)m->UP * 16, 0.0f);
        for (int k = 0; k < u; ++k) {
            sc[k] = scale[k];
            for (int c = 0; c < C; ++c) wtop[(size_t)k * 16 + c] = ffk[(size_t)k * C + c];
        }
        CREATE_HIP(hipMalloc((void **)&m->d_scale, sc.size() * 4));
        CREATE_HIP(hipMemcpy(m->d_scale, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
        CREATE_HIP(hipMalloc((void **)&m->d_wtop, wtop.size() * 4));
        CREATE_HIP(hipMemcpy(m->d_wtop, wtop.data(), wtop.size() * 4, hipMemcpyHostToDevice));
    }
    CREATE_HIP(upload_raw(m, kernel, rec, bias, 2 * (int64_t)u3, ffk, (int64_t)(attention ? 2 : 1) * u * C, ffb,
                          attention ? scale : nullptr));
    if (m->NW <= 4) {
        // lo halves of the recurrent fragments for the split-operand kernel: what fp16 rounding dropped from the
        // scaled weight, in the kernel's consumption order: k-step major, gates r, g, z
        std::vector<uint16_t> lo((size_t)m->NW * 3 * KS * 64 * 8, 0);
        static const int order[3] = { 1, 2, 0 };
        for (int w = 0; w < m->NW; ++w)
            for (int ks = 0; ks < KS; ++ks)
                for (int gi = 0; gi < 3; ++gi) {
                    const int g = order[gi];
                    const float gs = g < 2 ? -1.4426950408889634f : 2.8853900817779268f;
                    for (int l = 0; l < 64; ++l) {
                        const int unit = 32 * w + (l & 31);
                        for (int j = 0; j < 8; ++j) {
                            const int k = 16 * ks + 8 * (l >> 5) + j;
                            uint16_t v = 0;
                            if (unit < u && k < u) {
                                const float x = gs * rec[(size_t)k * u3 + g * u + unit];
                                v = f2h(x - h2f(f2h(x)));
                            }
                            lo[((((size_t)w * KS + ks) * 3 + gi) * 64 + l) * 8 + j] = v;
                        }
                    }
            }
        CREATE_HIP(hipMalloc((void **)&m->d_pack_lo, lo.size() * 2));
        CREATE_HIP(hipMemcpy(m->d_pack_lo, lo.data(), lo.size() * 2, hipMemcpyHostToDevice));
        if (m->NW == 4) {
            // gru_split2_kernel's operands (gru_split2.hip).  A fragment of v_mfma_f32_16x16x32_f16: lane l, element j <-> A[row l & 15][k 8 (l >> 4) + j];
            // here row = unit 32 w + 16 uh + (l & 15), k = recurrent row 32 ks + 8 (l >> 4) + j.  Same scaled values, same hi/lo split as above.
            static const int order16[3] = { 1, 2, 0 };                                  // r, g, z in Keras column blocks [z | r | h]
            std::vector<uint16_t> p16((size_t)4 * 48 * 64 * 8, 0);
            for (int w = 0; w < 4; ++w)
                for (int gi = 0; gi < 3; ++gi)
                    for (int ks = 0; ks < 4; ++ks)
                        for (int uh = 0; uh < 2; ++uh)
                            for (int l = 0; l < 64; ++l)
                                for (int j = 0; j < 8; ++j) {
                                    const int g = order16[gi], unit = 32 * w + 16 * uh + (l & 15), k = 32 * ks + 8 * (l >> 4) + j;
                                    uint16_t hi = 0, lo16 = 0;
                                    if (unit < u && k < u) {
                                        const float gs = g < 2 ? -1.4426950408889634f : 2.8853900817779268f;
                                        const float x = gs * rec[(size_t)k * u3 + g * u + unit];
                                        hi = f2h(x);
                                        lo16 = f2h(x - h2f(hi));
                                    }
                                    const size_t f = (size_t)gi * 8 + ks * 2 + uh;
                                    p16[(((size_t)w * 48 + f) * 64 + l) * 8 + j] = hi;
                                    p16[(((size_t)w * 48 + 24 + f) * 64 + l) * 8 + j] = lo16;
                                }
            // input projection per base (one-hot input => a row lookup): kinds r, g (recurrent bias, inside r * (...)), z, x
            std::vector<float> xt((size_t)5 * 4 * 128, 0.0f);
            const double cs = -1.4426950408889634, ch = 2.8853900817779268;
            for (int b = 0; b < 5; ++b)
                for (int unit = 0; unit < u; ++unit) {
                    float *row = &xt[(size_t)b * 512];
                    row[0 * 128 + unit] = (float)(cs * ((double)kernel[(size_t)b * u3 + u + unit] + (double)bias[u + unit] + (double)bias[u3 + u + unit]));
                    row[1 * 128 + unit] = (float)(ch * (double)bias[u3 + 2 * u + unit]);
                    row[2 * 128 + unit] = (float)(cs * ((double)kernel[(size_t)b * u3 + unit] + (double)bias[unit] + (double)bias[u3 + unit]) + (m->onercp ? 1.0 : 0.0));
                    row[3 * 128 + unit] = (float)(ch * ((double)kernel[(size_t)b * u3 + 2 * u + unit] + (double)bias[2 * u + unit]));
                }
            CREATE_HIP(hipMalloc((void **)&m->d_pack16, p16.size() * 2));
            CREATE_HIP(hipMemcpy(m->d_pack16, p16.data(), p16.size() * 2, hipMemcpyHostToDevice));
            CREATE_HIP(hipMalloc((void **)&m->d_xtab, xt.size() * 4));
            CREATE_HIP(hipMemcpy(m->d_xtab, xt.data(), xt.size() * 4, hipMemcpyHostToDevice));
        }
        if (u <= 64) {
            // gru_wave_kernel's operands (gru_wave.hip): every wave holds ALL units.  A fragment of v_mfma_f32_16x16x32_f16: lane l, element j
            // <-> A[row l & 15][k 8 (l >> 4) + j]; here row = unit 16 ug + (l & 15), k = tile column 32 ks + 8 (l >> 4) + j.  Dense B fragment:
            // lane l, element j <-> B[k 8 (l >> 4) + j][column l & 15] = 0.5 * FF kernel[row0 + unit of tile column 32 ks + k][class].  Same scaled values and
            // hi/lo split as the other split-operand kernels.
            const int NU = (u + 15) / 16, KSw = (NU + 1) / 2, UP16 = 16 * NU, NFw = 6 * KSw * NU;
            // column c of the kernel's hidden tile holds unit wave_col(c): inside every 32 columns the 4 units a lane owns of unit group
            // 2 j and of group 2 j + 1 alternate (the kernel publishes them with one 16-byte store)
            auto wave_col = [](int c) { const int j = c / 32, q = (c % 32) / 8, sub = c % 8; return 16 * (2 * j + (sub >= 4 ? 1 : 0)) + 4 * q + (sub & 3); };
            static const int orderw[3] = { 1, 2, 0 };                                   // r, g, z in Keras column blocks [z | r | h]
            std::vector<uint16_t> pw((size_t)(NFw + 2 * KSw) * 64 * 8, 0);
            for (int gi = 0; gi < 3; ++gi)
                for (int ks = 0; ks < KSw; ++ks)
                    for (int ug = 0; ug < NU; ++ug)
                        for (int l = 0; l < 64; ++l)
                            for (int j = 0; j < 8; ++j) {
                                const int g = orderw[gi], unit = 16 * ug + (l & 15), k = wave_col(32 * ks + 8 * (l >> 4) + j);
                                if (unit >= u || k >= u) continue;
                                const float gs = g < 2 ? -1.4426950408889634f : 2.8853900817779268f;
                                const float x = gs * rec[(size_t)k * u3 + g * u + unit];
                                const uint16_t hi = f2h(x);
                                pw[((size_t)((gi * KSw + ks) * NU + ug) * 64 + l) * 8 + j] = hi;
                                pw[((size_t)(((3 + gi) * KSw + ks) * NU + ug) * 64 + l) * 8 + j] = f2h(x - h2f(hi));
                            }
            const int drow0 = attention ? u : 0;
            for (int ks = 0; ks < KSw; ++ks)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int k = wave_col(32 * ks + 8 * (l >> 4) + j), c = l & 15;
                        if (k >= u || c >= C) continue;
                        const float v = 0.5f * ffk[(size_t)(drow0 + k) * C + c];
                        const uint16_t hi = f2h(v);
                        pw[((size_t)(NFw + 2 * ks) * 64 + l) * 8 + j] = hi;
                        pw[((size_t)(NFw + 2 * ks + 1) * 64 + l) * 8 + j] = f2h(v - h2f(hi));
                    }
            std::vector<float> xt((size_t)5 * 4 * UP16, 0.0f);
            const double cs = -1.4426950408889634, ch = 2.8853900817779268;
            for (int b = 0; b < 5; ++b)
                for (int unit = 0; unit < u; ++unit) {
                    float *row = &xt[(size_t)b * 4 * UP16];
                    row[0 * UP16 + unit] = (float)(cs * ((double)kernel[(size_t)b * u3 + u + unit] + (double)bias[u + unit] + (double)bias[u3 + u + unit]));
                    row[1 * UP16 + unit] = (float)(ch * (double)bias[u3 + 2 * u + unit]);
                    row[2 * UP16 + unit] = (float)(cs * ((double)kernel[(size_t)b * u3 + unit] + (double)bias[unit] + (double)bias[u3 + unit]) + (m->onercp ? 1.0 : 0.0));
                    row[3 * UP16 + unit] = (float)(ch * ((double)kernel[(size_t)b * u3 + 2 * u + unit] + (double)bias[2 * u + unit]));
                }
            CREATE_HIP(hipMalloc((void **)&m->d_packw, pw.size() * 2));
            CREATE_HIP(hipMemcpy(m->d_packw, pw.data(), pw.size() * 2, hipMemcpyHostToDevice));
            CREATE_HIP(hipMalloc((void **)&m->d_xtabw, xt.size() * 4));
            CREATE_HIP(hipMemcpy(m->d_xtabw, xt.data(), xt.size() * 4, hipMemcpyHostToDevice));
            m->NU16 = NU;
        }
        // Default for the models it covers: the split-operand kernel, the one that keeps every base within the 1e-3 of
        // the north star whatever the model's conditioning.  dgrp_model_set_precision(m, 0) or
        // DGRP_GRU_PRECISION=0 selects the 2.5x faster fp16-operand kernel.
        const char *pe = getenv("DGRP_GRU_PRECISION");
        m->precision = !(pe && pe[0] == '0') ? 1 : 0;
    }
    if (m->NW > 4) {
        // 129-256 units: the streamed split-operand kernel (rnn_stream.hip) is the default, the fp16-operand kernel the --fast mode
        static const float gs3[3] = { -1.4426950408889634f, -1.4426950408889634f, 2.8853900817779268f };
        CREATE_HIP(upload_stream(m, rec, 3, gs3));
        {
            // gru_stream64_kernel's input-projection table (rnn_stream.hip): [5 bases][4 kinds r, g (recurrent bias only), z, x][UP units]
            // fp32, exp2 domain, both biases folded (the two-reciprocal gate chain: no "+1" in the z rows)
            const int UPm = m->UP;
            std::vector<float> xt((size_t)5 * 4 * UPm, 0.0f);
            const double cs = -1.4426950408889634, ch = 2.8853900817779268;
            for (int b = 0; b < 5; ++b)
                for (int unit = 0; unit < u; ++unit) {
                    float *row = &xt[(size_t)b * 4 * UPm];
                    row[0 * UPm + unit] = (float)(cs * ((double)kernel[(size_t)b * u3 + u + unit] + (double)bias[u + unit] + (double)bias[u3 + u + unit]));
                    row[1 * UPm + unit] = (float)(ch * (double)bias[u3 + 2 * u + unit]);
                    row[2 * UPm + unit] = (float)(cs * ((double)kernel[(size_t)b * u3 + unit] + (double)bias[unit] + (double)bias[u3 + unit]));
                    row[3 * UPm + unit] = (float)(ch * ((double)kernel[(size_t)b * u3 + 2 * u + unit] + (double)bias[2 * u + unit]));
                }
            CREATE_HIP(hipMalloc((void **)&m->d_xtab, xt.size() * 4));
            CREATE_HIP(hipMemcpy(m->d_xtab, xt.data(), xt.size() * 4, hipMemcpyHostToDevice));
        }
        const char *pe = getenv("DGRP_GRU_PRECISION");
        m->precision = !(pe && pe[0] == '0') ? 1 : 0;
    }
#undef CREATE_HIP
    *out = m;
    return DGRP_OK;
}

DGRP_EXPORT int dgrp_model_create_lstm(dgrp_model **out, int T, int u, int C, const float *kernel, const float *rec,
                                       const float *bias, const float *ffk, const float *ffb)
{
    DGRP_REQUIRE(out, "dgrp_model_create_lstm: NULL out");
    *out = nullptr;
    DGRP_REQUIRE(T >= 1 && T <= 65535, "dgrp_model_create_lstm: window size %d out of range", T);
    DGRP_REQUIRE(u >= 1 && u <= 2048, "dgrp_model_create_lstm: units=%d not supported (1..2048)", u);
    DGRP_REQUIRE(C >= 2 && C <= DGRP_MAXC, "dgrp_model_create_lstm: classes=%d not supported (2..%d)", C, DGRP_MAXC);
    DGRP_REQUIRE(kernel && rec && bias && ffk && ffb, "dgrp_model_create_lstm: NULL tensor");
    char nm[8];
    int rc = dgrp_device_info(nm, sizeof(nm), nullptr, nullptr);
    if (rc != DGRP_OK) return rc;
    dgrp_model *m = new dgrp_model();
    m->T = T; m->u = u; m->C = C; m->attention = 0; m->cell = 1;
    m->ref_only = 0; m->is_view = 0;
    if (u > 256 || C > 16) return create_ref_only(out, m, kernel, rec, bias, 4 * (int64_t)u, ffk, (int64_t)u * C, ffb, nullptr);
    m->UP = (u + 31) / 32 * 32;
    m->NW = m->UP / 32;
    m->KS = m->UP / 16;
    m->nfrag = 4 * (m->KS + 1) + 2;
    m->d_pack = nullptr; m->d_ffb = nullptr; m->d_scale = nullptr; m->d_wtop = nullptr; m->d_raw = nullptr;
    m->d_pack_lo = nullptr; m->precision = 0; m->d_pack16 = nullptr; m->d_xtab = nullptr; m->d_stream = nullptr;
    m->d_packw = nullptr; m->d_xtabw = nullptr; m->NU16 = 0;
    const int KS = m->KS, NF = m->nfrag, u4 = 4 * u;
    std::vector<uint16_t> pack((size_t)m->NW * NF * 64 * 8, 0);
    frag_writer fw{ pack, NF };
    for (int w = 0; w < m->NW; ++w)
        for (int l = 0; l < 64; ++l) {
            const int unit = 32 * w + (l & 31);
            const bool uok = unit < u;
            for (int g = 0; g < 4; ++g) {
                const float gs = g == 2 ? 2.8853900817779268f : -1.4426950408889634f;     // c: tanh, i/f/o: sigmoid
                pack_recurrent(fw, w, l, g * (KS + 1), KS, rec, u4, g * u + unit, u, uok, gs);
                float v8[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
                if (uok) {
                    for (int j = 0; j < 5; ++j) v8[j] = gs * kernel[(size_t)j * u4 + g * u + unit];
                    v8[5] = gs * bias[g * u + unit];
                }
                pack_input(fw, w, l, g * (KS + 1) + KS, v8);
            }
            pack_dense(fw, w, l, 4 * (KS + 1), 4 * (KS + 1) + 1, ffk, 0, u, C);
        }
    float ffb16[16] = { 0 };
    for (int c = 0; c < C; ++c) ffb16[c] = ffb[c];
    hipError_t e = hipMalloc((void **)&m->d_pack, pack.size() * 2);
    if (e == hipSuccess) e = hipMemcpy(m->d_pack, pack.data(), pack.size() * 2, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_ffb, sizeof(ffb16));
    if (e == hipSuccess) e = hipMemcpy(m->d_ffb, ffb16, sizeof(ffb16), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = upload_raw(m, kernel, rec, bias, u4, ffk, (int64_t)u * C, ffb, nullptr);
    if (e == hipSuccess) {
        static const float gs4[4] = { -1.4426950408889634f, -1.4426950408889634f, 2.8853900817779268f, -1.4426950408889634f };
        e = upload_stream(m, rec, 4, gs4);
        const char *pe = getenv("DGRP_GRU_PRECISION");
        m->precision = !(pe && pe[0] == '0') ? 1 : 0;          // split operands by default, like every other model
    }
    if (e != hipSuccess) {
        dgrp_set_error("dgrp_model_create_lstm: %s", hipGetErrorString(e));
        dgrp_model_destroy(m);
        return DGRP_EHIP;
    }
    *out = m;
    return DGRP_OK;
}

DGRP_EXPORT int dgrp_model_view(const dgrp_model *m, int level, dgrp_model **out)
{
    DGRP_REQUIRE(m && out, "dgrp_model_view: NULL argument");
    *out = nullptr;
    dgrp_model *v = new dgrp_model(*m);
    v->is_view = 1;
    const int rc = dgrp_model_set_precision(v, level);
    if (rc != DGRP_OK) { delete v; return rc; }
    *out = v;
    return DGRP_OK;
}

DGRP_EXPORT int dgrp_model_destroy(dgrp_model *m)
{
    if (!m) return DGRP_OK;
    if (m->is_view) { delete m; return DGRP_OK; }
    if (m->d_pack) (void)hipFree(m->d_pack);
    if (m->d_ffb) (void)hipFree(m->d_ffb);
    if (m->d_scale) (void)hipFree(m->d_scale);
    if (m->d_wtop) (void)hipFree(m->d_wtop);
    if (m->d_raw) (void)hipFree(m->d_raw);
    if (m->d_pack_lo) (void)hipFree(m->d_pack_lo);
    if (m->d_pack16) (void)hipFree(m->d_pack16);
    if (m->d_xtab) (void)hipFree(m->d_xtab);
    if (m->d_stream) (void)hipFree(m->d_stream);
    if (m->d_packw) (void)hipFree(m->d_packw);
    if (m->d_xtabw) (void)hipFree(m->d_xtabw);
    delete m;
    return DGRP_OK;
}

DGRP_EXPORT int dgrp_model_dims(const dgrp_model *m, int *T, int *u, int *C, int *attention)
{
    DGRP_REQUIRE(m, "dgrp_model_dims: NULL model");
    if (T) *T = m->T;
    if (u) *u = m->u;
    if (C) *C = m->C;
    if (attention) *attention = m->attention;
    return DGRP_OK;
}

DGRP_EXPORT int dgrp_model_flags(const dgrp_model *m)
{
    DGRP_REQUIRE(m, "dgrp_model_flags: NULL model");
    return (m->onercp ? 1 : 0) | (m->precision == 1 ? 2 : 0) | (m->ref_only ? 4 : 0);
}

DGRP_EXPORT int dgrp_model_set_precision(dgrp_model *m, int level)
{
    DGRP_REQUIRE(m, "dgrp_model_set_precision: NULL model");
    DGRP_REQUIRE(level == 0 || level == 1, "dgrp_model_set_precision: level must be 0 (fp16 operands) or 1 (split operands)");
    if (m->ref_only) return DGRP_OK;                          // one kernel set: plain fp32 whatever the level
    DGRP_REQUIRE(level == 0 || m->d_pack_lo || m->d_stream, "dgrp_model_set_precision: this model has no split-operand kernel");
    m->precision = level;
    return DGRP_OK;
}

// attention keeps avg[t] ([nw,T,UP]: fp32 behind the split-operand pre-pass, fp16 behind the fp16-operand one -- sized for fp32, the
// level can change between the query and the call) and the avg half of the logits (fp32 [nw,T,C]) between kernels
// fp32 path: windows per pass of the plain-fp32 kernels (their h_t of both strands is 2 T u floats per window), and the carve
static int64_t ref_sub_windows(const dgrp_model *m)
{
    const int64_t per = 2 * (int64_t)m->T * m->u * 4 + (int64_t)m->T * (m->C + 1) * 4 + 2 * (int64_t)m->u * 4;
    return std::max<int64_t>(16, std::min<int64_t>(4096, (512ll << 20) / per) / 16 * 16);
}

DGRP_EXPORT int64_t dgrp_forward_workspace_bytes(const dgrp_model *m, int64_t nw)
{
    if (!m || nw < 0) return 0;
    if (m->ref_only) {
        const int64_t sub = std::min<int64_t>(ref_sub_windows(m), std::max<int64_t>(nw, 1));
        return dgrp_align_up(dgrp_forward_reference_workspace_bytes(m, sub), 256) + dgrp_align_up(sub * m->T * (int64_t)m->C * 4, 256);
    }
    if (!m->attention) return 256;
    return dgrp_align_up(nw * m->T * (int64_t)m->UP * 4, 256) + dgrp_align_up(nw * m->T * (int64_t)m->C * 4, 256);
}

// the forward entry points of a ref_only model: plain-fp32 kernels a few thousand windows at a time; merged output = the reference's own
// loop (deepgrp/prediction.py:104-110) on runs of windows whose rows are equally spaced (the short last batch sits elsewhere: SURVEY Q2)
static int forward_ref(const dgrp_model *m, const uint8_t *d_idx, int64_t n, int64_t s, dgrp_placement place, int64_t w0, int64_t nw,
                       int merge, float *d_out, void *d_work, int64_t work_bytes, hipStream_t stream)
{
    if (work_bytes < dgrp_forward_workspace_bytes(m, nw) || !d_work) {
        dgrp_set_error("dgrp_forward: this model runs on the fp32 kernels and needs %lld bytes of workspace for %lld windows",
                       (long long)dgrp_forward_workspace_bytes(m, nw), (long long)nw);
        return DGRP_ENOMEM;
    }
    const int64_t sub = std::min<int64_t>(ref_sub_windows(m), nw);
    const int64_t ref_bytes = dgrp_align_up(dgrp_forward_reference_workspace_bytes(m, sub), 256);
    float *probs = (float *)((char *)d_work + ref_bytes);
    for (int64_t a = w0; a < w0 + nw; a += sub) {
        const int64_t k = std::min<int64_t>(sub, w0 + nw - a);
        float *dst = merge ? probs : d_out + (a - w0) * (int64_t)m->T * m->C;
        int rc = dgrp_forward_windows_reference(m, d_idx, n, s, a, k, dst, d_work, ref_bytes, stream);
        if (rc) return rc;
        if (!merge) continue;
        for (int64_t b = a; b < a + k;) {                             // runs inside one placement regime
            const int64_t e = (b < place.nfullB && a + k > place.nfullB) ? place.nfullB : a + k;
            const int64_t row = dgrp_place_row(place, b, s);
            if (row < n) {
                rc = dgrp_get_max(d_out + row * m->C, n - row, probs + (b - a) * (int64_t)m->T * m->C, m->T, m->C, s, e - b, stream);
                if (rc) return rc;
            }
            b = e;
        }
    }
    return DGRP_OK;
}

static int forward_common(const dgrp_model *m, const uint8_t *d_idx, int64_t n, int64_t s, int64_t batch, int64_t w0,
                          int64_t nw, int merge, float *d_out, void *d_work, int64_t work_bytes, hipStream_t stream)
{
    DGRP_REQUIRE(m && s >= 1 && w0 >= 0 && nw >= 0 && n >= 0, "dgrp_forward: bad arguments");
    if (nw == 0) return DGRP_OK;
    DGRP_REQUIRE(d_idx && d_out, "dgrp_forward: NULL pointer");
    DGRP_REQUIRE((w0 + nw - 1) * s + m->T <= n, "dgrp_forward: window %lld (start %lld) runs past n=%lld",
                 (long long)(w0 + nw - 1), (long long)((w0 + nw - 1) * s), (long long)n);
    dgrp_placement place = { 0, 0 };
    if (merge) {
        DGRP_REQUIRE(batch >= 1, "dgrp_forward_merge: batch must be >= 1");
        const int64_t total = dgrp_window_count(n, m->T, s);
        DGRP_REQUIRE(w0 + nw <= total, "dgrp_forward_merge: windows %lld..%lld exceed the record's %lld",
                     (long long)w0, (long long)(w0 + nw), (long long)total);
        place = dgrp_make_placement(total, batch);
    }
    if (m->ref_only) return forward_ref(m, d_idx, n, s, place, w0, nw, merge, d_out, d_work, work_bytes, stream);
    if (!m->attention) return dgrp_gru_launch(m, d_idx, n, s, place, w0, nw, merge ? 0 : 1, d_out, nullptr, stream);
    if (work_bytes < dgrp_forward_workspace_bytes(m, nw) || !d_work) {
        dgrp_set_error("dgrp_forward: attention model needs %lld bytes of workspace for %lld windows",
                       (long long)dgrp_forward_workspace_bytes(m, nw), (long long)nw);
        return DGRP_ENOMEM;
    }
    void *avg = d_work;
    float *pl = (float *)((char *)d_work + dgrp_align_up(nw * m->T * (int64_t)m->UP * 4, 256));
    int rc = dgrp_gru_launch(m, d_idx, n, s, place, w0, nw, 2, pl, avg, stream);
    if (rc) return rc;
    return dgrp_attention_launch(m, s, place, w0, nw, merge, n, avg, pl, d_out, stream);
}

DGRP_EXPORT int dgrp_forward_windows(const dgrp_model *m, const uint8_t *d_idx, int64_t n, int64_t s, int64_t w0,
                                     int64_t nw, float *d_probs, void *d_work, int64_t work_bytes, void *stream)
{
    return forward_common(m, d_idx, n, s, 1, w0, nw, 0, d_probs, d_work, work_bytes, (hipStream_t)stream);
}

DGRP_EXPORT int dgrp_forward_merge(const dgrp_model *m, const uint8_t *d_idx, int64_t n, int64_t s, int64_t batch,
                                   int64_t w0, int64_t nw, float *d_out, void *d_work, int64_t work_bytes, void *stream)
{
    return forward_common(m, d_idx, n, s, batch, w0, nw, 1, d_out, d_work, work_bytes, (hipStream_t)stream);
}

// ---- the whole per-record chain of deepgrp/__main__.py:46-83 + :288-292 behind one call ---------------------
// (A3-A11: windows -> forward -> max-merge -> scores / softmax -> MSS labels -> segments).  Everything lives in the
// caller's workspace; a host thread per stream can run records concurrently with nothing but this call in between.
// Bytes of avg[t] spill one launch of an attention model may use.  The spill is the only reason to cut a record's windows into
// several launches, and a launch that is not a whole number of rounds of workgroups ends with a partly filled one (the one-tile
// kernels keep 4-8 workgroups of 16 windows on a CU: 16 384-32 768 windows per round; the 4 GiB this cap used to be gave the
// reference's default model 2.75 rounds per launch -- three rounds of time, 17 % of the pre-pass).  So the cap follows the card:
// 1/32 of its memory, at most 8 GiB (the command line runs up to 16 records at once, each with a spill of its own);
// DGRP_SPILL_BYTES overrides it.
static int64_t spill_cap_bytes()
{
    static const int64_t cap = [] {
        if (const char *e = getenv("DGRP_SPILL_BYTES")) {
            const long long v = atoll(e);
            if (v >= (1ll << 20)) return (int64_t)v;
        }
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) return (int64_t)(4ll << 30);
        return std::min<int64_t>(8ll << 30, std::max<int64_t>(1ll << 30, (int64_t)(total_b / 32)));
    }();
    return cap;
}

static int64_t record_window_chunk(const dgrp_model *m)
{
    if (m->ref_only) return ref_sub_windows(m);
    // no spill bounds a launch: 8 M windows at a time (a 250 Mbp chromosome in ONE launch: 5 launches of 2^20 windows each ended in a
    // round of workgroups that filled a fraction of the chip: +0.8 % on the benchmark record)
    if (!m->attention) return 1ll << 23;
    // whole rounds of workgroups where possible: 32 768 windows = 256 CUs x 8 workgroups x 16 windows is a whole number of rounds
    // for every recurrent kernel (8, 4, 2 or 1 workgroups of 16 windows, or one of 32, per CU); below that 4096 = 256 CUs x 16
    // (a launch of 4112 windows costs a large model two rounds for the work of one)
    const int64_t per = (int64_t)m->T * ((int64_t)m->UP * 4 + (int64_t)m->C * 4);
    int64_t c = spill_cap_bytes() / per;
    c = c >= 32768 ? c / 32768 * 32768 : c >= 4096 ? c / 4096 * 4096 : c / 16 * 16;
    if (c < 16) c = 16;
    return c < (1ll << 20) ? c : (1ll << 20);
}

DGRP_EXPORT int64_t dgrp_forward_window_chunk(const dgrp_model *m)
{
    return m ? record_window_chunk(m) : 0;
}

// ---- lanes: the chunks of an attention model's record alternate between DGRP_LANES internal streams ------------------------------
// The recurrent pre-pass is bound by the matrix cores and the vector unit and moves little memory; the second kernel is bound by HBM
// and leaves the CUs it sits on mostly waiting.  Chunk after chunk on ONE stream the two never meet; on lanes the second kernel of
// one chunk runs beside the pre-pass of the next (r03, tools/two_stream_probe.py: the reference's default model +4.5 %, its
// hyper-parameter space +6.5 %, 128 units and more +1 %: those stay on one stream).  The merged output is a max: the order in
// which chunks land does not matter, bit for bit.  Each lane has a spill of its own, a third of the one-stream chunk.
#define DGRP_LANES 3
static int64_t lane_window_chunk(const dgrp_model *m)            // 0: one stream
{
    if (m->ref_only || !m->attention) return 0;
    if (const char *e = getenv("DGRP_LANE_CHUNK")) {            // tests: lanes on tiny records
        const long long v = atoll(e);
        return v >= 16 ? (int64_t)(v / 16 * 16) : 0;
    }
    const int64_t c = record_window_chunk(m);
    if (c < 65536) return 0;
    const int64_t lc = c / DGRP_LANES / 32768 * 32768;
    return lc >= 32768 ? lc : 32768;
}
struct lane_pool { hipStream_t s[DGRP_LANES]; int dev; bool ok; };
static lane_pool *lanes_get()
{
    static std::mutex mu;
    static std::vector<lane_pool *> pools;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    for (lane_pool *q : pools)
        if (q->dev == dev) return q->ok ? q : nullptr;
    lane_pool *q = new lane_pool();
    q->dev = dev; q->ok = true;
    for (int i = 0; i < DGRP_LANES; ++i)
        if (hipStreamCreateWithFlags(&q->s[i], hipStreamNonBlocking) != hipSuccess) q->ok = false;
    pools.push_back(q);
    return q->ok ? q : nullptr;
}

// workspace of dgrp_forward_merge_record: one forward workspace per lane (or the one of a whole chunk)
static int64_t record_forward_bytes(const dgrp_model *m, int64_t nwin)
{
    const int64_t lc = lane_window_chunk(m);
    if (lc > 0 && nwin > lc)
        return DGRP_LANES * dgrp_align_up(std::max<int64_t>(dgrp_forward_workspace_bytes(m, std::min(lc, nwin)), 256), 256);
    const int64_t chunk = std::min<int64_t>(record_window_chunk(m), nwin > 0 ? nwin : 1);
    return std::max<int64_t>(dgrp_forward_workspace_bytes(m, chunk), 256);
}

DGRP_EXPORT int64_t dgrp_forward_merge_record_workspace_bytes(const dgrp_model *m, int64_t n, int64_t s)
{
    if (!m || n < 0 || s < 1) return 0;
    return record_forward_bytes(m, dgrp_window_count(n, m->T, s));
}

// prediction.py:89-111 for a whole record: every window, chunk by chunk, max-merged into d_out [n, C] (which the caller has zeroed)
DGRP_EXPORT int dgrp_forward_merge_record(const dgrp_model *m, const uint8_t *d_idx, int64_t n, int64_t s, int64_t batch, float *d_out,
                                          void *d_work, int64_t work_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    DGRP_REQUIRE(m && n >= 0 && s >= 1 && batch >= 1, "dgrp_forward_merge_record: bad arguments");
    const int64_t nwin = dgrp_window_count(n, m->T, s);
    if (nwin == 0) return DGRP_OK;
    DGRP_REQUIRE(d_idx && d_out && d_work, "dgrp_forward_merge_record: NULL pointer");
    const int64_t need = record_forward_bytes(m, nwin);
    if (work_bytes < need) {
        dgrp_set_error("dgrp_forward_merge_record: workspace %lld < %lld bytes", (long long)work_bytes, (long long)need);
        return DGRP_ENOMEM;
    }
    const int64_t lc = lane_window_chunk(m);
    lane_pool *lp = lc > 0 && nwin > lc ? lanes_get() : nullptr;
    if (!lp) {
        const int64_t chunk = lc > 0 && nwin > lc ? lc : record_window_chunk(m);      // (no lanes to be had: their chunk, one stream)
        for (int64_t w0 = 0; w0 < nwin; w0 += chunk) {
            const int64_t nw = std::min<int64_t>(chunk, nwin - w0);
            const int rc = dgrp_forward_merge(m, d_idx, n, s, batch, w0, nw, d_out, d_work, work_bytes, stream);
            if (rc) return rc;
        }
        return DGRP_OK;
    }
    const int64_t per = need / DGRP_LANES;
    hipEvent_t fork = nullptr, join[DGRP_LANES] = { nullptr, nullptr, nullptr };
    int rc = DGRP_OK;
    auto bail = [&](hipError_t e) { if (e != hipSuccess && rc == DGRP_OK) { dgrp_set_error("dgrp_forward_merge_record: %s", hipGetErrorString(e)); rc = DGRP_EHIP; } };
    bail(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (int i = 0; i < DGRP_LANES; ++i) bail(hipEventCreateWithFlags(&join[i], hipEventDisableTiming));
    if (rc == DGRP_OK) {
        bail(hipEventRecord(fork, stream));                       // the lanes start behind what the caller's stream holds (the zeroed output)
        for (int i = 0; i < DGRP_LANES; ++i) bail(hipStreamWaitEvent(lp->s[i], fork, 0));
        int64_t k = 0;
        for (int64_t w0 = 0; w0 < nwin && rc == DGRP_OK; w0 += lc, ++k) {
            const int64_t nw = std::min<int64_t>(lc, nwin - w0);
            const int lane = (int)(k % DGRP_LANES);
            rc = dgrp_forward_merge(m, d_idx, n, s, batch, w0, nw, d_out, (char *)d_work + lane * per, per, lp->s[lane]);
        }
        // join: the caller's stream goes on behind every lane (also after an error: nothing may still run when the caller frees)
        for (int i = 0; i < DGRP_LANES; ++i) {
            bail(hipEventRecord(join[i], lp->s[i]));
            bail(hipStreamWaitEvent(stream, join[i], 0));
        }
    }
    if (fork) (void)hipEventDestroy(fork);
    for (int i = 0; i < DGRP_LANES; ++i)
        if (join[i]) (void)hipEventDestroy(join[i]);
    return rc;
}

struct record_layout {
    int64_t out, scores, cls, labels, count, fwd, post, bytes, post_bytes, fwd_bytes;
};

static record_layout record_carve(const dgrp_model *m, int64_t n, int64_t s, int use_mss)
{
    record_layout l;
    int64_t p = 0;
    auto take = [&](int64_t b) { const int64_t q = p; p += dgrp_align_up(b, 256); return q; };
    const int64_t nwin = dgrp_window_count(n, m->T, s);
    l.out = take(n * m->C * 4);
    l.scores = take(use_mss ? n * 8 : 0);
    l.cls = take(use_mss ? n : 0);
    l.labels = take(n);
    l.count = take(8);
    l.fwd_bytes = record_forward_bytes(m, nwin);
    l.fwd = take(l.fwd_bytes);
    l.post_bytes = std::max<int64_t>(std::max<int64_t>(use_mss ? dgrp_mss_workspace_bytes(n) : 4096, dgrp_segments_workspace_bytes(n)), 4096);
    l.post = take(l.post_bytes);
    l.bytes = p;
    return l;
}

DGRP_EXPORT int64_t dgrp_record_workspace_bytes(const dgrp_model *m, int64_t n, int64_t s, int use_mss)
{
    if (!m || n < 0 || s < 1) return 0;
    return record_carve(m, n, s, use_mss).bytes;
}

DGRP_EXPORT int dgrp_predict_record(const dgrp_model *m, const uint8_t *d_idx, int64_t n, int64_t s, int64_t batch,
                                    int min_mss_len, int xdrop_len, int use_mss, int64_t offset, int32_t contig,
                                    dgrp_segment *d_records, int64_t cap, int64_t *h_count, void *d_work,
                                    int64_t work_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    DGRP_REQUIRE(m && n >= 0 && s >= 1 && batch >= 1 && cap >= 0 && h_count, "dgrp_predict_record: bad arguments");
    *h_count = 0;
    if (n == 0) return DGRP_OK;
    DGRP_REQUIRE(d_idx && d_work && (cap == 0 || d_records), "dgrp_predict_record: NULL pointer");
    const record_layout l = record_carve(m, n, s, use_mss);
    if (work_bytes < l.bytes) {
        dgrp_set_error("dgrp_predict_record: workspace %lld < %lld bytes", (long long)work_bytes, (long long)l.bytes);
        return DGRP_ENOMEM;
    }
    char *w = (char *)d_work;
    float *out = (float *)(w + l.out);
    int8_t *labels = (int8_t *)(w + l.labels);
    int64_t *d_count = (int64_t *)(w + l.count);
    DGRP_HIP(hipMemsetAsync(out, 0, (size_t)n * m->C * 4, stream));                    // np.zeros, prediction.py:103
    int rc = dgrp_forward_merge_record(m, d_idx, n, s, batch, out, w + l.fwd, l.fwd_bytes, stream);
    if (rc) return rc;
    if (use_mss) {
        rc = dgrp_scores(out, n, m->C, (double *)(w + l.scores), (int8_t *)(w + l.cls), stream);
        if (rc) return rc;
        rc = dgrp_mss_labels((const double *)(w + l.scores), (const int8_t *)(w + l.cls), n, m->C, min_mss_len, xdrop_len,
                             labels, nullptr, w + l.post, l.post_bytes, stream);
    } else {
        rc = dgrp_softmax_labels(out, n, m->C, nullptr, labels, w + l.post, l.post_bytes, stream);
    }
    if (rc) return rc;
    rc = dgrp_segments(labels, n, offset, contig, d_records, cap, d_count, w + l.post, l.post_bytes, stream);
    if (rc) return rc;
    DGRP_HIP(hipMemcpyAsync(h_count, d_count, 8, hipMemcpyDeviceToHost, stream));
    DGRP_HIP(hipStreamSynchronize(stream));
    return DGRP_OK;
}

// ---- the same for a BATCH of short records in a handful of launches (files of thousands of contigs) -------------
// (post_kernels.hip)
int dgrp_batch_marks(const int64_t *d_start, const int64_t *d_len, int64_t nrec, int64_t total_n, uint8_t *d_marks,
                     double *d_scores, int8_t *d_cls, hipStream_t stream);
int dgrp_batch_segments(const int8_t *d_labels, const uint8_t *d_marks, int64_t total_n, const int64_t *d_start, int64_t nrec,
                        const int64_t *d_startpos, const int32_t *d_contig, dgrp_segment *d_records, int64_t cap,
                        int64_t *d_count, void *d_work, int64_t work_bytes, hipStream_t stream);

struct batch_layout {
    int64_t out, scores, cls, labels, marks, count, recs, wgf, start, len, spos, contig, post, post_bytes, avg, pl, bytes;
};

static batch_layout batch_carve(const dgrp_model *m, int64_t nrec, int64_t total_rows, int64_t total_windows)
{
    batch_layout l;
    int64_t p = 0;
    auto take = [&](int64_t b) { const int64_t q = p; p += dgrp_align_up(b, 256); return q; };
    l.out = take(total_rows * m->C * 4);
    l.scores = take(total_rows * 8);
    l.cls = take(total_rows);
    l.labels = take(total_rows);
    l.marks = take(total_rows);
    l.count = take(8);
    l.recs = take(nrec * 64);
    l.wgf = take((nrec + 1) * 8);
    l.start = take((nrec + 1) * 8);
    l.len = take(nrec * 8);
    l.spos = take(nrec * 8);
    l.contig = take(nrec * 4);
    l.post_bytes = std::max<int64_t>(std::max<int64_t>(dgrp_mss_batch_workspace_bytes(total_rows, nrec), dgrp_segments_workspace_bytes(total_rows)), 4096);
    l.post = take(l.post_bytes);
    // attention: avg[t] of every window (fp16) and the avg half of the logits
    l.avg = take(m->attention ? total_windows * m->T * (int64_t)m->UP * 4 : 0);
    l.pl = take(m->attention ? total_windows * m->T * (int64_t)m->C * 4 : 0);
    l.bytes = p;
    return l;
}

static int64_t batch_windows(const dgrp_model *m, int64_t nrec, const int64_t *h_n, int64_t s)
{
    int64_t w = 0;
    for (int64_t r = 0; r < nrec; ++r) w += dgrp_window_count(h_n[r], m->T, s);
    return w;
}

static int64_t batch_rows(int64_t nrec, const int64_t *h_n)
{
    int64_t rows = 0;
    for (int64_t r = 0; r < nrec; ++r) rows += dgrp_align_up(h_n[r], 64);
    return rows;
}

DGRP_EXPORT int64_t dgrp_batch_workspace_bytes(const dgrp_model *m, int64_t nrec, const int64_t *h_n, int64_t s)
{
    if (!m || nrec < 0 || s < 1 || (nrec > 0 && !h_n)) return 0;
    for (int64_t r = 0; r < nrec; ++r)
        if (h_n[r] < 1) return 0;
    return batch_carve(m, nrec, batch_rows(nrec, h_n), batch_windows(m, nrec, h_n, s)).bytes;
}

DGRP_EXPORT int dgrp_predict_batch(const dgrp_model *m, const uint8_t *d_idx, int64_t nrec, const int64_t *h_idx_off,
                                   const int64_t *h_n, const int64_t *h_startpos, const int32_t *h_contig, int64_t s,
                                   int64_t batch, int min_mss_len, int xdrop_len, dgrp_segment *d_records, int64_t cap,
                                   int64_t *h_count, void *d_work, int64_t work_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    DGRP_REQUIRE(m && nrec >= 0 && s >= 1 && batch >= 1 && cap >= 0 && h_count, "dgrp_predict_batch: bad arguments");
    *h_count = 0;
    if (nrec == 0) return DGRP_OK;
    DGRP_REQUIRE(!m->ref_only && (m->cell == 0 || (m->cell == 1 && m->NW <= 8)), "dgrp_predict_batch: unsupported model (fp32 path: record by record)");
    DGRP_REQUIRE(d_idx && h_idx_off && h_n && h_startpos && h_contig && d_work && (cap == 0 || d_records), "dgrp_predict_batch: NULL pointer");
    for (int64_t r = 0; r < nrec; ++r)
        DGRP_REQUIRE(h_n[r] >= 1 && h_idx_off[r] >= 0, "dgrp_predict_batch: record %lld: empty records do not belong in a batch", (long long)r);
    const int64_t rows = batch_rows(nrec, h_n);
    DGRP_REQUIRE(rows < (1ll << 31), "dgrp_predict_batch: %lld rows in one batch (limit 2^31)", (long long)rows);
    const int64_t windows = batch_windows(m, nrec, h_n, s);
    DGRP_REQUIRE(windows < (1ll << 31), "dgrp_predict_batch: too many windows in one batch");
    const batch_layout l = batch_carve(m, nrec, rows, windows);
    if (work_bytes < l.bytes) {
        dgrp_set_error("dgrp_predict_batch: workspace %lld < %lld bytes", (long long)work_bytes, (long long)l.bytes);
        return DGRP_ENOMEM;
    }
    char *w = (char *)d_work;
    // ---- tables
    std::vector<int64_t> recs((size_t)nrec * 8, 0), wgf((size_t)nrec + 1, 0), start((size_t)nrec + 1, 0);
    int64_t wfirst = 0;
    for (int64_t r = 0; r < nrec; ++r) {
        const int64_t nwin = dgrp_window_count(h_n[r], m->T, s);
        const dgrp_placement pl = dgrp_make_placement(nwin, batch);
        int64_t *e = recs.data() + (size_t)r * 8;
        e[0] = h_idx_off[r]; e[1] = h_n[r]; e[2] = start[(size_t)r]; e[3] = nwin; e[4] = pl.nfullB; e[5] = pl.shift; e[6] = wfirst;
        wfirst += nwin;
        wgf[(size_t)r + 1] = wgf[(size_t)r] + (nwin + 15) / 16;
        start[(size_t)r + 1] = start[(size_t)r] + dgrp_align_up(h_n[r], 64);
    }
    DGRP_HIP(hipMemcpyAsync(w + l.recs, recs.data(), recs.size() * 8, hipMemcpyHostToDevice, stream));
    DGRP_HIP(hipMemcpyAsync(w + l.wgf, wgf.data(), wgf.size() * 8, hipMemcpyHostToDevice, stream));
    DGRP_HIP(hipMemcpyAsync(w + l.start, start.data(), start.size() * 8, hipMemcpyHostToDevice, stream));
    DGRP_HIP(hipMemcpyAsync(w + l.len, h_n, (size_t)nrec * 8, hipMemcpyHostToDevice, stream));
    DGRP_HIP(hipMemcpyAsync(w + l.spos, h_startpos, (size_t)nrec * 8, hipMemcpyHostToDevice, stream));
    DGRP_HIP(hipMemcpyAsync(w + l.contig, h_contig, (size_t)nrec * 4, hipMemcpyHostToDevice, stream));
    // ---- A3-A6 for all records: one launch
    float *out = (float *)(w + l.out);
    DGRP_HIP(hipMemsetAsync(out, 0, (size_t)rows * m->C * 4, stream));
    int rc;
    if (!m->attention) {
        rc = dgrp_gru_launch_batch(m, d_idx, s, w + l.recs, (const int64_t *)(w + l.wgf), nrec, wgf[(size_t)nrec], 0, out, nullptr, stream);
        if (rc) return rc;
    } else {
        // GRU pre-pass for all windows (avg[t] and the avg half of the logits, windows numbered through the batch),
        // then the attention kernel with the same record table for placement
        rc = dgrp_gru_launch_batch(m, d_idx, s, w + l.recs, (const int64_t *)(w + l.wgf), nrec, wgf[(size_t)nrec], 2,
                                   (float *)(w + l.pl), w + l.avg, stream);
        if (rc) return rc;
        rc = dgrp_attention_launch_recs(m, s, dgrp_placement{ 0, 0 }, 0, windows, 1, rows, w + l.avg, (const float *)(w + l.pl), out,
                                        w + l.recs, nrec, stream);
        if (rc) return rc;
    }
    // ---- A7: scores over all rows, then padding rows to (0.0, class 0) and the record marks
    double *scores = (double *)(w + l.scores);
    int8_t *cls = (int8_t *)(w + l.cls), *labels = (int8_t *)(w + l.labels);
    uint8_t *marks = (uint8_t *)(w + l.marks);
    rc = dgrp_scores(out, rows, m->C, scores, cls, stream);
    if (rc) return rc;
    rc = dgrp_batch_marks((const int64_t *)(w + l.start), (const int64_t *)(w + l.len), nrec, rows, marks, scores, cls, stream);
    if (rc) return rc;
    // ---- A9+A10 (synchronises: the host tables above are safe to drop afterwards)
    rc = dgrp_mss_labels_batch(scores, cls, rows, nrec, start.data(), m->C, min_mss_len, xdrop_len, labels, w + l.post, l.post_bytes, stream);
    if (rc) return rc;
    // ---- A11
    int64_t *d_count = (int64_t *)(w + l.count);
    rc = dgrp_batch_segments(labels, marks, rows, (const int64_t *)(w + l.start), nrec, (const int64_t *)(w + l.spos),
                             (const int32_t *)(w + l.contig), d_records, cap, d_count, w + l.post, l.post_bytes, stream);
    if (rc) return rc;
    DGRP_HIP(hipMemcpyAsync(h_count, d_count, 8, hipMemcpyDeviceToHost, stream));
    DGRP_HIP(hipStreamSynchronize(stream));
    return DGRP_OK;
}


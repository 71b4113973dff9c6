// Device-resident model: the Keras tensors of deepgrp/model.py:293-336 re-laid out as MFMA
// operand fragments (see gru_kernel.hip for the layout each fragment follows).
#pragma once
#include "dgrp_common.h"

struct dgrp_model {
    int T, u, C, attention;
    int cell;    // 0 = GRU (reset_after), 1 = LSTM
    int onercp;  // GRU, u <= 128: pre-activations provably small enough for the one-reciprocal blend (gru_kernel.hip)
    int UP;      // units padded to a multiple of 32
    int NW;      // waves per workgroup = UP / 32, one 32-unit column slice of every gate per wave
    int KS;      // 16-deep k-steps of the recurrent contraction = UP / 16
    int nfrag;   // fragments per wave: 3*(KS+1) gate fragments + x->h~ + dense hi + dense lo
    int precision;    // 0: fused fp16-operand kernel; 1: split-operand kernel where it applies (dgrp_model_set_precision)
    int is_view;      // dgrp_model_view: shares every device buffer of the model it was made from and frees none of them
    int ref_only;     // more units than any fused kernel takes (GRU > 256, LSTM > 256): every forward call runs the plain-fp32 kernels
                      // of ref_kernels.hip (the reference takes any `units`, deepgrp/model.py:117,219-229) -- slow, but not refused
    uint4 *d_pack;    // [NW][nfrag][64] 8 x fp16 per lane
    uint4 *d_pack_lo; // GRU, NW <= 4: [NW][KS][3][64] lo halves of the recurrent fragments (k-step major, gates r, g, z), or NULL
    // GRU, NW == 4 (97-128 units), for gru_split2_kernel (gru_split2.hip): the recurrent kernel as v_mfma_f32_16x16x32_f16 A fragments,
    // [4 waves][(pass*3 + gate)*8 + kstep*2 + unit-half][64] with pass hi|lo, gates r, g, z, k-steps of 32, unit halves of 16;
    // and the input projection as a table [5 bases][4 kinds r, g (recurrent bias only), z, x][128 units] fp32, exp2 domain
    uint4 *d_pack16;
    float *d_xtab;
    // GRU with 129-256 units and LSTM, for rnn_split_stream_kernel (rnn_stream.hip): the recurrent kernel as 32x32x16 fragments, hi and
    // lo halves, in consumption order [NW][KS][2 G][64] (k-step major; G hi fragments in pack gate order, then the G lo fragments), or NULL
    uint4 *d_stream;
    // GRU up to 64 units, for gru_wave_kernel (gru_wave.hip): NU16 = ceil(u / 16) unit groups, KS = ceil(NU16 / 2) k-steps of 32; the recurrent
    // kernel as 16x16x32 A fragments [((pass * 3 + gate) * KS + k-step) * NU16 + unit group][64] (pass hi|lo, gates r, g, z), then the Dense
    // kernel as B fragments [k-step][hi|lo][64]; the input projection as a table [5 bases][4 kinds][16 NU16 units] fp32, exp2 domain
    uint4 *d_packw;
    float *d_xtabw;
    int NU16;
    float *d_ffb;     // [16] dense bias, zero padded
    float *d_scale;   // [UP] attention scale (zero padded) or NULL
    float *d_wtop;    // [UP][16] rows of the dense kernel that multiply the context vector, or NULL
    // the tensors as given, fp32, for the full-precision yardstick (ref_kernels.hip); offsets in floats into d_raw
    float *d_raw;
    int64_t raw_kernel, raw_rec, raw_bias, raw_ffk, raw_ffb, raw_scale;
};

// dgrp_kernel_timer_* (api.hip): brackets a recurrent-kernel launch with HIP events while the calling thread has the timer enabled
struct dgrp_timer_scope {
    hipStream_t stream;
    int64_t windows;
    hipEvent_t start;
    bool on;
    dgrp_timer_scope(hipStream_t s, int64_t nw);
    ~dgrp_timer_scope();
};

// rows of LDS the fused kernel may use to pre-merge a workgroup's windows
int dgrp_gru_launch(const dgrp_model *m, const uint8_t *d_idx, int64_t n, int64_t s, dgrp_placement place,
                    int64_t w0, int64_t nw, int mode, float *d_out, void *d_avg, hipStream_t stream);
int dgrp_attention_launch(const dgrp_model *m, int64_t s, dgrp_placement place, int64_t w0, int64_t nw,
                          int merge, int64_t n, const void *d_avg, const float *d_pl, float *d_out,
                          hipStream_t stream);
// rnn_stream.hip: the streamed split-operand kernel (cell 0 = GRU with 5..8 waves, 1 = LSTM with 1..4 waves)
struct gru_params;
int dgrp_stream_launch(const gru_params &p, int cell, int NW, int64_t groups, size_t lds, hipStream_t stream);
// gru_wave.hip: the wave-local split-operand kernel of GRU models up to 64 units (groups = groups of 16 windows, one per wave)
int dgrp_wave_carve(int NU, gru_params &p, int mode, int64_t s, int64_t budget);
int dgrp_wave_table_bytes(int NU);
int dgrp_wave_launch(const gru_params &p, int NU, int64_t groups, int wave_bytes, bool onercp, hipStream_t stream);
int dgrp_spill_row(const dgrp_model *m);   // row length of the avg[t] spill (gru_kernel.hip)
// rnn_stream.hip: GRU with 129-256 units on waves of 64 units with resident hi fragments (NW = 32-unit slices of the model)
size_t dgrp_stream64_carve(int NW, gru_params &p, int mode, int64_t s, int64_t budget);
int dgrp_stream64_launch(const gru_params &p, int NW, int64_t groups, size_t lds, hipStream_t stream);
// batched records (mode 0): see gru_kernel.hip
int dgrp_gru_launch_batch(const dgrp_model *m, const uint8_t *d_idx, int64_t s, const void *d_recs, const int64_t *d_wg_first,
                          int64_t nrec, int64_t total_groups, int mode, float *d_out, void *d_avg, hipStream_t stream);
int dgrp_attention_launch_recs(const dgrp_model *m, int64_t s, dgrp_placement place, int64_t w0, int64_t nw,
                               int merge, int64_t n, const void *d_avg, const float *d_pl, float *d_out,
                               const void *d_recs, int64_t nrec, hipStream_t stream);
// layout of one record-table entry (64 bytes): idx_off, n, out_row, nwin, place.nfullB, place.shift, win_first, 0


/*
 * deepgrp_hip.h -- C ABI of libdeepgrp_hip.so, the MI355X (gfx950) implementation of
 * DeepGRP's prediction hot path.
 *
 * This is the drop-in boundary: plain C, raw pointers and sizes, no torch / Python
 * types.  Every entry point names the reference interface it replaces (paths are
 * relative to the upstream fhausmann/deepgrp repository).  The reference has two C
 * symbols on this path (`_get_max`, deepgrp/maxcalc.h:3-4, and `mss_find_all`,
 * deepgrp/_mss/mss.h:16-17) and otherwise crosses into compiled code through Cython
 * (deepgrp/sequence.pyx, deepgrp/_mss/pymss.pyx) and TensorFlow
 * (`model.predict_on_batch`, deepgrp/prediction.py:106).  INTEGRATION.md shows the
 * ctypes stubs a maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success and a negative DGRP_E* code on failure;
 *     dgrp_last_error() returns a thread-local message for the last failure;
 *   - pointers prefixed d_ are DEVICE pointers (HBM, allocated by the caller, e.g. by
 *     torch.empty(..., device="cuda")); h_ are host pointers;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work
 *     is enqueued asynchronously on it unless a function says it synchronises;
 *   - functions are re-entrant; a dgrp_model may be shared by threads that use
 *     different streams and workspaces.
 */
#ifndef DEEPGRP_HIP_H_
#define DEEPGRP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGRP_OK 0
#define DGRP_EINVAL (-1)   /* bad argument (shape, NULL pointer, unsupported size)   */
#define DGRP_EHIP (-2)     /* a HIP runtime call or kernel launch failed              */
#define DGRP_ENOMEM (-3)   /* caller-provided workspace / output capacity too small   */
#define DGRP_ENODEV (-4)   /* no gfx950 device visible                                */

#define DGRP_ABI_VERSION 1

/* Segment record written by dgrp_segments(): one TSV row of deepgrp/__main__.py:288-292
 * without the two name columns. */
typedef struct dgrp_segment {
    int64_t start;   /* 0-based, inclusive, in ORIGINAL (unstripped) record coordinates */
    int64_t end;     /* exclusive                                                         */
    int32_t label;   /* 1..C-1                                                            */
    int32_t contig;  /* caller-defined tag copied from the call (record index)            */
} dgrp_segment;

typedef struct dgrp_model dgrp_model;   /* opaque: packed weights resident in HBM */

int dgrp_abi_version(void);
const char *dgrp_last_error(void);
/* Name / CU count / HBM bytes of the current device (synchronous). */
int dgrp_device_info(char *name, size_t name_cap, int *cu_count, int64_t *hbm_bytes);

/* ---- A2: deepgrp.sequence.one_hot_encode_dna_sequence (deepgrp/sequence.pyx:19-36, :55-58)
 * Host helper: number of leading exact 'N' bytes and kept length after dropping leading and
 * trailing 'N' (negative for an all-N record: the reference raises ValueError there). */
int dgrp_strip_n(const uint8_t *h_seq, int64_t len, int64_t *startpos, int64_t *kept);
/* bytes -> class index 0..4 per base (A,a=0 C,c=1 G,g=2 T,t=3 else 4), the compact form every
 * other kernel consumes. */
int dgrp_encode(const uint8_t *d_seq, int64_t n, uint8_t *d_idx, void *stream);
/* bytes -> int8 [5, n] C-order, exactly the array the reference function returns. */
int dgrp_onehot(const uint8_t *d_seq, int64_t n, int8_t *d_onehot, void *stream);

/* ---- A1+A2 fused for one record body: _read_multi_fasta's per-line strip()/upper()/join
 * (deepgrp/__main__.py:31-41) followed by one_hot_encode_dna_sequence's N stripping and class lookup
 * (deepgrp/sequence.pyx:27-35), on the raw bytes between a header line and the next header.
 * d_raw [nbytes] device bytes as they are in the file; d_idx receives one class index per sequence
 * character (line ends removed), BEFORE N stripping, capacity nbytes.  h_info (host, filled after a
 * stream synchronisation): [0] 1 if the body is "plain" (ASCII, no whitespace except LF / CRLF line
 * ends, no blank line) so that deleting line ends equals stripping every line -- otherwise the caller
 * must parse this record with the reference loop; [1] sequence length; [2] startpos = number of leading
 * 'N'/'n'; [3] kept length after dropping leading and trailing N (negative for an all-N record).
 * The class indices of the kept part are d_idx[startpos .. startpos + kept). */
int64_t dgrp_fasta_workspace_bytes(int64_t nbytes);
int dgrp_fasta_encode(const uint8_t *d_raw, int64_t nbytes, uint8_t *d_idx, int64_t *h_info, void *d_work,
                      int64_t work_bytes, void *stream);

/* The same for nrec record bodies of ONE device buffer in a single call (one read-back, one synchronisation):
 * record r = bytes [h_off[r], h_off[r] + h_len[r]) of d_raw, its indices go to d_idx + h_off[r], its four info
 * values to h_info[4 r ..].  For files of many short records. */
int64_t dgrp_fasta_batch_workspace_bytes(int64_t nrec, int64_t total_bytes);
int dgrp_fasta_encode_batch(const uint8_t *d_raw, int64_t nrec, const int64_t *h_off, const int64_t *h_len,
                            uint8_t *d_idx, int64_t *h_info, void *d_work, int64_t work_bytes, void *stream);

/* ---- A1: where the records of an uploaded FASTA file start (_read_multi_fasta, deepgrp/__main__.py:31-41: a line whose first
 * character is '>' opens a record).  d_raw [nbytes] = the file as it is, 16-byte aligned.  A chunk starts at byte 0 and at every '>'
 * that directly follows a line feed.  h_start[0..n) ascending chunk starts, h_first_lf[i] = position of the first line feed in chunk i
 * (nbytes if it has none: the header line runs to the chunk's end).  *n_chunks = chunks found; if it exceeds cap nothing else is
 * valid: call again with cap >= *n_chunks.  Synchronous.  Replaces host passes over the file bytes (numpy compare + nonzero). */
int64_t dgrp_fasta_chunks_workspace_bytes(int64_t cap);
int dgrp_fasta_chunks(const uint8_t *d_raw, int64_t nbytes, int64_t cap, int64_t *h_start, int64_t *h_first_lf,
                      int64_t *n_chunks, void *d_work, int64_t work_bytes, void *stream);

/* ---- A3: deepgrp.prediction.fetch_validation_batch (deepgrp/prediction.py:14-37)
 * Number of windows len(range(0, n - T, s)). */
int64_t dgrp_window_count(int64_t n, int64_t T, int64_t s);
/* Materialise windows w0 .. w0+nw-1 as one-hot [nw, T, 5]; elem = 2 (fp16) or 4 (fp32, what the
 * reference's generator yields).  Not used by the fused path; kept for API parity and as the
 * HBM-roofline probe of the sliding encoder. */
int dgrp_windows_onehot(const uint8_t *d_idx, int64_t n, int64_t T, int64_t s, int64_t w0,
                        int64_t nw, int elem, void *d_out, void *stream);

/* ---- A13: the tensors tf.keras.models.load_model extracts from the HDF5 file
 * (deepgrp/__main__.py:264-270; layer graph deepgrp/model.py:293-336).  Host float32 arrays in
 * Keras layout, gate columns z|r|h:  kernel [5,3u], recurrent [u,3u], bias [2,3u],
 * scale [u] or NULL (no attention), ff_kernel [(attention?2u:u), C], ff_bias [C].
 * Packs them into MFMA fragment order and uploads (synchronous).
 * Sizes: 1 <= u <= 2048, 2 <= C <= 64 (the reference's classes are len(repeats_to_search) + 1 = 5; labels are int8), 1 <= T <= 65535.
 * Up to 256 units and 16 classes the fused kernels run (their logit tile is 16 wide); beyond either, the model is created on the
 * "fp32 path" (dgrp_model_flags bit 2): every forward call goes through the plain-fp32 kernels of ref_kernels.hip, tens of Mbp/s --
 * the reference takes any `units` (deepgrp/model.py:117,219-229), so a large model is slow here, not refused. */
int dgrp_model_create(dgrp_model **out, int T, int u, int C, int attention, const float *h_kernel,
                      const float *h_recurrent, const float *h_bias, const float *h_scale,
                      const float *h_ff_kernel, const float *h_ff_bias);
/* rnn = "LSTM" variant (deepgrp/model.py:219-223, no attention): kernel [5,4u], recurrent [u,4u],
 * bias [4u], gate columns i|f|c|o; ff_kernel [u, C], ff_bias [C].  Up to 128 units both kernel sets exist; 129-256 units run the
 * streamed split-operand kernel at either precision level; beyond 256 units the fp32 path, as above. */
int dgrp_model_create_lstm(dgrp_model **out, int T, int u, int C, const float *h_kernel,
                           const float *h_recurrent, const float *h_bias, const float *h_ff_kernel,
                           const float *h_ff_bias);
int dgrp_model_destroy(dgrp_model *m);
int dgrp_model_dims(const dgrp_model *m, int *T, int *u, int *C, int *attention);
/* Which kernel variant the constructor chose (diagnostic; results are the same within tolerance):
 * bit 0 = GRU blend with one reciprocal per (row, unit) -- only when the weights' absolute column sums
 * prove (1 + 2^az)(1 + 2^ag) finite in float32; otherwise (or with DGRP_GRU_SAFE=1 in the environment at
 * construction) the two-reciprocal form; bit 1 = precision level 1 (below); bit 2 = the model runs on the fp32 path (more units
 * than the fused kernels take).  Negative on a NULL model. */
int dgrp_model_flags(const dgrp_model *m);
/* (addition) Precision of the recurrent contraction for every later call on this handle:
 * 1 = split operands, the DEFAULT of every model: weights and hidden state enter the matrix cores as fp16 hi+lo pairs, three MFMA
 * passes, fp32-grade pre-activations -- class probabilities within 1e-5 of a float64 evaluation in the package's tests (GRU up to 64
 * units: gru_wave_kernel; 97-128 units: gru_split2_kernel; other GRU sizes up to 128 units: gru_split_kernel; GRU with 129-256 units
 * and the LSTM cell up to 256 units: rnn_split_stream_kernel).  With attention the recurrent pre-pass runs split and avg[t] crosses to
 * the second kernel as FLOAT32 (fp16 at level 0).
 * 0 = fp16 MFMA operands (`predict --fast`): 2-2.5x faster, class probabilities within 1e-3 of fp32 except on ill-conditioned windows
 * (measure with `python -m deepgrp_amd verify`).  The LSTM cell beyond 128 units and models on the fp32 path (dgrp_model_flags bit
 * 2) have one kernel set and accept either level.  DGRP_EINVAL for any other level.  dgrp_model_flags bit 1 reports the level.
 * The level is a property of the HANDLE, read by every call at launch time: do not change it on a handle other host threads are
 * using -- give each user its own view (dgrp_model_view) instead. */
int dgrp_model_set_precision(dgrp_model *m, int level);
/* (addition) A second handle on the same device-resident model with its own precision level: shares every buffer of `m` (which must
 * outlive it), costs nothing, is released with dgrp_model_destroy.  The package's pipelines hold one view each, so that pipelines of
 * different levels can run records on a pool of host threads without touching each other's setting. */
int dgrp_model_view(const dgrp_model *m, int level, dgrp_model **out);

/* ---- A4: model.predict_on_batch (deepgrp/prediction.py:106)
 * Bytes of scratch HBM dgrp_forward_* needs for `nw` windows in one call. */
int64_t dgrp_forward_workspace_bytes(const dgrp_model *m, int64_t nw);
/* Windows per call the library itself uses inside dgrp_predict_record, and the size callers of dgrp_forward_merge should cut a
 * record into (the reference's predict loop, deepgrp/prediction.py:104-110, goes batch by batch; here a "batch" is a launch):
 * 2^23 without attention (a 250 Mbp chromosome at stride 50 in one launch); with attention as many as keep the avg[t] spill of one launch within 1/32 of the card's memory, at most
 * 8 GiB (environment DGRP_SPILL_BYTES overrides), in multiples of 32 768 windows (whole rounds of workgroups for every recurrent
 * kernel) or, below that, of 4096.  0 for a null model. */
int64_t dgrp_forward_window_chunk(const dgrp_model *m);
/* Class probabilities [nw, T, C] float32 of windows w0 .. w0+nw-1 of the index array.  (A workgroup stages its 16
 * windows in LDS: 16 T bytes next to ~25 KiB of state at 128 units, so T up to about 8 000; larger windows are
 * refused with DGRP_EINVAL.) */
int dgrp_forward_windows(const dgrp_model *m, const uint8_t *d_idx, int64_t n, int64_t s,
                         int64_t w0, int64_t nw, float *d_probs, void *d_work, int64_t work_bytes,
                         void *stream);

/* ---- A4+A5+A6 fused: deepgrp.prediction.predict (deepgrp/prediction.py:89-111) with
 * get_max (deepgrp/sequence.pyx:65-76 -> deepgrp/maxcalc.c:10-24) folded into the classifier
 * epilogue.  d_out is float32 [n, C] and MUST be zero-filled by the caller before the first call
 * for a record (np.zeros, prediction.py:103).  Windows w0 .. w0+nw-1 are max-merged at the rows
 * the reference would use for a user batch size `batch` INCLUDING its partial-last-batch
 * offset, computed from the record's total window count. */
int dgrp_forward_merge(const dgrp_model *m, const uint8_t *d_idx, int64_t n, int64_t s,
                       int64_t batch, int64_t w0, int64_t nw, float *d_out, void *d_work,
                       int64_t work_bytes, void *stream);

/* prediction.py:89-111 for a WHOLE record: every window, chunk by chunk (dgrp_forward_window_chunk), max-merged into d_out [n, C],
 * which the caller has zeroed.  Attention models whose chunk is at least 65 536 windows (up to 64 units at the usual window sizes)
 * alternate the chunks between three internal streams ("lanes"), each with a spill of its own: the second kernel of one chunk --
 * HBM-bound -- runs beside the recurrent pre-pass of the next; the caller's stream is forked in front of the first chunk and joined
 * behind the last.  A max-merge: the result does not depend on the order, bit for bit.  dgrp_predict_record runs this. */
int64_t dgrp_forward_merge_record_workspace_bytes(const dgrp_model *m, int64_t n, int64_t s);
int dgrp_forward_merge_record(const dgrp_model *m, const uint8_t *d_idx, int64_t n, int64_t s, int64_t batch, float *d_out,
                              void *d_work, int64_t work_bytes, void *stream);

/* Accuracy yardstick (an addition; no counterpart in the reference, whose TensorFlow graph IS fp32): the same windows
 * through a plain fp32 evaluation of deepgrp/model.py:293-336 on the device -- fp32 weights and state, expf/tanhf, no
 * fp16, no MFMA -- so that the deviation of the fp16-operand fused kernel can be measured on the caller's own weights
 * and sequence (`python -m deepgrp_amd verify`).  Slow (one workgroup per window and strand); meant for hundreds of
 * windows.  Nothing on the prediction path calls it. */
int64_t dgrp_forward_reference_workspace_bytes(const dgrp_model *m, int64_t nw);
int dgrp_forward_windows_reference(const dgrp_model *m, const uint8_t *d_idx, int64_t n, int64_t s, int64_t w0,
                                   int64_t nw, float *d_probs, void *d_work, int64_t work_bytes, void *stream);

/* ---- A6 standalone: float *_get_max(float *output, float *inputs, size_t dim0, size_t dim1,
 * size_t stride, size_t batchsize)  (deepgrp/maxcalc.h:3-4).  Same argument meaning; buffers on
 * the device; out_rows bounds the writes (the reference has no bounds check). */
int dgrp_get_max(float *d_output, int64_t out_rows, const float *d_inputs, int64_t dim0,
                 int64_t dim1, int64_t stride, int64_t batchsize, void *stream);

/* ---- A7: the score transform of deepgrp.prediction.apply_mss (deepgrp/prediction.py:51-57)
 * probs float32 [n, C] -> scores float64 [n] and classes int8 [n] (argmax, first maximum). */
int dgrp_scores(const float *d_probs, int64_t n, int C, double *d_scores, int8_t *d_cls,
                void *stream);
/* ---- A8: deepgrp.prediction.softmax + argmax (deepgrp/prediction.py:62-65,
 * deepgrp/__main__.py:81-83): labels int8 [n]; d_softmax (float32 [n, C]) may be NULL. */
int dgrp_softmax_labels(const float *d_probs, int64_t n, int C, float *d_softmax, int8_t *d_labels,
                        void *d_work, int64_t work_bytes, void *stream);

/* ---- A9+A10: deepgrp.mss.find_mss_labels (deepgrp/_mss/pymss.pyx:16-80) over
 * msseg_t *mss_find_all(int n, const double *S, double min_sc, double xdrop, int *n_seg)
 * (deepgrp/_mss/mss.h:16-17).  scores float64 [n], labels int8 [n] in; labels int8 [n] out (the
 * argmax of the reference's one-hot rows).  n < 2^31 like the reference.  If d_nseg != NULL it
 * receives the number of maximal segments kept (device int64).  Synchronises the stream (the
 * fixed-point loop over independently scanned stretches reads a flag back). */
int64_t dgrp_mss_workspace_bytes(int64_t n);
int dgrp_mss_labels(const double *d_scores, const int8_t *d_cls, int64_t n, int nof_labels,
                    int min_mss_len, int xdrop_len, int8_t *d_labels_out, int64_t *d_nseg,
                    void *d_work, int64_t work_bytes, void *stream);
/* The segments themselves (st, en as int32 pairs, in order) for the last dgrp_mss_labels call on
 * this workspace: copies up to cap pairs to the host, returns the count via *n_seg (synchronous). */
int dgrp_mss_segments_host(const void *d_work, int64_t work_bytes, int32_t *h_st_en, int64_t cap,
                           int64_t *n_seg);

/* The same for MANY records side by side (files of thousands of short records): record r occupies
 * [h_start[r], h_start[r+1]) of d_scores / d_cls / d_labels_out, every start a multiple of 64, h_start[nrec] =
 * total_n < 2^31; positions between a record's last base and the next start must hold score 0.0 and class 0 (they
 * behave like the end of the sequence, deepgrp/_mss/mss.c:96).  One wave per record, one launch per kernel for all
 * of them; synchronises the stream. */
int64_t dgrp_mss_batch_workspace_bytes(int64_t total_n, int64_t nrec);
int dgrp_mss_labels_batch(const double *d_scores, const int8_t *d_cls, int64_t total_n, int64_t nrec,
                          const int64_t *h_start, int nof_labels, int min_mss_len, int xdrop_len,
                          int8_t *d_labels_out, void *d_work, int64_t work_bytes, void *stream);

/* ---- A11: deepgrp.sequence.yield_segments / get_segments (deepgrp/sequence.pyx:38-53,
 * :79-85) filtered by label > 0 (deepgrp/__main__.py:290): run-length extraction with the
 * "last element is its own segment" behaviour.  Writes up to cap records (device) and the total
 * count (device int64; may exceed cap, in which case only cap were written). */
int64_t dgrp_segments_workspace_bytes(int64_t n);
int dgrp_segments(const int8_t *d_labels, int64_t n, int64_t offset, int32_t contig,
                  dgrp_segment *d_records, int64_t cap, int64_t *d_count, void *d_work,
                  int64_t work_bytes, void *stream);

/* ---- A12: the TSV rows of deepgrp/__main__.py:291-292 as text (host code, host buffers): per row
 * "<prefix>start\tend\tlabel\n", prefix i = bytes [prefix_off[i], prefix_off[i+1]) of `prefixes` ("file\theader\t" of record i);
 * a row takes prefix rows[r].contig if by_contig != 0, else prefix 0.  out capacity >= dgrp_format_rows_bound(nrows, longest
 * prefix); *written = bytes produced. */
int64_t dgrp_format_rows_bound(int64_t nrows, int64_t longest_prefix);
int dgrp_format_rows(const char *prefixes, const int64_t *prefix_off, int64_t nprefix, int by_contig,
                     const dgrp_segment *rows, int64_t nrows, char *out, int64_t cap, int64_t *written);

/* ---- A3-A11 in one call: everything deepgrp/__main__.py:46-83 and :288-292 do for ONE record whose class indices
 * (after N stripping, startpos = offset) are in HBM: windows, forward, max-merge with the reference's placement for
 * `batch`, then scores + MSS labels (use_mss != 0; deepgrp/prediction.py:40-59) or softmax + argmax
 * (deepgrp/__main__.py:81-83), then the label > 0 segments.  Writes up to cap records to d_records, the total to
 * *h_count (host; larger than cap means: call again with more room), synchronises the stream.  All intermediates
 * live in d_work (dgrp_record_workspace_bytes); one host thread per stream may run records concurrently. */
int64_t dgrp_record_workspace_bytes(const dgrp_model *m, int64_t n, int64_t s, int use_mss);
int dgrp_predict_record(const dgrp_model *m, const uint8_t *d_idx, int64_t n, int64_t s, int64_t batch,
                        int min_mss_len, int xdrop_len, int use_mss, int64_t offset, int32_t contig,
                        dgrp_segment *d_records, int64_t cap, int64_t *h_count, void *d_work, int64_t work_bytes,
                        void *stream);

/* The same chain for a BATCH of short records in a handful of launches (a file of thousands of contigs): record r =
 * h_n[r] >= 1 class indices at d_idx + h_idx_off[r] (after N stripping, startpos h_startpos[r]); any model (attention
 * spills avg[t] of every window of the batch into the workspace, so batches are small), MSS labels (the -m path goes
 * record by record: its softmax subtracts the record's global maximum).  One GRU / LSTM launch covers the windows
 * of all records, the post-processing runs on the records laid side by side (64-aligned), one wave per record in
 * the MSS scan.  d_records receives the segments of all records in record order, then position order, `contig` =
 * h_contig[r]; *h_count as in dgrp_predict_record.  Synchronises the stream.  Rows (each record rounded up to 64)
 * must stay below 2^31. */
int64_t dgrp_batch_workspace_bytes(const dgrp_model *m, int64_t nrec, const int64_t *h_n, int64_t s);
int dgrp_predict_batch(const dgrp_model *m, const uint8_t *d_idx, int64_t nrec, const int64_t *h_idx_off,
                       const int64_t *h_n, const int64_t *h_startpos, const int32_t *h_contig, int64_t s, int64_t batch,
                       int min_mss_len, int xdrop_len, dgrp_segment *d_records, int64_t cap, int64_t *h_count,
                       void *d_work, int64_t work_bytes, void *stream);

/* ---- N2: evaluation helpers of deepgrp.prediction on label arrays that are already in HBM.
 * deepgrp.prediction.confusion_matrix (deepgrp/prediction.py:204-222): d_cnf int64 [ncls, ncls] (zeroed here),
 * cnf[true, pred] += 1 per base; labels int8 in [0, ncls), ncls <= 64, arrays 16-byte aligned.  *d_bad (device
 * int) is set to 1 if any label falls outside [0, ncls) -- the reference raises IndexError there. */
int dgrp_confusion_matrix(const int8_t *d_true, const int8_t *d_pred, int64_t n, int ncls, int64_t *d_cnf,
                          int *d_bad, void *stream);
/* deepgrp.prediction.filter_segments (deepgrp/prediction.py:244-260): runs of one positive label shorter than
 * min_len become 0.  d_out may be d_labels (the reference works in place). */
int dgrp_filter_segments(const int8_t *d_labels, int8_t *d_out, int64_t n, int64_t min_len, void *stream);

/* ---- instrumentation (bench.py's roofline figure; no counterpart in the reference, no effect on results).
 * While enabled for the CALLING HOST THREAD, every launch of a recurrent forward kernel (GRU / LSTM, fused or split) that this
 * thread makes through any entry point above is bracketed by two HIP events on the launch's stream.  dgrp_kernel_timer_read waits
 * for the recorded events, returns the summed device time in milliseconds, the number of launches and the windows they covered,
 * and forgets them; enabling again also starts from an empty list. */
int dgrp_kernel_timer_enable(int on);
int dgrp_kernel_timer_read(double *h_ms, int64_t *h_launches, int64_t *h_windows);

#ifdef __cplusplus
}
#endif
#endif /* DEEPGRP_HIP_H_ */

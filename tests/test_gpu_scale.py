"""BASELINE.json configurations as parity cases: cfg1 (1 Mbp, defaults.toml model) against the
full CPU oracle, and cfg2-size runs checked through exact post-processing parity plus
size-independent properties of the merged array and the segment list."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from deepgrp_amd import synthetic                                    # noqa: E402
from deepgrp_amd.pipeline import ContigPipeline, DeviceModel, upload_sequence   # noqa: E402


def _rows3(rows):
    return np.stack([rows["start"], rows["end"], rows["label"]], 1).reshape(-1, 3)


def test_cfg1_defaults_model_1mbp_vs_full_oracle(orc):
    """configs[0]: 1 Mbp random-ACGTN FASTA, defaults.toml model (T=342, u=60, attention) -- the whole
    path on the CPU oracle (float32 NN like TF) next to the GPU: probabilities within 1e-3, and the
    TSV rows identical once the oracle post-processes the GPU's probabilities."""
    T, u, s, B = 342, 60, 50, 256
    w = orc.Weights.random(u, 5, T, True, seed=11, gain=2.0)
    dm = DeviceModel(w.kernel, w.recurrent, w.bias, w.ff_kernel, w.ff_bias, w.scale, vecsize=T)
    seq = synthetic.synthetic_chromosome(1_000_000, contig=7).decode()
    pipe = ContigPipeline(dm, s, B, 50, 50, True)
    st, d_idx = upload_sequence(seq.encode())
    idx = d_idx.cpu().numpy()
    n = idx.size
    nwin = orc.window_count(n, T, s)
    merged = pipe.merged(d_idx)
    ref_probs = orc.nn_forward(idx, w, s, 0, nwin, np.float32)
    ref_merged = orc.merge_all(ref_probs, n, s, B)
    assert np.abs(merged.cpu().numpy() - ref_merged).max() < 1e-3
    rows = pipe.segments(pipe.labels(merged), st)
    probs = dm.forward_windows(d_idx, s, 0, nwin).cpu().numpy()
    want = orc.predict_contig(seq, lambda _i: (lambda a, b: probs[a:a + b]), T, 5, s, B, 50, 50, True)
    np.testing.assert_array_equal(_rows3(rows), want)
    dm.close()


@pytest.fixture(scope="module")
def trained():
    w = synthetic.trained_weights()
    dm = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, 200)
    yield dm
    dm.close()


def _far_windows_vs_oracle(orc, owts, model, d_idx, s, w0, nw=32, tol=1e-5):
    """Windows [w0, w0 + nw) of a device-resident record against the float64 CPU statement (oracle/dgrp_oracle.c): the oracle is
    handed only the bases those windows cover, so it costs milliseconds wherever in the record they lie."""
    T = model.vecsize
    w0 = max(0, min(w0, len(range(0, d_idx.numel() - T, s)) - nw))        # (a short last batch may hold fewer than nw / 2 windows)
    a = w0 * s
    tail = d_idx[a:a + (nw - 1) * s + T].cpu().numpy()
    want = orc.nn_forward(tail, owts, s, 0, nw, np.float64)
    got = model.forward_windows(d_idx, s, w0, nw).cpu().numpy()
    err = float(np.abs(got - want).max())
    assert err < tol, f"windows {w0}..{w0 + nw}: |dp| = {err:.3e} against the float64 oracle"
    return err


@pytest.fixture(scope="module")
def trained_oracle_weights(orc):
    w = synthetic.trained_weights()
    return orc.Weights(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, 200)


def test_20mbp_postprocessing_exact(orc, trained):
    """Scores, classes, labels and rows of a 20 Mbp record: the GPU path against the sequential
    oracle on the same merged probabilities (thousands of independently scanned stretches)."""
    raw = synthetic.synthetic_chromosome(20_000_000, contig=3)
    st, d_idx = upload_sequence(raw)
    pipe = ContigPipeline(trained)
    merged = pipe.merged(d_idx)
    labels = pipe.labels(merged)
    rows = pipe.segments(labels, st)
    m = merged.cpu().numpy()
    sc, cl = orc.scores(m)
    lab = orc.find_mss_labels(sc, cl, 5, 50, 50)
    np.testing.assert_array_equal(labels.cpu().numpy(), lab)
    np.testing.assert_array_equal(_rows3(rows), orc.segments(lab, st))


def _size_independent_properties(trained, n_bases, contig, other_chunk):
    """Properties that hold for any input, at any size; returns what the size-specific checks go on with."""
    T, s, C = 200, 50, 5
    raw = synthetic.synthetic_chromosome(n_bases, contig=contig)
    st, d_idx = upload_sequence(raw)
    del raw
    n = d_idx.numel()
    assert st == 10_000 and n == n_bases - 20_000
    pipe = ContigPipeline(trained)
    merged = pipe.merged(d_idx)
    # chunking the windows differently must not change a bit (max is order independent)
    merged2 = ContigPipeline(trained, chunk_windows=other_chunk).merged(d_idx)
    assert torch.equal(merged, merged2)
    del merged2
    nwin = len(range(0, n - T, s))
    covered = (nwin // 256 * 256 - 1) * s + T                  # full batches sit where they belong
    assert float(merged[:covered].max(dim=1).values.min()) >= 1.0 / C - 1e-6       # a softmax row's max
    assert float(merged[:covered].sum(dim=1).min()) >= 1.0 - 1e-5                  # sum_c max_w >= max_w sum_c
    assert float(merged.max()) <= 1.0 + 1e-6 and float(merged.min()) >= 0.0
    last_placed = max((nwin // 256 * (nwin % 256) + (nwin % 256) - 1) * s + T, covered)
    assert float(merged[last_placed:].abs().max()) == 0.0                           # never-written tail stays zero
    labels = pipe.labels(merged)
    assert int(labels.min()) >= 0 and int(labels.max()) < C
    rows = pipe.segments(labels, st)
    assert (rows["start"] < rows["end"]).all() and (rows["start"][1:] >= rows["end"][:-1]).all()
    assert rows["start"].min() >= st and rows["end"].max() <= st + n and (rows["label"] > 0).all()
    # independent run-length encoding on the host (numpy) of the same labels
    lab = labels.cpu().numpy()
    body = lab[:-1]
    change = np.flatnonzero(np.diff(body) != 0) + 1
    starts = np.concatenate([[0], change])
    ends = np.concatenate([change, [body.size]])
    keep = body[starts] > 0
    want = np.stack([starts[keep] + st, ends[keep] + st, body[starts[keep]]], 1)
    if lab[-1] > 0:
        want = np.concatenate([want, [[n - 1 + st, n + st, lab[-1]]]])
    np.testing.assert_array_equal(_rows3(rows), want)
    # MSS only ever relabels zeros inside kept segments
    sc = torch.empty(n, dtype=torch.float64, device=merged.device)
    cls = merged.argmax(dim=1)
    changed = labels.long() != cls
    assert bool((cls[changed] == 0).all())
    return pipe, st, d_idx, merged, rows


def test_50mbp_properties(trained):
    """configs[1] size."""
    _size_independent_properties(trained, 50_000_000, 0, 77_776)


def test_250mbp_properties_and_far_offsets(orc, trained, trained_oracle_weights):
    """configs[2]: the 250 Mbp chromosome the north-star target is quoted on (a 1.25 G-element merged array: the place a
    32-bit index would slip).  The property set of the 50 Mbp case, the one-call path (dgrp_predict_record, what the
    command line and bench.py run) against the staged one, and -- since no CPU statement covers this size -- windows
    and merged rows at the FAR end of the record against the plain-fp32 kernels: window probabilities of the last 4 096
    windows within 1e-5 (fp32-grade) of the yardstick, and the merged rows they cover (incl. the reference's
    partial-last-batch placement) bit-identical to a max-merge of those same probabilities done here."""
    T, s, C, B = 200, 50, 5, 256
    pipe, st, d_idx, merged, rows = _size_independent_properties(trained, 250_000_000, 2, 300_016)
    n = d_idx.numel()
    one_call = pipe.run_idx(d_idx, st)
    np.testing.assert_array_equal(_rows3(one_call), _rows3(rows))
    assert len(rows) > 50_000
    nwin = len(range(0, n - T, s))
    nw = 4096
    w0 = nwin - nw
    got = trained.forward_windows(d_idx, s, w0, nw)
    ref = trained.forward_windows_reference(d_idx, s, w0, nw)
    assert float((got - ref).abs().max()) < 1e-5
    # ... and against the float64 oracle itself: the last 32 windows, 32 of the short last batch's, 32 from the middle
    for start in (nwin - 32, nwin // B * B - 16, nwin // 2):
        _far_windows_vs_oracle(orc, trained_oracle_weights, trained, d_idx, s, start)
    # placement at far offsets: full-batch windows sit at w * s, the short last batch (r windows) at
    # (nfull * r + (w - nfull * B)) * s -- for this record in the middle of the array
    nfull, r = nwin // B, nwin % B
    assert r > 0 and nfull * B > w0
    # rows reached by NO window outside [w0, nfull * B): a max-merge of `got` done here must equal the array bit for bit
    row_a, row_b = w0 * s + T, nfull * B * s
    want = torch.zeros((row_b - w0 * s + T, C), dtype=torch.float32, device=merged.device)
    for w in range(w0, nfull * B):
        a = (w - w0) * s
        want[a:a + T] = torch.maximum(want[a:a + T], got[w - w0])
    assert torch.equal(merged[row_a:row_b], want[T:T + row_b - row_a])
    # the short batch: every one of its windows is dominated by the array at its (shifted) rows
    for w in range(nfull * B, nwin):
        a = (nfull * r + (w - nfull * B)) * s
        assert bool((merged[a:a + T] >= got[w - w0]).all())


def test_900mbp_beyond_32bit_elements(orc, trained, trained_oracle_weights):
    """A record larger than any of BASELINE's configurations: 900 Mbp (the largest chromosomes sequenced so far are of this order),
    i.e. a merged array of 4.5 G elements -- past 2^32, where an index held in 32 bits anywhere on the path would wrap -- 18 M
    windows in 18 launches, and (SURVEY 8b iii) still inside the reference's own limit of n < 2^31 for `mss_find_all`.  Checked
    like the 250 Mbp record: the size-independent properties (incl. the host's own run-length encoding of the labels), the
    one-call path against the staged one, the LAST windows against the plain-fp32 kernels and the rows they cover bit for bit."""
    T, s, C, B = 200, 50, 5, 256
    pipe, st, d_idx, merged, rows = _size_independent_properties(trained, 900_000_000, 5, 1_000_016)
    n = d_idx.numel()
    assert n * C > 2 ** 32
    one_call = pipe.run_idx(d_idx, st)
    np.testing.assert_array_equal(_rows3(one_call), _rows3(rows))
    nwin = len(range(0, n - T, s))
    nw = 2048
    nfull, r = nwin // B, nwin % B
    w0 = nfull * B - nw
    got = trained.forward_windows(d_idx, s, w0, nw)
    ref = trained.forward_windows_reference(d_idx, s, w0, nw)
    assert float((got - ref).abs().max()) < 1e-5
    for start in (nwin - 32, w0 + nw - 32, (2 ** 32 // C) // s + 7):       # the last windows; the last full batch; rows around element 2^32
        _far_windows_vs_oracle(orc, trained_oracle_weights, trained, d_idx, s, start)
    row_a, row_b = w0 * s + T, nfull * B * s
    want = torch.zeros((row_b - w0 * s + T, C), dtype=torch.float32, device=merged.device)
    for w in range(w0, nfull * B):
        a = (w - w0) * s
        want[a:a + T] = torch.maximum(want[a:a + T], got[w - w0])
    assert torch.equal(merged[row_a:row_b], want[T:T + row_b - row_a])


def test_cfg5_contig_125mbp_u256_attention(orc):
    """BASELINE configs[4]'s per-GPU workload in its real form: ONE 125 Mbp contig (the 3 Gbp genome is 24 of them, three per GPU)
    through the 256-unit attention model at window 500, stride 25 -- 5 M windows, a 20 KiB/bp avg[t] spill chunked over ~150
    launches, ~7 s of GPU time per pass.  Size-independent properties of the merged array and the rows, chunk invariance (another
    launch size: bit-identical), the one-call path against the staged one, far-end / short-batch / middle windows against the
    float64 oracle within 1e-5, and the merged rows under the last full-batch windows bit-identical to a max-merge done here."""
    T, s, C, B, u = 500, 25, 5, 256, 256
    w = orc.Weights.random(u, C, T, True, seed=21, gain=1.5)
    w.ff_bias[0] += 1.5                          # mostly background with scattered calls (pure noise would be ~10^8 one-base rows)
    dm = DeviceModel(w.kernel, w.recurrent, w.bias, w.ff_kernel, w.ff_bias, w.scale, vecsize=T)
    raw = synthetic.synthetic_chromosome(125_000_000, contig=9)
    st, d_idx = upload_sequence(raw)
    del raw
    n = d_idx.numel()
    nwin = len(range(0, n - T, s))
    pipe = ContigPipeline(dm, s, B, 50, 50, True)
    merged = pipe.merged(d_idx)
    # chunk invariance on a 5 Mbp slice of the same record (a second full pass would double the test's 7 s for nothing new):
    # the slice's own merged array from two launch sizes, bit for bit
    part = d_idx[60_000_000:65_000_000]
    m_a = ContigPipeline(dm, s, B, 50, 50, True).merged(part)
    m_b = ContigPipeline(dm, s, B, 50, 50, True, chunk_windows=10_000).merged(part)
    assert torch.equal(m_a, m_b)
    del m_a, m_b
    nfull, r = nwin // B, nwin % B
    covered = (nfull * B - 1) * s + T
    assert float(merged[:covered].max(dim=1).values.min()) >= 1.0 / C - 1e-6
    assert float(merged[:covered].sum(dim=1).min()) >= 1.0 - 1e-5
    assert float(merged.max()) <= 1.0 + 1e-6 and float(merged.min()) >= 0.0
    last_placed = max((nfull * r + r - 1) * s + T if r else 0, covered)
    assert float(merged[last_placed:].abs().max()) == 0.0
    labels = pipe.labels(merged)
    rows = pipe.segments(labels, st)
    assert (rows["start"] < rows["end"]).all() and (rows["start"][1:] >= rows["end"][:-1]).all()
    assert rows["start"].min() >= st and rows["end"].max() <= st + n and (rows["label"] > 0).all()
    lab = labels.cpu().numpy()
    sc, cl = orc.scores(merged[-2_000_000:].cpu().numpy())          # the score transform at the far end, bit for bit
    sc_d = torch.empty(2_000_000, dtype=torch.float64, device=merged.device)
    cl_d = torch.empty(2_000_000, dtype=torch.int8, device=merged.device)
    from deepgrp_amd._lib import check, lib
    from deepgrp_amd.pipeline import stream_ptr
    tail = merged[-2_000_000:].contiguous()
    check(lib().dgrp_scores(tail.data_ptr(), 2_000_000, C, sc_d.data_ptr(), cl_d.data_ptr(), stream_ptr()), "dgrp_scores")
    assert np.array_equal(sc_d.cpu().numpy().view(np.int64), sc.view(np.int64)) and np.array_equal(cl_d.cpu().numpy(), cl)
    np.testing.assert_array_equal(_rows3(rows), orc.segments(lab, st))
    one_call = pipe.run_idx(d_idx, st)
    np.testing.assert_array_equal(_rows3(one_call), _rows3(rows))
    for start in (nwin - 32, nfull * B - 16, nwin // 2, 0):
        _far_windows_vs_oracle(orc, w, dm, d_idx, s, start)
    # merged rows under the last full-batch windows that no other window reaches
    nw = 512
    w0 = nfull * B - nw
    got = dm.forward_windows(d_idx, s, w0, nw)
    row_a, row_b = w0 * s + T, nfull * B * s
    want = torch.zeros((row_b - w0 * s + T, C), dtype=torch.float32, device=merged.device)
    for k in range(nw):
        want[k * s:k * s + T] = torch.maximum(want[k * s:k * s + T], got[k])
    assert torch.equal(merged[row_a:row_b], want[T:T + row_b - row_a])
    dm.close()

"""The NN part of the oracle.  The reference's TensorFlow numerics cannot be run
offline ("parity unpinned"), so three independent statements of the same
equations are held against each other: the C code (float64 and float32), a
numpy statement written from model.py:293-336, and torch.nn.GRU on CPU with the
gate columns permuted (Keras z|r|h -> torch r|z|n)."""
import numpy as np
import pytest
import torch


def _idx(rng, n):
    return rng.choice(5, size=n, p=[0.24, 0.24, 0.24, 0.24, 0.04]).astype(np.uint8)


@pytest.mark.parametrize("u,T,attention", [(8, 20, False), (8, 20, True), (32, 50, True), (60, 34, True),
                                           (128, 40, False)])
def test_c_vs_numpy(orc, u, T, attention):
    rng = np.random.default_rng(u + T)
    w = orc.Weights.random(u, 5, T, attention, seed=u, gain=1.5)
    s = 7
    idx = _idx(rng, T + s * 9 + 3)
    a = orc.nn_forward(idx, w, s, 0, 10, np.float64)
    b = orc.nn_forward_numpy(idx, w, s, 0, 10)
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)
    c = orc.nn_forward(idx, w, s, 0, 10, np.float32)
    np.testing.assert_allclose(c, b, rtol=0, atol=2e-5)
    np.testing.assert_allclose(a.sum(axis=2), 1.0, atol=1e-12)
    # window offset argument
    d = orc.nn_forward(idx, w, s, 3, 4, np.float64)
    np.testing.assert_array_equal(d, a[3:7])


def test_reverse_complement_kat(orc):
    """tests/test_model.py:207-229 of the reference: the 6x5 ReverseComplement
    known answer, through the index form used by the oracle."""
    from oracle.oracle import _COMP
    inp = np.array([0, 1, 2, 3, 4, 0])
    expected = np.array([3, 4, 0, 1, 2, 3])
    np.testing.assert_array_equal(_COMP[inp[::-1]], expected)
    assert list(_COMP) == [3, 2, 1, 0, 4]          # tests/test_model.py:182-184


@pytest.mark.parametrize("u,T", [(16, 30), (64, 25)])
def test_gru_vs_torch(orc, u, T):
    """No-attention model == Dense(softmax) over the mean of torch.nn.GRU outputs
    on the window and on its reverse complement (not re-reversed)."""
    rng = np.random.default_rng(11)
    w = orc.Weights.random(u, 5, T, False, seed=3, gain=1.2)
    idx = _idx(rng, T + 40)
    nw, s = 6, 8
    ours = orc.nn_forward(idx, w, s, 0, nw, np.float64)

    gru = torch.nn.GRU(5, u, batch_first=True).double()
    perm = np.concatenate([np.arange(u, 2 * u), np.arange(0, u), np.arange(2 * u, 3 * u)])   # z|r|h -> r|z|n
    with torch.no_grad():
        gru.weight_ih_l0.copy_(torch.from_numpy(w.kernel.astype(np.float64)[:, perm].T.copy()))
        gru.weight_hh_l0.copy_(torch.from_numpy(w.recurrent.astype(np.float64)[:, perm].T.copy()))
        gru.bias_ih_l0.copy_(torch.from_numpy(w.bias.astype(np.float64)[0, perm].copy()))
        gru.bias_hh_l0.copy_(torch.from_numpy(w.bias.astype(np.float64)[1, perm].copy()))
    win = np.stack([idx[i * s:i * s + T] for i in range(nw)]).astype(np.int64)
    x = torch.from_numpy(np.eye(5)[win])
    comp = np.array([3, 2, 1, 0, 4])
    xrc = torch.from_numpy(np.eye(5)[comp[win[:, ::-1]]])
    with torch.no_grad():
        f, _ = gru(x)
        r, _ = gru(xrc)
    avg = ((f + r) / 2).numpy()
    logits = avg @ w.ff_kernel.astype(np.float64) + w.ff_bias.astype(np.float64)
    p = np.exp(logits - logits.max(axis=2, keepdims=True))
    p /= p.sum(axis=2, keepdims=True)
    np.testing.assert_allclose(ours, p, rtol=0, atol=1e-12)


@pytest.mark.parametrize("u,T", [(16, 30), (48, 25)])
def test_lstm_vs_torch(orc, u, T):
    """rnn="LSTM" (deepgrp/model.py:219-223): the C statement against torch.nn.LSTM (same gate order
    i|f|g|o as Keras' i|f|c|o; torch carries two bias vectors, the second is zero here)."""
    rng = np.random.default_rng(5)
    w = orc.LSTMWeights.random(u, 5, T, seed=4, gain=1.3)
    idx = _idx(rng, T + 40)
    nw, s = 6, 8
    ours = orc.lstm_forward(idx, w, s, 0, nw, np.float64)
    ours32 = orc.lstm_forward(idx, w, s, 0, nw, np.float32)
    np.testing.assert_allclose(ours32, ours, atol=2e-5)
    lstm = torch.nn.LSTM(5, u, batch_first=True).double()
    with torch.no_grad():
        lstm.weight_ih_l0.copy_(torch.from_numpy(w.kernel.astype(np.float64).T.copy()))
        lstm.weight_hh_l0.copy_(torch.from_numpy(w.recurrent.astype(np.float64).T.copy()))
        lstm.bias_ih_l0.copy_(torch.from_numpy(w.bias.astype(np.float64)))
        lstm.bias_hh_l0.zero_()
    win = np.stack([idx[i * s:i * s + T] for i in range(nw)]).astype(np.int64)
    comp = np.array([3, 2, 1, 0, 4])
    with torch.no_grad():
        f, _ = lstm(torch.from_numpy(np.eye(5)[win]))
        r, _ = lstm(torch.from_numpy(np.eye(5)[comp[win[:, ::-1]]]))
    avg = ((f + r) / 2).numpy()
    logits = avg @ w.ff_kernel.astype(np.float64) + w.ff_bias.astype(np.float64)
    p = np.exp(logits - logits.max(axis=2, keepdims=True))
    p /= p.sum(axis=2, keepdims=True)
    np.testing.assert_allclose(ours, p, rtol=0, atol=1e-12)

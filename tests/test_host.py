"""Host-side logic that needs no GPU: Options, the HDF5 reader/writer, the CLI surface, the
FASTA reader, contig sharding and the record gather (gloo, world_size 2), and the C-ABI
library's symbol table.  Reference tests mirrored where they exist (cited per test)."""
import io
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

from deepgrp_amd import hdf5, model as dgmodel, synthetic
from deepgrp_amd.__main__ import CommandLineParser, _read_multi_fasta

_DEFAULTS = {'project_root_dir': '.', 'repeats_to_search': [1, 2, 3, 4], 'vecsize': 150, 'n_epochs': 200,
             'n_batches': 250, 'early_stopping_th': 10, 'batch_size': 256, 'repeat_probability': 0.3,
             'optimizer': 'RMSprop', 'learning_rate': 0.001, 'momentum': 0.9, 'rho': 0.9, 'epsilon': 1e-10, 'rnn': 'GRU',
             'units': 32, 'dropout': 0.25, 'attention': False, 'min_mss_len': 50, 'xdrop_len': 50}
_CUSTOM = {'project_root_dir': 'test', 'repeats_to_search': [1, 2, 3], 'vecsize': 157, 'n_epochs': 206, 'n_batches': 256,
           'early_stopping_th': 11, 'batch_size': 259, 'repeat_probability': 0.2, 'optimizer': 'Adam', 'learning_rate': 0.002,
           'momentum': 0.8, 'rho': 0.8, 'epsilon': 1e-08, 'rnn': 'LSTM', 'units': 326, 'dropout': 0.256, 'attention': True,
           'min_mss_len': 507, 'xdrop_len': 507, 'some_additional': 'test'}
_CASES = [({}, _DEFAULTS), (_CUSTOM, _CUSTOM)]


class TestOptions:
    """tests/test_model.py:27-125 of the reference."""

    @pytest.mark.parametrize('init_args, expected', _CASES)
    def test_init(self, init_args, expected):
        got = dgmodel.Options(**init_args)
        for attribute, value in expected.items():
            assert getattr(got, attribute) == value

    @pytest.mark.parametrize('init_args, expected', _CASES)
    def test_fromdict_todict(self, init_args, expected):
        got = dgmodel.Options()
        got.fromdict(init_args)
        for attribute, value in expected.items():
            assert getattr(got, attribute) == value
        assert dgmodel.Options(**init_args).todict() == expected

    @pytest.mark.parametrize('init_args, expected', _CASES)
    def test_toml_round_trip(self, init_args, expected, tmp_path):
        import tomli
        path = tmp_path.joinpath('testfile.toml')
        with path.open('w') as file:
            dgmodel.Options(**init_args).to_toml(file)
        assert tomli.loads(path.read_text()) == expected
        got = dgmodel.Options.from_toml(str(path))
        with path.open() as fh:
            got2 = dgmodel.Options.from_toml(fh)
        for attribute, value in expected.items():
            assert getattr(got, attribute) == value and getattr(got2, attribute) == value
        with pytest.raises(TypeError):
            dgmodel.Options.from_toml(3)

    def test_aliases_and_items(self):
        opt = dgmodel.Options(gru_units=77, gru_dropout=0.5)
        assert opt.units == 77 and opt.dropout == 0.5 and "gru_units" not in opt.todict()
        opt["gru_units"] = 5
        assert opt["units"] == 5 and opt["gru_units"] == 5
        assert "units" in str(opt)

    def test_defaults_toml_of_the_reference_shape(self):
        txt = 'vecsize = 342\nunits = 60\nattention = true\nrepeats_to_search = [ 1, 2, 3, 4,]\nrnn = "GRU"\n'
        opt = dgmodel.Options.from_toml(io.StringIO(txt))
        assert (opt.vecsize, opt.units, opt.attention, opt.repeats_to_search) == (342, 60, True, [1, 2, 3, 4])


def test_get_dna_encoding():
    """tests/test_model.py:182-184 of the reference."""
    assert dgmodel._get_dna_encoding() == [3, 2, 1, 0, 4]


# ------------------------------------------------------------------------------------------ HDF5
@pytest.mark.parametrize("name", ["model_u8_T20", "model_u60_T342_att", "model_u16_T30_att_vlen"])
def test_read_keras_hdf5_written_by_libhdf5(name):
    """Files produced with h5py/libhdf5 (oracle/make_h5_fixtures.py) through the package's own reader."""
    w = dgmodel.read_keras_hdf5(os.path.join(GOLDEN, name + ".h5"))
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    for k in ("kernel", "recurrent_kernel", "bias", "ff_kernel", "ff_bias"):
        np.testing.assert_array_equal(w[k], z[k])
    assert (w["vecsize"], w["units"], w["classes"], w["attention"]) == (int(z["T"]), int(z["u"]), int(z["C"]), bool(z["attention"]))
    if w["attention"]:
        np.testing.assert_array_equal(w["scale"], z["scale"])
    names = [l["name"] for l in w["config"]["config"]["layers"]]
    assert names[:3] == ["input_1", "reverse_complement", "BGRU"] and names[-2:] == ["FF", "softmax"]


def test_hdf5_generic_access():
    with hdf5.File(os.path.join(GOLDEN, "model_u8_T20.h5")) as f:
        assert f.attrs["keras_version"] == b"2.5.0" and f.attrs["backend"] == b"tensorflow"
        assert set(f.keys()) == {"model_weights", "optimizer_weights"}
        assert f["model_weights"].attrs["layer_names"][2] == b"BGRU"
        ds = f["model_weights/BGRU/BGRU/gru_cell/bias:0"]
        assert ds.is_dataset and ds.read().shape == (2, 24)
        assert int(f["optimizer_weights/training/RMSprop/iter:0"].read()) == 1234
        with pytest.raises(KeyError):
            f["model_weights/nope"]
    with pytest.raises(hdf5.HDF5Error):
        hdf5.File(__file__)


@pytest.mark.parametrize("attention", [False, True])
def test_hdf5_writer_round_trip(tmp_path, attention):
    w = synthetic.synthetic_weights(60, 5, attention, seed=3)
    path = str(tmp_path / "m.hdf5")
    dgmodel.save_keras_hdf5(path, w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], w["scale"], vecsize=342)
    r = dgmodel.read_keras_hdf5(path)
    for k in ("kernel", "recurrent_kernel", "bias", "ff_kernel", "ff_bias"):
        np.testing.assert_array_equal(r[k], w[k])
    assert (r["vecsize"], r["units"], r["classes"], r["attention"]) == (342, 60, 5, attention)
    if attention:
        np.testing.assert_array_equal(r["scale"], w["scale"])


def test_hdf5_lstm_round_trip(tmp_path, orc):
    """rnn="LSTM" model files (BLSTM/lstm_cell/... tensors, class_name LSTM in model_config)."""
    w = orc.LSTMWeights.random(24, 5, 30)
    path = str(tmp_path / "lstm.hdf5")
    dgmodel.save_keras_hdf5(path, w.kernel, w.recurrent, w.bias, w.ff_kernel, w.ff_bias, None, vecsize=30, rnn="LSTM")
    r = dgmodel.read_keras_hdf5(path)
    assert (r["rnn"], r["units"], r["vecsize"], r["classes"], r["attention"]) == ("LSTM", 24, 30, 5, False)
    for k, v in (("kernel", w.kernel), ("recurrent_kernel", w.recurrent), ("ff_kernel", w.ff_kernel), ("ff_bias", w.ff_bias)):
        np.testing.assert_array_equal(r[k], v)
    np.testing.assert_array_equal(r["bias"].reshape(-1), w.bias)


def test_model_format_errors(tmp_path):
    w = hdf5.Writer()
    w.set_attr("/", "model_config", b'{"class_name": "Functional", "config": {"layers": [{"class_name": "LSTM", "name": "BLSTM", "config": {}}]}}')
    w.create_group("model_weights")
    w.save(str(tmp_path / "lstm.h5"))
    with pytest.raises(dgmodel.ModelFormatError, match="layer graph"):
        dgmodel.read_keras_hdf5(str(tmp_path / "lstm.h5"))
    w2 = hdf5.Writer()
    w2.create_dataset("x", np.arange(3, dtype=np.float32))
    w2.save(str(tmp_path / "nocfg.h5"))
    with pytest.raises(dgmodel.ModelFormatError, match="model_config"):
        dgmodel.read_keras_hdf5(str(tmp_path / "nocfg.h5"))


def test_reverse_complement_layer():
    """tests/test_model.py:182-251 of the reference: table, no weights, the 6x5 known answer, masking, serialisation."""
    assert dgmodel._get_dna_encoding() == [3, 2, 1, 0, 4]
    layer = dgmodel.ReverseComplement(complements=[3, 2, 1, 0, 4])
    layer.build((1, 10, 5))
    assert len(layer.trainable_weights) == 0 and len(layer.weights) == 0
    input_data = np.array([[1, 0, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 0, 1, 0], [0, 0, 0, 0, 1],
                           [1, 0, 0, 0, 0]]).reshape((1, -1, 5))
    expected = np.array([[0, 0, 0, 1, 0], [0, 0, 0, 0, 1], [1, 0, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 1, 0, 0],
                         [0, 0, 0, 1, 0]]).reshape((1, -1, 5))
    out = layer(input_data)
    assert out.shape == (1, 6, 5)
    np.testing.assert_equal(out, expected)
    import torch
    np.testing.assert_equal(layer(torch.from_numpy(input_data)).numpy(), expected)
    assert layer.compute_mask(input_data, [None, None]) is None
    with pytest.raises(TypeError, match="does not support masking, but was passed an input_mask"):
        layer.compute_mask(input_data, input_data)
    again = dgmodel.ReverseComplement.from_config(layer.get_config())
    np.testing.assert_equal(again(input_data), expected)
    assert layer.get_config() == {"name": "reverse_complement", "trainable": True, "dtype": "float32", "complements": [3, 2, 1, 0, 4]}
    # applying it twice is the identity for a self-inverse table
    rng = np.random.default_rng(0)
    x = np.eye(5, dtype=np.float32)[rng.integers(0, 5, size=(3, 17))]
    np.testing.assert_equal(layer(layer(x)), x)


def test_pipeline_kernel_selection_table():
    """Which forward kernels a pipeline uses: split operands by default for EVERY model (also with precise=True, which
    used to mean the plain-fp32 kernels for attention models), fp16 operands with fast=True, the plain-fp32 yardstick only when asked
    for by name; a model on the fp32 path (more units than the fused kernels take) does not batch."""
    from types import SimpleNamespace
    from deepgrp_amd.pipeline import ContigPipeline
    gru = SimpleNamespace(supports_split=True, attention=False)
    att = SimpleNamespace(supports_split=True, attention=True)
    big = SimpleNamespace(supports_split=True, attention=False, fp32_only=True)          # more than 256 units
    for model in (gru, att, big):
        for kw, want in (({}, (True, False)), ({"precise": True}, (True, False)), ({"fast": True}, (False, False)),
                         ({"fp32": True}, (False, True))):
            pipe = ContigPipeline(model, 50, 256, 50, 50, True, **kw)
            assert (pipe.split, pipe.fp32) == want, (model, kw)
            assert pipe.batchable() == (not want[1] and model is not big)
    with pytest.raises(ValueError, match="exclude"):
        ContigPipeline(gru, precise=True, fast=True)
    with pytest.raises(ValueError):
        ContigPipeline(gru, step_size=0)


def test_create_model_config_matches_the_reference_fixture():
    """tests/test_model.py:254-262 of the reference: create_model(Options(attention=True, rnn=rnn)).get_config() equals
    the stored config of its TensorFlow minor version -- 2.5 here, the version its poetry.lock pins (fixture copied
    as data from tests/test_model.json).  Same order as the reference's parametrisation: the layer-name counters of
    the second model continue those of the first."""
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "keras_model_config_tf25.json")) as fh:
        expected = json.load(fh)["2.5"]
    dgmodel.reset_layer_names()
    for rnn in ("GRU", "LSTM"):
        got = json.loads(json.dumps(dgmodel.model_config(dgmodel.Options(attention=True, rnn=rnn))))
        assert got == expected[rnn], rnn
    # a model file always carries first-of-a-session names, whatever was built before
    cfg = dgmodel.keras_config(150, 32, 5, True)
    assert cfg["class_name"] == "Functional" and cfg["config"] == expected["GRU"]
    dgmodel.reset_layer_names()


def test_initial_weights_follow_the_keras_initialisers():
    for rnn, attention in (("GRU", True), ("GRU", False), ("LSTM", True)):
        o = dgmodel.Options(units=24, vecsize=40, attention=attention, rnn=rnn)
        w = dgmodel.initial_weights(o, seed=5)
        g = 4 if rnn == "LSTM" else 3
        att = attention and rnn == "GRU"
        assert w["kernel"].shape == (5, g * 24) and w["recurrent_kernel"].shape == (24, g * 24)
        assert np.abs(w["kernel"]).max() <= np.sqrt(6 / (5 + g * 24)) and np.abs(w["kernel"]).max() > 0.1
        np.testing.assert_allclose(w["recurrent_kernel"] @ w["recurrent_kernel"].T, np.eye(24), atol=1e-5)   # orthogonal rows
        assert w["ff_kernel"].shape == ((2 if att else 1) * 24, 5) and not w["ff_bias"].any()
        if rnn == "LSTM":
            assert w["scale"] is None and w["bias"].shape == (96,)
            np.testing.assert_array_equal(w["bias"], np.r_[np.zeros(24), np.ones(24), np.zeros(48)])            # unit_forget_bias
        else:
            assert w["bias"].shape == (2, 72) and not w["bias"].any()
            assert (w["scale"] is not None) == att
        w2 = dgmodel.initial_weights(o, seed=5)
        np.testing.assert_array_equal(w["kernel"], w2["kernel"])


# ------------------------------------------------------------------------------------------ CLI
class TestCommandLineParser:
    def test_init(self):
        """tests/test_main.py:38-45 of the reference."""
        import argparse
        parser = CommandLineParser()
        assert isinstance(parser.parser, argparse.ArgumentParser)
        assert parser.threads == 1 and not parser.xla and parser.verbose == 0 and parser.args is None

    def test_reference_grammar_and_defaults(self):
        a = CommandLineParser().parse_args(["predict", "model.hdf5", "a.fa"]).args
        assert (a.batch_size, a.step_size, a.xdrop_length, a.min_mss_length, a.threads, a.xla, a.verbose) == (256, 50, 50, 50, 1, False, 0)
        assert (a.model, a.FASTA, a.output, a.no_use_mss) == ("model.hdf5", ["a.fa"], "-", False)
        a = CommandLineParser().parse_args(["-b", "4", "-s", "10", "-x", "-1", "-l", "3", "-t", "0", "--xla", "-vv", "predict", "m", "a", "b",
                                            "--output", "o.tsv", "-m"]).args
        assert (a.batch_size, a.step_size, a.xdrop_length, a.min_mss_length, a.threads, a.xla, a.verbose) == (4, 10, -1, 3, 0, True, 2)
        assert (a.FASTA, a.output, a.no_use_mss) == (["a", "b"], "o.tsv", True)

    def test_readme_grammar(self):
        """README.rst:94 `deepgrp <modelfile> <fastafile>`."""
        a = CommandLineParser().parse_args(["-b", "8", "model.hdf5", "x.fa", "-"]).args
        assert (a.command, a.model, a.FASTA, a.batch_size) == ("predict", "model.hdf5", ["x.fa", "-"], 8)

    def test_train_is_refused(self):
        p = CommandLineParser().parse_args(["train", "p.toml", "a.npz", "b.npz", "c.bed"])
        with pytest.raises(SystemExit):
            p.run()


@pytest.mark.parametrize("n", [1, 2, 3])
def test_read_multi_fasta(tmp_path, n):
    """tests/test_main.py:241-250 of the reference."""
    rng = np.random.default_rng(n)
    sequences = {f"chr{i+1}": "".join(rng.choice(["N", "A", "C", "G", "T"], size=100)) for i in range(n)}
    path = tmp_path / "chr_dummy.fa"
    path.write_text("\n".join(f">{h}\n{s}" for h, s in sequences.items()))
    with path.open() as fh:
        assert dict(_read_multi_fasta(fh)) == sequences


def test_read_multi_fasta_quirks():
    txt = ["ACGT\n", ">chr1 desc\n", "acgt\n", "NNa\n", ">chr2\n", ">chr3\n", "tt\n"]
    assert list(_read_multi_fasta(txt)) == [("chr1 desc", "ACGTNNA"), ("chr3", "TT")] or \
        list(_read_multi_fasta(txt)) == [("chr1 desc", "ACGTNNA"), ("chr2", ""), ("chr3", "TT")]
    with pytest.raises(IndexError):
        list(_read_multi_fasta([">a\n", "\n"]))


def test_read_multi_fasta_matches_oracle(orc):
    txt = ["ACGT\n", ">chr1 desc\n", "acgt\n", "NNa\n", ">chr2\n", ">chr3\n", "tt\n", ">\n", "AC\n"]
    assert list(_read_multi_fasta(txt)) == orc.read_multi_fasta(txt)


# ------------------------------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    from deepgrp_amd import _lib
    header = open(os.path.join(ROOT, "include", "deepgrp_hip.h")).read()
    declared = set(re.findall(r"\b(dgrp_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/deepgrp_hip.h but not exported"
    assert declared == set(_lib.exported_symbols())
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (dgrp_[a-z0-9_]+)", nm))
    assert exported == declared
    assert L.dgrp_abi_version() == 1
    assert L.dgrp_window_count(1000, 200, 50) == 16 and L.dgrp_window_count(200, 200, 50) == 0


def test_host_only_entry_points():
    import ctypes as C
    from deepgrp_amd import _lib
    L = _lib.lib()
    st, kept = C.c_int64(), C.c_int64()
    raw = np.frombuffer(b"NNACGTNANN", np.uint8)
    assert L.dgrp_strip_n(raw.ctypes.data_as(C.c_void_p), raw.size, C.byref(st), C.byref(kept)) == 0
    assert (st.value, kept.value) == (2, 6)
    raw = np.frombuffer(b"NNNN", np.uint8)
    L.dgrp_strip_n(raw.ctypes.data_as(C.c_void_p), raw.size, C.byref(st), C.byref(kept))
    assert kept.value < 0
    assert L.dgrp_mss_workspace_bytes(1000) > 0 and L.dgrp_segments_workspace_bytes(1000) > 0
    # argument errors come back as codes with a message, not crashes
    assert L.dgrp_scores(None, 10, 99, None, None, None) == -1
    assert b"bad n/C" in L.dgrp_last_error()


def test_no_cpu_fallback():
    """Without a GPU the product path raises instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from deepgrp_amd import pipeline, sequence, mss, prediction
    w = synthetic.synthetic_weights(8, 5, False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pipeline.DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, 20)
    with pytest.raises(RuntimeError):
        sequence.one_hot_encode_dna_sequence("ACGT")
    with pytest.raises(RuntimeError):
        mss.find_mss_labels(np.ones(4), np.ones(4, np.int64), 3, 0, 0)
    with pytest.raises(RuntimeError):
        prediction.apply_mss(np.ones((4, 5), np.float32), dgmodel.Options())


def test_product_never_touches_the_oracle():
    for dirpath, _dirs, files in os.walk(os.path.join(ROOT, "deepgrp_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in txt.lower() or fn == "Makefile" and False, f"{fn} mentions the oracle"


# ------------------------------------------------------------------------------------------ multi-GPU host logic
def test_shard_contigs():
    from deepgrp_amd.distributed import shard_contigs
    lengths = [250, 10, 240, 30, 100, 100, 5, 90]
    for world in (1, 2, 3, 8, 16):
        parts = shard_contigs(lengths, world)
        assert len(parts) == world and sorted(i for p in parts for i in p) == list(range(len(lengths)))
        loads = [sum(lengths[i] for i in p) for p in parts]
        assert max(loads) <= max(max(lengths), -(-sum(lengths) // world) + max(lengths) // 2 + 60)
    assert shard_contigs([5, 5, 5, 5], 2) == [[0, 2], [1, 3]]


def _gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepgrp_amd.distributed import predict_records_sharded
    from deepgrp_amd.pipeline import SEGMENT_DTYPE
    records = [(f"chr{i}", "A" * n) for i, n in enumerate([50, 10, 40, 30, 5])]

    def run_one(seq, idx):                      # stands in for the device pipeline: 2 rows per record
        rows = np.zeros(2 if len(seq) > 5 else 0, SEGMENT_DTYPE)
        for j in range(rows.shape[0]):
            rows[j] = (j * 7 + idx, j * 7 + idx + len(seq), 1 + j, idx)
        return rows

    out = predict_records_sharded(records, run_one, torch.device("cpu"))
    q.put((rank, out.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_records_gloo_world2():
    """N > 1 path on CPU: contig sharding + the record gather, world_size 2 over gloo."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1] == []
    rows = got[0]
    assert [r[3] for r in rows] == [0, 0, 1, 1, 2, 2, 3, 3]          # ordered by record, record 4 contributed none
    assert rows[0] == (0, 50, 1, 0) and rows[3] == (8, 18, 2, 1)


def test_synthetic_inputs_are_deterministic():
    a = synthetic.synthetic_chromosome(50_000, contig=3)
    assert a == synthetic.synthetic_chromosome(50_000, contig=3) and a != synthetic.synthetic_chromosome(50_000, contig=4)
    assert a[:10_000] == b"N" * 10_000 and a[-10_000:] == b"N" * 10_000 and set(a) <= set(b"ACGTN")
    idx, truth = synthetic.synthetic_truth(50_000, contig=3)
    assert idx.shape == truth.shape == (50_000,) and truth.max() <= 4 and (truth > 0).mean() > 0.05
    w = synthetic.trained_weights()
    assert w["kernel"].shape == (5, 384) and w["recurrent_kernel"].shape == (128, 384) and w["ff_kernel"].shape == (128, 5)


@pytest.mark.parametrize("begin", (3, 5, 50))
@pytest.mark.parametrize("startpos", [0, 10, 22, 33])
@pytest.mark.parametrize("endpos", [0, 44, 54, 62])
@pytest.mark.parametrize("label", [1, 2, 3])
def test_get_segments_reference_test(begin, startpos, endpos, label):
    """tests/test_sequence.py:30-44 of the reference on the package's host mirror."""
    from deepgrp_amd import sequence as dgseq
    data = np.zeros(100, dtype=np.int64)
    if endpos > 0:
        data[startpos:endpos] = label
    st, en, lb = dgseq.get_segments(data, begin)
    assert st == (max(startpos, begin) if endpos > begin else 99)
    assert en == (endpos if endpos > begin else 100)
    assert lb == (label if endpos > begin else 0)


def test_get_segments_walk_matches_golden(orc):
    from conftest import golden
    from deepgrp_amd import sequence as dgseq
    g = golden("segments.npz")
    for k in range(int(g["count"])):
        lab = g[f"lab{k}"]
        i, walked = 0, []
        while i < lab.size:
            st, en, lb = dgseq.get_segments(lab, i)
            assert (st, en, lb) == orc.get_segments(lab, i)
            i = en
            walked.append((st + 5, en + 5, lb))
        np.testing.assert_array_equal(np.array(walked, np.int64).reshape(-1, 3), g[f"all{k}"])


def test_window_share_and_placement_rows(orc):
    """Host arithmetic of the intra-record split: shares tile the windows, and every share's row range
    contains exactly the rows the oracle's placement gives its windows."""
    from deepgrp_amd.distributed import placement_rows, window_share
    for nwin, world in ((0, 3), (5, 4), (16, 2), (1000, 8), (999_996, 8), (77, 5)):
        cover = []
        for r in range(world):
            a, b = window_share(nwin, world, r)
            assert 0 <= a <= b <= nwin and (a % 16 == 0 or a == nwin)
            cover += list(range(a, b)) if nwin < 5000 else []
        if nwin < 5000:
            assert cover == list(range(nwin))
    T, s = 200, 50
    for nwin, B in ((100, 7), (1000, 256), (37, 50), (64, 16)):
        for world in (1, 2, 3, 8):
            for r in range(world):
                a, b = window_share(nwin, world, r)
                lo, hi = placement_rows(a, b, nwin, B, s, T)
                rows = [orc.place_row(w, nwin, B, s) for w in range(a, b)]
                if rows:
                    assert lo == min(rows) and hi == max(rows) + T
                else:
                    assert (lo, hi) == (0, 0)


def test_fast_fasta_reader_equals_reference_loop(tmp_path):
    """deepgrp_amd.fasta.read_multi_fasta_file (bytes.translate fast path + per-record fallback) against
    the reference's line loop (deepgrp/__main__.py:20-43) incl. where exceptions are raised."""
    from deepgrp_amd.fasta import read_multi_fasta_file, read_multi_fasta_lines
    rng = np.random.default_rng(0)

    def seq(n):
        return "".join(rng.choice(list("ACGTNacgtn"), size=n))

    def wrap(s, w=60, nl="\n"):
        return nl.join(s[i:i + w] for i in range(0, len(s), w))

    cases = [">a\n" + wrap(seq(500)) + "\n>b desc\n" + wrap(seq(130)) + "\n", "junk\nACGT\n>a\n" + wrap(seq(100)) + "\n",
             ">a\r\n" + wrap(seq(200), 60, "\r\n") + "\r\n>b\r\nAC\r\n", ">a\n" + wrap(seq(100)) + "\n\n>b\nAC\n",
             ">a\nAC GT\n  >b\nTT\n>c\n\tGG \n", ">a\nACGT", ">a\nACGT\n>\nGG\n>c\nTT\n", ">a\n>b\n>c\nA\n", "",
             ">only header\n", ">a\nAC\rGT\n", ">a\nACGT\n\n", "\n>a\nAC\n", ">a\n\xc3\xa4CGT\n",
             ">h\rnn", ">h x\rACGT\n>b\nTT\n", ">h\r\nAC\r\n", ">h\r",
             # a Latin-1 byte in a header / in a body: not UTF-8, the reference's text-mode open raises UnicodeDecodeError -- and so does this
             ">a\nACGT\n>h\xe4 x\nACGT\n", ">a\nAC\xe4GT\n"]
    for i, text in enumerate(cases):
        path = tmp_path / f"c{i}.fa"
        path.write_bytes(text.encode("latin-1"))

        def run(fn):
            out, err = [], None
            try:
                for h, s in fn():
                    out.append((h, s.decode() if isinstance(s, bytes) else s))
            except Exception as e:      # noqa: BLE001
                err = type(e).__name__
            return out, err

        def ref():
            with open(path, "r") as fh:
                yield from read_multi_fasta_lines(fh)

        got, want = run(lambda: read_multi_fasta_file(str(path))), run(ref)
        if want[1] == "UnicodeDecodeError":
            # text mode decodes a buffer of 8 KiB ahead of the line it hands out, so WHICH records the reference still yields in front
            # of an undecodable byte depends on its buffering; this reader decodes record by record: the same error, and every record
            # the reference got out is one of ours, in order
            assert got[1] == want[1] and got[0][:len(want[0])] == want[0], f"case {i}: {text[:30]!r}"
            continue
        assert got == want, f"case {i}: {text[:30]!r}"


def test_record_runner_grouping_and_order():
    """RecordRunner.work_items / in_order without a GPU: consecutive short device records of ONE ingest buffer form a
    batch, anything else (other buffer, long record, text record, empty or all-N record) cuts it; results come back in
    input order and an exception surfaces at its position."""
    from deepgrp_amd.fasta import DeviceRecord
    from deepgrp_amd import runner as rn

    class FakeModel:
        vecsize, units, classes, attention = 20, 32, 5, False

    class FakePipe:
        model, step = FakeModel(), 4
        def batchable(self):
            return True
        def run_batch(self, base, offsets, lengths, startposes, contigs):
            out = np.zeros(len(lengths), dtype=[("start", "<i8"), ("end", "<i8"), ("label", "<i4"), ("contig", "<i4")])
            out["start"], out["end"], out["label"], out["contig"] = offsets, lengths, 1, contigs
            return out
        def run_idx(self, d_idx, startpos, contig=0):
            return np.array([(startpos, startpos + 1, 2, contig)], dtype=[("start", "<i8"), ("end", "<i8"), ("label", "<i4"), ("contig", "<i4")])
        def run(self, seq, contig=0):
            if seq == "boom":
                raise RuntimeError("boom")
            return self.run_idx(None, len(seq), contig)

    class Buf:                                              # stands in for a device tensor: only identity and slicing matter
        def __getitem__(self, _s):
            return self
        def numel(self):
            return 0

    a, b = Buf(), Buf()
    recs = [("r0", DeviceRecord(0, None, 100, a, 0)), ("r1", DeviceRecord(2, None, 50, a, 200)),
            ("r2", DeviceRecord(0, None, 70, b, 0)),                          # other buffer: new batch
            ("r3", "ACGT"),                                                   # text record: single
            ("r4", DeviceRecord(0, None, 30, b, 100)), ("r5", DeviceRecord(0, None, rn.SMALL_RECORD + 1, b, 200)),   # long: single
            ("r6", DeviceRecord(0, None, 0, b, 300)),                          # empty: single
            ("r7", DeviceRecord(1, None, 10, b, 400)), ("r8", DeviceRecord(1, None, 10, b, 500))]
    r = rn.RecordRunner(FakePipe(), workers=3)
    items = list(r.work_items(recs))
    shape = [("batch", [kk for kk, _ in v]) if k is rn._BATCH else (k, None) for k, v in items]
    assert shape == [("batch", ["r0", "r1"]), ("batch", ["r2"]), ("r3", None), ("batch", ["r4"]), ("r5", None), ("r6", None),
                     ("batch", ["r7", "r8"])]
    out = list(r.results(recs))
    assert [(kind, key) for kind, key, _rows in out] == [("batch", ["r0", "r1"]), ("batch", ["r2"]), ("one", "r3"), ("batch", ["r4"]),
                                                          ("one", "r5"), ("one", "r6"), ("batch", ["r7", "r8"])]
    assert out[0][2]["start"].tolist() == [0, 200] and out[0][2]["contig"].tolist() == [0, 1]
    # an all-N record raises in place: everything before it is delivered first
    bad = recs[:3] + [("n", DeviceRecord(4, None, -4, b, 0))] + recs[3:]
    got = []
    with pytest.raises(ValueError, match="negative dimensions"):
        for kind, key, _rows in r.results(bad):
            got.append(key)
    assert got == [["r0", "r1"], ["r2"]]
    with pytest.raises(RuntimeError, match="boom"):
        list(r.results([("x", "ACGT"), ("y", "boom"), ("z", "AC")]))
    # a record whose header IS the word "batch" (the key slot once carried that string for batches): on its own as text
    # (stdin / odd records), as a long device record, and inside a batch -- every header the reference accepts works here
    named = [("batch", "ACGT"), ("batch", DeviceRecord(0, None, rn.SMALL_RECORD + 1, b, 0)), ("batch", DeviceRecord(0, None, 5, b, 9)),
             ("other", DeviceRecord(0, None, 6, b, 20))]
    res = list(r.results(named))
    assert [(kind, key) for kind, key, _rows in res] == [("one", "batch"), ("one", "batch"), ("batch", ["batch", "other"])]
    assert res[0][2]["start"].tolist() == [4] and res[2][2]["contig"].tolist() == [0, 1]
    # TSV text of a batch = the per-record texts one after the other
    rows = out[0][2]
    assert rn.rows_text_batch("f.fa", ["r0", "r1"], rows) == rn.rows_text("f.fa", "r0", rows[:1]) + rn.rows_text("f.fa", "r1", rows[1:])


def test_rows_text_is_the_reference_format():
    """__main__.py:291-292: '\t'.join(str(x) for x in (filename, header, start, end, label)) per row -- dgrp_format_rows (host code
    of the library) against exactly that, incl. 19-digit coordinates, a non-ASCII header and per-row records of a batch."""
    from deepgrp_amd import runner as rn
    from deepgrp_amd.pipeline import SEGMENT_DTYPE
    rng = np.random.default_rng(3)
    n = 5000
    rows = np.zeros(n, SEGMENT_DTYPE)
    rows["start"] = rng.integers(0, 1 << 62, n) >> rng.integers(0, 62, n)
    rows["end"] = rows["start"] + rng.integers(1, 1 << 40, n)
    rows["label"] = rng.integers(1, 5, n)
    rows["contig"] = rng.integers(0, 3, n)
    heads = ["chr1 any text", "sp|Q9Ünï|x", "c"]
    want = "".join("\t".join(str(x) for x in ("a b/f.fa", heads[r["contig"]], r["start"], r["end"], r["label"])) + "\n" for r in rows)
    assert rn.rows_text_batch("a b/f.fa", heads, rows) == want
    one = rows[rows["contig"] == 1]
    assert rn.rows_text("a b/f.fa", heads[1], one) == "".join(
        "\t".join(str(x) for x in ("a b/f.fa", heads[1], r["start"], r["end"], r["label"])) + "\n" for r in one)
    assert rn.rows_text("f", "h", rows[:0]) == "" and rn.rows_text_batch("f", heads, rows[:0]) == ""
    bad = rows[:2].copy()
    bad["contig"] = 7
    with pytest.raises(RuntimeError, match="names record 7 of 3"):
        rn.rows_text_batch("f", heads, bad)


def test_packaging_console_scripts():
    """pyproject.toml declares the reference's three commands (/root/reference pyproject.toml:34-37) on this package's callables."""
    import importlib
    import tomli
    with open(os.path.join(ROOT, "pyproject.toml"), "rb") as fh:
        meta = tomli.load(fh)
    scripts = meta["project"]["scripts"]
    assert set(scripts) == {"deepgrp", "parse_rm", "preprocess_sequence"}
    assert scripts["deepgrp"] == "deepgrp_amd.__main__:main"
    for target in scripts.values():
        mod, fn = target.split(":")
        assert callable(getattr(importlib.import_module(mod), fn))
    assert "deepgrp_amd" in meta["tool"]["setuptools"]["packages"]


def test_cli_grammar_both_forms():
    """`deepgrp <modelfile> <fastafile>` (README.rst:94) and `deepgrp [flags] predict <model> <FASTA>...` parse to
    the same command; flags keep the reference's defaults (__main__.py:103-148)."""
    from deepgrp_amd.__main__ import CommandLineParser
    a = CommandLineParser().parse_args(["m.hdf5", "x.fa"]).args
    b = CommandLineParser().parse_args(["predict", "m.hdf5", "x.fa"]).args
    c = CommandLineParser().parse_args(["-b", "7", "-s", "25", "m.hdf5", "x.fa", "y.fa.gz.npz"]).args
    assert a.command == b.command == c.command == "predict" and a.model == b.model == "m.hdf5" and a.FASTA == b.FASTA == ["x.fa"]
    assert (a.batch_size, a.step_size, a.xdrop_length, a.min_mss_length, a.output, a.no_use_mss) == (256, 50, 50, 50, "-", False)
    assert (c.batch_size, c.step_size, c.FASTA) == (7, 25, ["x.fa", "y.fa.gz.npz"])


def test_split_plan_partitions_rows_and_covers_spills():
    """distributed.split_plan: the owned row ranges partition [0, n), a rank's full-batch windows start inside its own range,
    every row a window writes is owned by its rank or by one behind it (the spill transfers), and the short last batch
    lands where the reference puts it -- for window counts around the share / batch boundaries."""
    from deepgrp_amd.distributed import split_plan
    for T, s, B in ((200, 50, 256), (500, 25, 256), (40, 7, 4), (30, 300, 3)):
        for world in (1, 2, 3, 8):
            for n in (T, T + 1, T + s * 5, T + s * (B - 1) + 3, T + s * B + 1, T + s * (B * 3 + 17) + 9, T + s * (16 * world * 2 + 5)):
                nwin, first_short, shares, owned, short = split_plan(n, T, s, B, world)
                assert nwin == len(range(0, n - T, s)) and first_short == nwin // B * B
                assert owned[0][0] == 0 and owned[-1][1] == n and all(owned[k][1] == owned[k + 1][0] for k in range(world - 1))
                assert all(0 <= lo <= hi <= n for lo, hi in owned)                  # (step > T once gave a rank a negative range)
                assert shares[0][0] == 0 and shares[-1][1] == first_short and all(shares[k][1] == shares[k + 1][0] for k in range(world - 1))
                for k, (a, b) in enumerate(shares):
                    for w in (a, b - 1) if b > a else ():
                        lo, hi = w * s, min(w * s + T, n)
                        assert owned[k][0] <= lo                                    # never writes in front of its own rows
                        # rows behind its own range belong to later ranks: all covered by the ownership partition
                        assert hi <= n
                r = nwin - first_short
                if r:
                    nfull = nwin // B
                    assert short == (nfull * r * s, min((nfull * r + r - 1) * s + T, n))
                else:
                    assert short == (0, 0)


def test_split_plan_step_larger_than_window():
    """split_plan(700, 30, 300, 3, 2) returned owned = [(0, 900), (900, 700)]."""
    from deepgrp_amd.distributed import split_plan
    for n, T, s, B, world in ((700, 30, 300, 3, 2), (700, 30, 300, 3, 8), (100, 10, 95, 1, 3), (31, 30, 300, 3, 4)):
        _nwin, _fs, _shares, owned, _short = split_plan(n, T, s, B, world)
        assert owned[0][0] == 0 and owned[-1][1] == n
        assert all(0 <= lo <= hi <= n for lo, hi in owned) and all(owned[k][1] == owned[k + 1][0] for k in range(world - 1))


# ---- rank-local ingest of the sharded command line (deepgrp_amd/__main__.py::_predict_sharded)
def _write_many(path, lengths, rng, width=70):
    with open(path, "wb") as fh:
        for k, n in enumerate(lengths):
            seq = bytes(rng.choice(list(b"ACGTN"), size=n).astype(np.uint8))
            fh.write(b">r%d some text\n" % k + b"\n".join(seq[i:i + width] for i in range(0, len(seq), width)) + b"\n")


def test_chunk_starts_host_slices_concatenate(tmp_path):
    from deepgrp_amd.fasta import chunk_starts_host
    rng = np.random.default_rng(3)
    path = tmp_path / "m.fa"
    _write_many(path, [int(x) for x in rng.integers(0, 400, size=60)], rng)
    raw = path.read_bytes()
    want = [0] + [i + 1 for i in range(len(raw) - 1) if raw[i:i + 2] == b"\n>"]
    assert chunk_starts_host(str(path)).tolist() == want
    for world in (2, 3, 7, 64):
        got = np.concatenate([chunk_starts_host(str(path), r * len(raw) // world, (r + 1) * len(raw) // world) for r in range(world)])
        assert got.tolist() == want
    # slice edges ON a chunk start and directly behind one
    for cut in (want[5], want[5] + 1, want[9] - 1):
        got = np.concatenate([chunk_starts_host(str(path), 0, cut), chunk_starts_host(str(path), cut, len(raw))])
        assert got.tolist() == want
    empty = tmp_path / "e.fa"
    empty.write_bytes(b"")
    assert chunk_starts_host(str(empty)).size == 0
    nohead = tmp_path / "n.fa"
    nohead.write_bytes(b"ACGT\n>a\nAC\n")
    assert chunk_starts_host(str(nohead)).tolist() == [0, 5]


def test_plan_file_shares_partitions_the_bytes():
    from deepgrp_amd.distributed import plan_file_shares
    rng = np.random.default_rng(5)
    # (1) BASELINE configs[3]: 8 equal contigs on 8 ranks -> one each; (2) thousands of short records -> a few ranges per rank;
    # (3) mixed, two files and host-parsed extras; (4) fewer chunks than ranks
    cases = [([np.arange(8) * 1000], [8000], [], 8), ([np.arange(5000) * 100], [500000], [], 8),
             ([np.sort(rng.choice(10 ** 6, 300, replace=False)) * 1, np.array([0, 10, 5000])], [10 ** 6 + 7, 9000], [3000, 10, 70000], 3),
             ([np.array([0, 400])], [1000], [], 8), ([np.zeros(0, np.int64)], [0], [], 2)]
    for tables, sizes, extra, world in cases:
        tables = [np.concatenate([[0], t[t > 0]]).astype(np.int64) if s else t for t, s in zip(tables, sizes)]
        ranges, extras = plan_file_shares(tables, sizes, extra, world)
        assert len(ranges) == world and len(extras) == world
        assert sorted(i for e in extras for i in e) == list(range(len(extra)))
        for f, (t, size) in enumerate(zip(tables, sizes)):
            got = sorted((a, b) for r in ranges for ff, a, b in r if ff == f)
            assert all(a < b for a, b in got)
            assert [a for a, _b in got][:1] == ([0] if size else []) and all(x[1] == y[0] for x, y in zip(got, got[1:]))
            assert (got[-1][1] if got else 0) == size
            assert all(a in set(t.tolist()) for a, _b in got)                      # every range starts at a chunk start
        loads = [sum(b - a for _f, a, b in r) + sum(extra[i] for i in e) for r, e in zip(ranges, extras)]
        total = sum(sizes) + sum(extra)
        biggest = max([int(x) for t, s in zip(tables, sizes) for x in np.diff(np.concatenate([t, [s]]))] + list(extra) + [0])
        assert max(loads) <= total / world + max(biggest, total / (world * 8)) + 1
    ranges, _ = plan_file_shares([np.arange(8) * 1000], [8000], [], 8)
    assert all(len(r) == 1 and r[0][2] - r[0][1] == 1000 for r in ranges)
    ranges, _ = plan_file_shares([np.arange(5000) * 100], [500000], [], 8)
    assert max(len(r) for r in ranges) <= 8                                        # contiguous runs, not 625 scattered records


def _sharded_cli_worker(rank, world, port, fa1, fa2, out, q):
    """_predict_sharded with the device pieces replaced by host stand-ins: the planning, the order keys and the collectives are real."""
    import argparse

    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo")
    from deepgrp_amd import fasta
    from deepgrp_amd.__main__ import CommandLineParser
    from deepgrp_amd.pipeline import SEGMENT_DTYPE
    torch.cuda.current_device = lambda: 0                                      # gather_records only builds a device object from it
    read = []

    def host_ingest(path, ranges=None, *a, **k):
        raw = open(path, "rb").read()
        for lo, hi in ranges:
            read.append(hi - lo)
            fasta.UPLOAD_STATS["bytes"] += hi - lo
            starts = [lo] + [i + 1 for i in range(lo, hi - 1) if raw[i:i + 2] == b"\n>"] + [hi]
            for a_, b_ in zip(starts[:-1], starts[1:]):
                loop = fasta.LineLoop()
                for h, s in loop.feed(fasta._text_lines(raw[a_:b_]), tag=a_):
                    yield loop.last_key, h, s
                for h, s in loop.flush():
                    yield loop.last_key, h, s
    fasta.ingest_ranges = host_ingest

    class Runner:                                                               # rows = f(record): two rows per non-empty record
        def results(self, records):
            for key, seq in records:
                rows = np.zeros(2 if len(seq) else 0, SEGMENT_DTYPE)
                for j in range(len(rows)):
                    rows[j] = (j * 5, j * 5 + len(seq), 1 + seq.count("A") % 3, 0)
                yield "one", key, rows

    def records_of(name):                                                      # the host-parsed kind of input (stdin / npz)
        yield from [("x1", "ACGTA" * 7), ("x2", ""), ("x3", "AAA")]

    args = argparse.Namespace(FASTA=[fa1, "-", fa2])
    stream = open(out, "w") if rank == 0 else None
    CommandLineParser._predict_sharded(args, Runner(), records_of, stream)
    if stream:
        stream.close()
    q.put((rank, sum(read), CommandLineParser.last_sharded))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_cli_gloo_world2(tmp_path):
    """The N > 1 command line on CPU (gloo, world_size 2): rank-local byte ranges (each rank reads about half of the files'
    bytes, together all of them once), rows gathered to rank 0 and written in input order -- three inputs, the middle one of
    the host-parsed kind."""
    import torch.multiprocessing as mp
    from deepgrp_amd.fasta import read_multi_fasta_file
    rng = np.random.default_rng(11)
    fa1, fa2, out = tmp_path / "a.fa", tmp_path / "b.fa", tmp_path / "out.tsv"
    _write_many(fa1, [int(x) for x in rng.integers(0, 3000, size=120)] + [60_000, 5, 45_000], rng)
    _write_many(fa2, [20_000, 20_000, 30], rng)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33000 + os.getpid() % 2000
    procs = [ctx.Process(target=_sharded_cli_worker, args=(r, 2, port, str(fa1), str(fa2), str(out), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (n, st) for r, n, st in (q.get(timeout=120) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = os.path.getsize(fa1) + os.path.getsize(fa2)
    assert got[0][0] + got[1][0] == total and abs(got[0][0] - got[1][0]) < 0.2 * total
    assert got[0][1]["uploaded_bytes"] == got[0][0] and got[0][1]["file_bytes"] == total
    want = []
    for name, recs in ((str(fa1), read_multi_fasta_file(str(fa1))), ("-", [("x1", "ACGTA" * 7), ("x2", ""), ("x3", "AAA")]),
                       (str(fa2), read_multi_fasta_file(str(fa2)))):
        for h, s in recs:
            s = s.decode() if isinstance(s, bytes) else s
            for j in range(2 if len(s) else 0):
                want.append(f"{name}\t{h}\t{j * 5}\t{j * 5 + len(s)}\t{1 + s.count('A') % 3}\n")
    assert open(out).read() == "".join(want)


def _raise_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo")
    from deepgrp_amd.distributed import raise_together
    raise_together(None)                                                      # nobody failed: returns everywhere
    try:
        raise_together(ValueError("negative dimensions are not allowed") if rank == 1 else None)
        q.put((rank, "no error"))
    except Exception as e:                                                    # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}"))
    dist.barrier()
    dist.destroy_process_group()


def test_raise_together_gloo_world2():
    """A per-record error on one rank (the all-N ValueError of sequence.pyx:32) reaches every rank before the next
    collective instead of leaving the others blocked in the gather."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + os.getpid() % 2000
    procs = [ctx.Process(target=_raise_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1] == "ValueError: negative dimensions are not allowed"
    assert got[0].startswith("RuntimeError: rank 1 failed: ValueError")


def test_split2_isa_lint(tmp_path):
    """tools/lint_split2_isa.py (run by the Makefile on the compiler's output for gru_split2.hip, whose MFMAs are inline asm): a
    VALU write directly in front of an MFMA operand and an early reader of an MFMA result are hits, an accumulate chain, an LDS
    load into an operand and a reader 12 wait states later are not."""
    import subprocess
    tool = os.path.join(ROOT, "tools", "lint_split2_isa.py")

    def run(body):
        p = tmp_path / "k.s"
        p.write_text("_ZN1x17gru_split2_kernelILi0ELb1EEEv:\n" + "".join(f"\t{ln}\n" for ln in body) + "\ts_endpgm\n")
        r = subprocess.run([sys.executable, tool, str(p)], capture_output=True, text=True)
        return r.returncode, r.stdout

    mf = "v_mfma_f32_16x16x32_f16 v[88:91], a[24:27], v[108:111], v[88:91]"
    rc, out = run(["ds_read_b128 v[108:111], v3", "s_waitcnt lgkmcnt(0)", mf, mf, "v_add_f32_e32 v1, v2, v3"] + ["s_nop 11", "v_exp_f32_e32 v5, v88"])
    assert rc == 0 and "0 hazard(s)" in out, out
    rc, out = run(["v_mov_b64_e32 v[90:91], v[38:39]", "v_mov_b64_e32 v[88:89], v[36:37]", mf])          # the copy the allocator once placed
    assert rc == 1 and out.count("A:") == 2, out
    rc, out = run(["v_mov_b32_e32 v108, v1", "s_nop 1", mf])                                             # two wait states: enough
    assert rc == 0, out
    rc, out = run([mf, "v_add_f32_e32 v1, v2, v3", "v_exp_f32_e32 v5, v89"])                             # result read one state later
    assert rc == 1 and "B:" in out, out
    rc, out = run([mf, "v_mfma_f32_16x16x32_f16 v[4:7], a[24:27], v[88:91], v[4:7]"])                    # result as another MFMA's B operand
    assert rc == 1 and "B:" in out, out
    rc, out = run(["v_accvgpr_read_b32 v1, a3", "s_nop 4", mf])                                          # a parked value: tolerated, reported
    assert rc == 0 and "v_accvgpr" in out, out
    rc, out = run(["v_accvgpr_read_b32 v1, a3"] * 17 + ["s_nop 4", mf])                                  # copies out of the weights' AGPRs wholesale
    assert rc == 1 and "more than" in out, out
    p = tmp_path / "k.s"                                                                                  # scratch in the kernel descriptor
    p.write_text("_ZN1x17gru_split2_kernelILi0ELb1EEEv:\n\ts_endpgm\n\t.amdhsa_kernel _ZN1x17gru_split2_kernelILi0ELb1EEEv\n"
                 "\t\t.amdhsa_private_segment_fixed_size 16\n")
    r = subprocess.run([sys.executable, tool, str(p)], capture_output=True, text=True)
    assert r.returncode == 1 and "scratch" in r.stdout, r.stdout


def test_split2_schedule_is_the_generators_default(tmp_path):
    """deepgrp_amd/csrc/gru_split2_phase.inc is generated and checked in: it must be what tools/gen_split2_schedule.py writes with its
    defaults (a hand edit, or a generator change without regenerating, would go unnoticed otherwise), with all 150 MFMAs of a phase in
    program order and the barrier behind every publish link."""
    import subprocess
    out = tmp_path / "phase.inc"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_split2_schedule.py"), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = out.read_text().splitlines()[1:]                  # (the first line records the path-independent arguments; compare all the same)
    have = open(os.path.join(ROOT, "deepgrp_amd", "csrc", "gru_split2_phase.inc")).read().splitlines()[1:]
    assert have == want
    body = " ".join(have)
    mk = re.findall(r"M_K\((\d), (\d+)\)", body)
    assert mk == [(str(ks), str(n)) for ks in range(4) for n in range(36)] and len(re.findall(r"M_D\(\d\)", body)) == 6
    assert all(body.index(f"PB({g}, 4)") < body.index("BAR") for g in range(4)) and body.index("BAR") < body.index("RD0")
    for e in range(16):                                       # a sub-tile's accumulators restart only behind its chains' last reads of them
        assert body.index(f"G({e}, 4)") < body.index(f"CI(0, {e // 4})")

"""deepgrp_amd.model -- hyper-parameter bag and model loading (mirror of deepgrp/model.py).

``Options`` keeps the reference's attribute names, defaults and TOML round trip
(deepgrp/model.py:28-199).  ``load_model`` replaces ``tf.keras.models.load_model``
(deepgrp/__main__.py:264-269): it reads the Keras HDF5 file with the built-in reader
(deepgrp_amd.hdf5), checks that the layer graph is the one ``create_model``
(deepgrp/model.py:293-336) builds, and uploads the tensors to the GPU.
``create_model`` / training are out of scope (TensorFlow's job in the reference).
"""
from __future__ import annotations

import json
from typing import Any, Dict, List, Optional, TextIO

import numpy as np

from . import hdf5


class Options:
    """Hyper-parameters of a DeepGRP model; see deepgrp/model.py:28-136 for the meaning of each
    attribute.  ``gru_units`` / ``gru_dropout`` are accepted as aliases of ``units`` / ``dropout``."""

    _DEFAULTS: Dict[str, Any] = dict(
        project_root_dir=".", repeats_to_search=[1, 2, 3, 4], vecsize=150, n_epochs=200, n_batches=250,
        early_stopping_th=10, batch_size=256, repeat_probability=0.3, optimizer="RMSprop", learning_rate=0.001,
        momentum=0.9, rho=0.9, epsilon=1e-10, rnn="GRU", units=32, dropout=0.25, attention=False, min_mss_len=50,
        xdrop_len=50)

    def __init__(self, **kwargs) -> None:
        for key, value in self._DEFAULTS.items():
            self.__dict__[key] = list(value) if isinstance(value, list) else value
        self.__dict__.update(kwargs)
        self._fold_aliases()

    def _fold_aliases(self) -> None:
        units = self.__dict__.pop("gru_units", None)
        dropout = self.__dict__.pop("gru_dropout", None)
        if units:
            self.units = units
        if dropout:
            self.dropout = dropout

    def __setitem__(self, key: str, item) -> None:
        self.__dict__[key.replace("gru_", "")] = item

    def __getitem__(self, key: str):
        return self.__dict__[key.replace("gru_", "")]

    def __str__(self) -> str:
        return str(self.__dict__)

    def todict(self) -> Dict[str, Any]:
        return self.__dict__.copy()

    def fromdict(self, dictionary: Dict[str, Any]) -> None:
        self.__dict__.update(dictionary)
        self._fold_aliases()

    @classmethod
    def from_toml(cls, file: TextIO) -> "Options":
        import os
        import tomli
        if isinstance(file, (str, os.PathLike)):          # toml.load of the reference takes paths too
            with open(file, "r") as fh:
                return cls(**tomli.loads(fh.read()))
        if not hasattr(file, "read"):
            raise TypeError("You can only load a file descriptor or filename")
        text = file.read()
        if isinstance(text, bytes):
            text = text.decode("utf-8")
        return cls(**tomli.loads(text))

    def to_toml(self, file: TextIO) -> None:
        if not hasattr(file, "write"):
            raise TypeError("to_toml expects a writable file")

        def fmt(v):
            if isinstance(v, bool):
                return "true" if v else "false"
            if isinstance(v, (int, float)):
                return repr(v)
            if isinstance(v, str):
                return json.dumps(v)
            if isinstance(v, (list, tuple)):
                return "[ " + ", ".join(fmt(x) for x in v) + ",]" if v else "[]"
            raise TypeError(f"cannot write {type(v).__name__} to TOML")

        for key, value in self.__dict__.items():
            file.write(f"{key} = {fmt(value)}\n")


def _get_dna_encoding() -> List[int]:
    """Complement table of the ReverseComplement layer (deepgrp/model.py:233-237)."""
    encoding = {"A": 0, "C": 1, "G": 2, "T": 3, "N": 4}
    pairs = (("A", "T"), ("T", "A"), ("C", "G"), ("G", "C"), ("N", "N"))
    complement = {encoding[k]: encoding[v] for k, v in pairs}
    return [v for _, v in sorted(complement.items())]


class ReverseComplement:
    """Mirror of the Keras layer ``deepgrp.model.ReverseComplement`` (deepgrp/model.py:240-290): reverse the time axis
    of a one-hot batch [b, T, len(complements)] and gather the channel axis through ``complements``.  On the
    prediction path the fused kernel reads the window backwards through the same table instead of materialising this
    tensor; the class exists for callers that pass it as ``custom_objects`` or use it on arrays (numpy or torch)."""

    def __init__(self, complements: List[int], name: Optional[str] = None, trainable: bool = True, dtype: str = "float32"):
        self._indices = [int(c) for c in complements]
        self._axis = [1]
        self.name = name or "reverse_complement"
        self.trainable, self.dtype = trainable, dtype
        self.weights: List[Any] = []
        self.trainable_weights: List[Any] = []

    def build(self, input_shape) -> None:
        return None

    def call(self, inputs):
        if hasattr(inputs, "flip"):                                        # torch tensor, any device
            return inputs.flip(1)[:, :, self._indices]
        return np.asarray(inputs)[:, ::-1, :][:, :, self._indices]

    __call__ = call

    def compute_mask(self, inputs, mask=None):
        if mask is not None and not (isinstance(mask, (list, tuple)) and all(m is None for m in mask)):
            raise TypeError(f"Layer {self.name} does not support masking, but was passed an input_mask: {mask}")
        return None

    def compute_output_shape(self, input_shape):
        return input_shape

    def get_config(self) -> Dict[str, Any]:
        return {"name": self.name, "trainable": self.trainable, "dtype": self.dtype, "complements": list(self._indices)}

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "ReverseComplement":
        return cls(**config)


# --------------------------------------------------------------------------------------------
# Keras HDF5
# --------------------------------------------------------------------------------------------
class ModelFormatError(ValueError):
    pass


def _as_text(v) -> str:
    if isinstance(v, bytes):
        return v.decode("utf-8")
    if isinstance(v, np.ndarray):
        return _as_text(v.item() if v.shape == () else v.tobytes())
    return str(v)


def read_keras_hdf5(path: str) -> Dict[str, Any]:
    """Tensors and hyper-parameters of a model file written by ``model.save`` of the reference.
    Returns dict(kernel, recurrent_kernel, bias, ff_kernel, ff_bias, scale|None, vecsize, units,
    classes, attention, config)."""
    with hdf5.File(path) as f:
        if "model_config" not in f.attrs:
            raise ModelFormatError(f"{path}: no model_config attribute -- not a Keras model file")
        config = json.loads(_as_text(f.attrs["model_config"]))
        layers = config.get("config", {}).get("layers", [])
        by_class: Dict[str, List[dict]] = {}
        for layer in layers:
            by_class.setdefault(layer["class_name"].split(">")[-1], []).append(layer)
        rnn = "LSTM" if "LSTM" in by_class else "GRU"
        if len(by_class.get(rnn, [])) != 1 or "InputLayer" not in by_class or "Softmax" not in by_class:
            raise ModelFormatError("layer graph is not the one deepgrp.model.create_model builds (one shared RNN layer)")
        gru = by_class[rnn][0]
        gcfg = gru["config"]
        if not (gcfg.get("activation", "tanh") == "tanh" and gcfg.get("recurrent_activation", "sigmoid") == "sigmoid"
                and gcfg.get("use_bias", True) and not gcfg.get("go_backwards", False)):
            raise ModelFormatError(f"{rnn} layer must use Keras defaults (tanh, sigmoid, bias)")
        if rnn == "GRU" and not gcfg.get("reset_after", True):
            raise ModelFormatError("GRU layer must use reset_after=True")
        rc = by_class.get("ReverseComplement", [])
        if len(rc) != 1 or list(rc[0]["config"].get("complements", [])) != _get_dna_encoding():
            raise ModelFormatError("ReverseComplement layer with complements [3,2,1,0,4] expected")
        if len(gru.get("inbound_nodes", [])) != 2:
            raise ModelFormatError("the GRU layer must be applied twice (window and reverse complement)")
        vecsize = int(by_class["InputLayer"][0]["config"]["batch_input_shape"][1])
        units = int(gcfg["units"])
        dense = [l for l in by_class.get("Dense", []) if l["config"].get("activation", "linear") in ("linear", None)]
        if len(dense) != 1:
            raise ModelFormatError("exactly one linear Dense layer expected")
        classes = int(dense[0]["config"]["units"])
        att = by_class.get("AdditiveAttention", [])
        attention = len(att) == 1

        mw = f["model_weights"]

        def layer_weights(name: str) -> Dict[str, np.ndarray]:
            g = mw[name]
            out = {}
            names = g.attrs.get("weight_names")
            if names is not None:
                names = [names] if isinstance(names, bytes) else list(names)
                for wn in names:
                    wn = _as_text(wn)
                    out[wn.split("/")[-1]] = g[wn].read()
            else:
                for node in g.walk():
                    if node.is_dataset:
                        out[node.name.split("/")[-1]] = node.read()
            return out

        gw = layer_weights(gru["name"])
        dw = layer_weights(dense[0]["name"])
        res: Dict[str, Any] = dict(kernel=gw["kernel:0"], recurrent_kernel=gw["recurrent_kernel:0"], bias=gw["bias:0"],
                                   ff_kernel=dw["kernel:0"], ff_bias=dw["bias:0"], scale=None, vecsize=vecsize, units=units,
                                   classes=classes, attention=attention, config=config, rnn=rnn)
        if rnn == "LSTM":
            if attention:
                raise ModelFormatError("attention is only defined for the GRU model (deepgrp/model.py:308)")
            if res["bias"].reshape(-1).shape != (4 * units,):
                raise ModelFormatError(f"LSTM bias has shape {res['bias'].shape}; expected ({4 * units},)")
            return res
        if attention:
            aw = layer_weights(att[0]["name"])
            if att[0]["config"].get("use_scale", True):
                res["scale"] = np.asarray(aw["scale:0"]).reshape(-1)
            else:
                res["scale"] = np.ones(units, np.float32)
        if res["bias"].shape != (2, 3 * units):
            raise ModelFormatError(f"GRU bias has shape {res['bias'].shape}; expected (2, {3 * units}) (reset_after=True)")
        return res


def load_model(path: str, custom_objects: Optional[dict] = None):
    """Drop-in for ``tf.keras.models.load_model(path, custom_objects=...)`` on the prediction path:
    returns a device-resident model exposing ``input_shape``, ``output_shape`` and
    ``predict_on_batch``."""
    from .pipeline import DeviceModel
    w = read_keras_hdf5(path)
    model = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], w["scale"],
                        vecsize=w["vecsize"], rnn=w["rnn"])
    model.config = w["config"]
    return model


# Keras names layers "<snake_case class>" then "<...>_1", "<...>_2" ... per process (InputLayer and Model count from
# 1 / from the bare name); the reference's fixture tests/test_model.json records exactly those names.
_LAYER_UIDS: Dict[str, int] = {}


def reset_layer_names() -> None:
    """What ``tf.keras.backend.clear_session()`` does to the name counters."""
    _LAYER_UIDS.clear()


def _unique_name(prefix: str, uids: Dict[str, int], first_is_bare: bool = True) -> str:
    n = uids.get(prefix, 0)
    uids[prefix] = n + 1
    if first_is_bare:
        return prefix if n == 0 else f"{prefix}_{n}"
    return f"{prefix}_{n + 1}"


def _functional_config(vecsize: int, units: int, classes: int, attention: bool, dropout: float, rnn: str,
                       uids: Dict[str, int]) -> dict:
    """``create_model(options).get_config()`` of the reference (deepgrp/model.py:293-336) as TensorFlow 2.5 -- the
    version pinned in the reference's poetry.lock -- serialises it; equal to tests/test_model.json["2.5"] of the
    reference key for key (tests/test_host.py)."""
    def node(*sources):
        return [[name, idx, tensor, {}] for name, idx, tensor in sources]

    def plain(class_name, name, inbound, **extra):
        return {"class_name": class_name, "config": {"name": name, "trainable": True, "dtype": "float32", **extra},
                "name": name, "inbound_nodes": inbound}

    rnn = rnn.upper()
    attention = bool(attention) and rnn == "GRU"                 # deepgrp/model.py:308: attention only with the GRU
    inp = _unique_name("input", uids, first_is_bare=False)
    rc = _unique_name("reverse_complement", uids)
    rname = "B" + rnn
    rnn_cfg: Dict[str, Any] = {
        "name": rname, "trainable": True, "dtype": "float32", "return_sequences": True, "return_state": attention,
        "go_backwards": False, "stateful": False, "unroll": False, "time_major": False, "units": int(units),
        "activation": "tanh", "recurrent_activation": "sigmoid", "use_bias": True,
        "kernel_initializer": {"class_name": "GlorotUniform", "config": {"seed": None}, "shared_object_id": 2},
        "recurrent_initializer": {"class_name": "Orthogonal", "config": {"gain": 1.0, "seed": None}, "shared_object_id": 3},
        "bias_initializer": {"class_name": "Zeros", "config": {}, "shared_object_id": 4}}
    if rnn == "LSTM":
        rnn_cfg["unit_forget_bias"] = True
    rnn_cfg.update({"kernel_regularizer": None, "recurrent_regularizer": None, "bias_regularizer": None,
                    "activity_regularizer": None, "kernel_constraint": None, "recurrent_constraint": None,
                    "bias_constraint": None, "dropout": float(dropout), "recurrent_dropout": 0.0, "implementation": 2})
    if rnn == "GRU":
        rnn_cfg["reset_after"] = True
    layers: List[dict] = [
        {"class_name": "InputLayer",
         "config": {"batch_input_shape": [None, int(vecsize), 5], "dtype": "float32", "sparse": False, "ragged": False, "name": inp},
         "name": inp, "inbound_nodes": []},
        plain("Custom>ReverseComplement", rc, [node((inp, 0, 0))], complements=_get_dna_encoding()),
        {"class_name": rnn, "config": rnn_cfg, "name": rname, "inbound_nodes": [node((inp, 0, 0)), node((rc, 0, 0))]},
    ]
    if attention:
        hidden = _unique_name("average", uids)
        reshape = _unique_name("reshape", uids)
        avg = _unique_name("average", uids)
        att = _unique_name("additive_attention", uids)
        flat = _unique_name("flatten", uids)
        rep = _unique_name("repeat_vector", uids)
        last = _unique_name("concatenate", uids)
        layers += [
            plain("Average", hidden, [node((rname, 0, 1), (rname, 1, 1))]),
            {"class_name": "Reshape", "config": {"name": reshape, "trainable": True, "batch_input_shape": [None, int(units)],
                                                 "dtype": "float32", "target_shape": [1, int(units)]},
             "name": reshape, "inbound_nodes": [node((hidden, 0, 0))]},
            plain("Average", avg, [node((rname, 0, 0), (rname, 1, 0))]),
            plain("AdditiveAttention", att, [node((reshape, 0, 0), (avg, 0, 0))], causal=False, use_scale=True, dropout=0.0),
            plain("Flatten", flat, [node((att, 0, 0))], data_format="channels_last"),
            plain("RepeatVector", rep, [node((flat, 0, 0))], n=int(vecsize)),
            plain("Concatenate", last, [node((rep, 0, 0), (avg, 0, 0))], axis=-1),
        ]
    else:
        last = _unique_name("average", uids)
        layers.append(plain("Average", last, [node((rname, 0, 0), (rname, 1, 0))]))
    soft = _unique_name("softmax", uids)
    layers += [
        plain("Dense", "FF", [node((last, 0, 0))], units=int(classes), activation="linear", use_bias=True,
              kernel_initializer={"class_name": "GlorotUniform", "config": {"seed": None}},
              bias_initializer={"class_name": "Zeros", "config": {}}, kernel_regularizer=None, bias_regularizer=None,
              activity_regularizer=None, kernel_constraint=None, bias_constraint=None),
        plain("Softmax", soft, [node(("FF", 0, 0))], axis=2),
    ]
    return {"name": _unique_name("model", uids), "layers": layers, "input_layers": [[inp, 0, 0]],
            "output_layers": [[soft, 0, 0]]}


def model_config(options: "Options") -> dict:
    """``create_model(options).get_config()`` without building anything; layer names continue this process's
    counters exactly as Keras' would (``reset_layer_names`` starts them over)."""
    return _functional_config(options.vecsize, options.units, len(options.repeats_to_search) + 1, options.attention,
                              options.dropout, options.rnn, _LAYER_UIDS)


def keras_config(vecsize: int, units: int, classes: int, attention: bool, dropout: float = 0.25, rnn: str = "GRU") -> dict:
    """The ``model_config`` attribute of a model file: the functional config under first-model-of-a-session names."""
    return {"class_name": "Functional", "config": _functional_config(vecsize, units, classes, attention, dropout, rnn, {})}


def initial_weights(options: "Options", seed: Optional[int] = None) -> Dict[str, Any]:
    """Fresh tensors from the initialisers the config above names (Keras defaults: glorot_uniform kernels, orthogonal
    recurrent kernel, zero biases -- ones for the LSTM forget gate --, glorot_uniform attention scale)."""
    rng = np.random.default_rng(seed)
    u, c = int(options.units), len(options.repeats_to_search) + 1
    rnn = options.rnn.upper()
    gates = 4 if rnn == "LSTM" else 3
    attention = bool(options.attention) and rnn == "GRU"

    def glorot(fan_in, fan_out, shape):
        limit = np.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-limit, limit, size=shape).astype(np.float32)

    def orthogonal(rows, cols):
        a = rng.standard_normal((max(rows, cols), min(rows, cols)))
        q, r = np.linalg.qr(a)
        q = q * np.sign(np.diag(r))
        return (q.T if rows < cols else q).astype(np.float32)

    if rnn == "LSTM":
        bias = np.zeros(4 * u, np.float32)
        bias[u:2 * u] = 1.0                                                    # unit_forget_bias
    else:
        bias = np.zeros((2, 3 * u), np.float32)
    rows = (2 if attention else 1) * u
    return dict(kernel=glorot(5, gates * u, (5, gates * u)), recurrent_kernel=orthogonal(u, gates * u), bias=bias,
                ff_kernel=glorot(rows, c, (rows, c)), ff_bias=np.zeros(c, np.float32),
                scale=glorot(u, u, (u,)) if attention else None, vecsize=int(options.vecsize), rnn=rnn)


def create_model(options: "Options", seed: Optional[int] = None):
    """Mirror of ``deepgrp.model.create_model`` (deepgrp/model.py:293-336) on the prediction side: a freshly
    initialised, device-resident model of that architecture (``predict_on_batch``, ``get_config``, ``save``).
    Training it is the reference's business (out of scope)."""
    from .pipeline import DeviceModel
    w = initial_weights(options, seed)
    model = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], w["scale"],
                        vecsize=w["vecsize"], rnn=w["rnn"])
    model.config = {"class_name": "Functional", "config": model_config(options)}
    return model


def save_keras_hdf5(path: str, kernel, recurrent_kernel, bias, ff_kernel, ff_bias, scale=None, vecsize: int = 200,
                    rnn: str = "GRU", config: Optional[dict] = None) -> None:
    """Write a model file in the layout ``model.save`` of the reference uses (root attribute
    ``model_config``; ``model_weights/<layer>/<weight name>`` datasets with ``weight_names``
    attributes).  Used for synthetic models in tests and benchmarks."""
    kernel = np.asarray(kernel, np.float32)
    units = int(np.asarray(recurrent_kernel).shape[0])
    classes = int(np.asarray(ff_bias).shape[0])
    attention = scale is not None
    w = hdf5.Writer()
    w.set_attr("/", "keras_version", b"2.5.0")
    w.set_attr("/", "backend", b"tensorflow")
    config = config or keras_config(vecsize, units, classes, attention, rnn=rnn)
    w.set_attr("/", "model_config", json.dumps(config).encode("utf-8"))
    rl, cell = ("BLSTM", "lstm_cell") if rnn == "LSTM" else ("BGRU", "gru_cell")
    layer_names = [layer["name"] for layer in config["config"]["layers"]]
    att_name = next((l["name"] for l in config["config"]["layers"] if l["class_name"] == "AdditiveAttention"), None)
    w.create_group("model_weights")
    w.set_attr("model_weights", "layer_names", [n.encode() for n in layer_names])
    w.set_attr("model_weights", "backend", b"tensorflow")
    w.set_attr("model_weights", "keras_version", b"2.5.0")
    tensors = {rl: [(f"{rl}/{cell}/kernel:0", kernel), (f"{rl}/{cell}/recurrent_kernel:0", recurrent_kernel),
                    (f"{rl}/{cell}/bias:0", bias)],
               "FF": [("FF/kernel:0", ff_kernel), ("FF/bias:0", ff_bias)]}
    if attention:
        tensors[att_name] = [(f"{att_name}/scale:0", np.asarray(scale, np.float32).reshape(-1))]
    for lname in layer_names:
        w.create_group(f"model_weights/{lname}")
        items = tensors.get(lname, [])
        w.set_attr(f"model_weights/{lname}", "weight_names", [n.encode() for n, _ in items])
        for wname, arr in items:
            w.create_dataset(f"model_weights/{lname}/{wname}", np.asarray(arr, np.float32))
    w.save(path)

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


def keras_config(vecsize: int, units: int, classes: int, attention: bool, dropout: float = 0.25, rnn: str = "GRU") -> dict:
    """The functional-model config ``create_model`` produces (layer list as in the reference's
    tests/test_model.json), reduced to the keys this package reads back."""
    layers: List[dict] = [
        {"class_name": "InputLayer", "name": "input_1", "inbound_nodes": [],
         "config": {"batch_input_shape": [None, vecsize, 5], "dtype": "float32", "sparse": False, "ragged": False, "name": "input_1"}},
        {"class_name": "Custom>ReverseComplement", "name": "reverse_complement", "inbound_nodes": [[["input_1", 0, 0, {}]]],
         "config": {"name": "reverse_complement", "trainable": True, "dtype": "float32", "complements": [3, 2, 1, 0, 4]}},
        {"class_name": "GRU", "name": "BGRU",
         "inbound_nodes": [[["input_1", 0, 0, {}]], [["reverse_complement", 0, 0, {}]]],
         "config": {"name": "BGRU", "trainable": True, "dtype": "float32", "return_sequences": True, "return_state": bool(attention),
                    "go_backwards": False, "stateful": False, "unroll": False, "time_major": False, "units": units,
                    "activation": "tanh", "recurrent_activation": "sigmoid", "use_bias": True, "dropout": dropout,
                    "recurrent_dropout": 0.0, "implementation": 2, "reset_after": True}},
    ]
    if rnn == "LSTM":
        layers[2] = {"class_name": "LSTM", "name": "BLSTM",
                     "inbound_nodes": [[["input_1", 0, 0, {}]], [["reverse_complement", 0, 0, {}]]],
                     "config": {"name": "BLSTM", "trainable": True, "dtype": "float32", "return_sequences": True, "return_state": False,
                                "go_backwards": False, "stateful": False, "unroll": False, "time_major": False, "units": units,
                                "activation": "tanh", "recurrent_activation": "sigmoid", "use_bias": True, "unit_forget_bias": True,
                                "dropout": dropout, "recurrent_dropout": 0.0, "implementation": 2}}
        layers.append({"class_name": "Average", "name": "average", "config": {"name": "average"},
                       "inbound_nodes": [[["BLSTM", 0, 0, {}], ["BLSTM", 1, 0, {}]]]})
        last = "average"
    elif attention:
        layers += [
            {"class_name": "Average", "name": "average", "config": {"name": "average"}, "inbound_nodes": [[["BGRU", 0, 1, {}], ["BGRU", 1, 1, {}]]]},
            {"class_name": "Reshape", "name": "reshape", "config": {"name": "reshape", "target_shape": [1, units]}, "inbound_nodes": [[["average", 0, 0, {}]]]},
            {"class_name": "Average", "name": "average_1", "config": {"name": "average_1"}, "inbound_nodes": [[["BGRU", 0, 0, {}], ["BGRU", 1, 0, {}]]]},
            {"class_name": "AdditiveAttention", "name": "additive_attention", "config": {"name": "additive_attention", "causal": False, "dropout": 0.0, "use_scale": True},
             "inbound_nodes": [[["reshape", 0, 0, {}], ["average_1", 0, 0, {}]]]},
            {"class_name": "Flatten", "name": "flatten", "config": {"name": "flatten"}, "inbound_nodes": [[["additive_attention", 0, 0, {}]]]},
            {"class_name": "RepeatVector", "name": "repeat_vector", "config": {"name": "repeat_vector", "n": vecsize}, "inbound_nodes": [[["flatten", 0, 0, {}]]]},
            {"class_name": "Concatenate", "name": "concatenate", "config": {"name": "concatenate", "axis": -1},
             "inbound_nodes": [[["repeat_vector", 0, 0, {}], ["average_1", 0, 0, {}]]]},
        ]
        last = "concatenate"
    else:
        layers.append({"class_name": "Average", "name": "average", "config": {"name": "average"},
                       "inbound_nodes": [[["BGRU", 0, 0, {}], ["BGRU", 1, 0, {}]]]})
        last = "average"
    layers += [
        {"class_name": "Dense", "name": "FF", "inbound_nodes": [[[last, 0, 0, {}]]],
         "config": {"name": "FF", "trainable": True, "dtype": "float32", "units": classes, "activation": "linear", "use_bias": True}},
        {"class_name": "Softmax", "name": "softmax", "config": {"name": "softmax", "axis": 2}, "inbound_nodes": [[["FF", 0, 0, {}]]]},
    ]
    return {"class_name": "Functional", "config": {"name": "model", "layers": layers, "input_layers": [["input_1", 0, 0]],
                                                  "output_layers": [["softmax", 0, 0]]}}


def save_keras_hdf5(path: str, kernel, recurrent_kernel, bias, ff_kernel, ff_bias, scale=None, vecsize: int = 200,
                    rnn: str = "GRU") -> None:
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
    w.set_attr("/", "model_config", json.dumps(keras_config(vecsize, units, classes, attention, rnn=rnn)).encode("utf-8"))
    rl, cell = ("BLSTM", "lstm_cell") if rnn == "LSTM" else ("BGRU", "gru_cell")
    layer_names = ["input_1", "reverse_complement", rl] + (
        ["average", "reshape", "average_1", "additive_attention", "flatten", "repeat_vector", "concatenate"] if attention else ["average"]
    ) + ["FF", "softmax"]
    w.create_group("model_weights")
    w.set_attr("model_weights", "layer_names", [n.encode() for n in layer_names])
    w.set_attr("model_weights", "backend", b"tensorflow")
    w.set_attr("model_weights", "keras_version", b"2.5.0")
    tensors = {rl: [(f"{rl}/{cell}/kernel:0", kernel), (f"{rl}/{cell}/recurrent_kernel:0", recurrent_kernel),
                    (f"{rl}/{cell}/bias:0", bias)],
               "FF": [("FF/kernel:0", ff_kernel), ("FF/bias:0", ff_bias)]}
    if attention:
        tensors["additive_attention"] = [("additive_attention/scale:0", np.asarray(scale, np.float32).reshape(-1))]
    for lname in layer_names:
        w.create_group(f"model_weights/{lname}")
        items = tensors.get(lname, [])
        w.set_attr(f"model_weights/{lname}", "weight_names", [n.encode() for n, _ in items])
        for wname, arr in items:
            w.create_dataset(f"model_weights/{lname}/{wname}", np.asarray(arr, np.float32))
    w.save(path)

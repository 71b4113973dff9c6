#!/opt/conda/bin/python3.9
"""Write small Keras-layout model files with the REAL HDF5 library (h5py 3.3 / libhdf5 1.10.6 of
the build container's conda python) so that deepgrp_amd.hdf5 -- the package's own reader -- is
tested against files it did not write.  Layout follows keras' save_model_to_hdf5 /
save_weights_to_hdf5_group (TF 2.5, a dependency of the reference, not part of its tree): root
attrs keras_version/backend/model_config as bytes, model_weights/<layer> groups with a
weight_names attribute, datasets at <layer>/<weight name>.

Run:  /opt/conda/bin/python3.9 oracle/make_h5_fixtures.py   (writes tests/golden/model_*.h5 + .npz)
Also verifies, when given --check FILE, that h5py can read a file written by deepgrp_amd.hdf5.Writer.
"""
import json
import os
import sys

import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def config(T, u, C, attention):
    layers = [
        {"class_name": "InputLayer", "name": "input_1", "inbound_nodes": [],
         "config": {"batch_input_shape": [None, T, 5], "dtype": "float32", "sparse": False, "ragged": False, "name": "input_1"}},
        {"class_name": "Custom>ReverseComplement", "name": "reverse_complement", "inbound_nodes": [[["input_1", 0, 0, {}]]],
         "config": {"name": "reverse_complement", "trainable": True, "dtype": "float32", "complements": [3, 2, 1, 0, 4]}},
        {"class_name": "GRU", "name": "BGRU", "inbound_nodes": [[["input_1", 0, 0, {}]], [["reverse_complement", 0, 0, {}]]],
         "config": {"name": "BGRU", "trainable": True, "dtype": "float32", "return_sequences": True, "return_state": attention,
                    "go_backwards": False, "stateful": False, "unroll": False, "time_major": False, "units": u, "activation": "tanh",
                    "recurrent_activation": "sigmoid", "use_bias": True, "dropout": 0.25, "recurrent_dropout": 0.0,
                    "implementation": 2, "reset_after": True}},
    ]
    if attention:
        layers += [{"class_name": "Average", "name": "average", "config": {"name": "average"}, "inbound_nodes": []},
                   {"class_name": "Reshape", "name": "reshape", "config": {"name": "reshape", "target_shape": [1, u]}, "inbound_nodes": []},
                   {"class_name": "Average", "name": "average_1", "config": {"name": "average_1"}, "inbound_nodes": []},
                   {"class_name": "AdditiveAttention", "name": "additive_attention",
                    "config": {"name": "additive_attention", "causal": False, "dropout": 0.0, "use_scale": True}, "inbound_nodes": []},
                   {"class_name": "Flatten", "name": "flatten", "config": {"name": "flatten"}, "inbound_nodes": []},
                   {"class_name": "RepeatVector", "name": "repeat_vector", "config": {"name": "repeat_vector", "n": T}, "inbound_nodes": []},
                   {"class_name": "Concatenate", "name": "concatenate", "config": {"name": "concatenate", "axis": -1}, "inbound_nodes": []}]
    else:
        layers.append({"class_name": "Average", "name": "average", "config": {"name": "average"}, "inbound_nodes": []})
    layers += [{"class_name": "Dense", "name": "FF", "inbound_nodes": [],
                "config": {"name": "FF", "trainable": True, "dtype": "float32", "units": C, "activation": "linear", "use_bias": True}},
               {"class_name": "Softmax", "name": "softmax", "config": {"name": "softmax", "axis": 2}, "inbound_nodes": []}]
    return {"class_name": "Functional", "config": {"name": "model", "layers": layers, "input_layers": [["input_1", 0, 0]],
                                                  "output_layers": [["softmax", 0, 0]]}}


def write(name, T, u, C, attention, seed, vlen_config=False):
    rng = np.random.default_rng(seed)
    lim = lambda a, b: np.sqrt(6.0 / (a + b))
    w = {"kernel": rng.uniform(-lim(5, 3 * u), lim(5, 3 * u), (5, 3 * u)),
         "recurrent_kernel": np.linalg.qr(rng.normal(size=(3 * u, u)))[0].T,
         "bias": rng.normal(scale=0.05, size=(2, 3 * u)),
         "ff_kernel": rng.uniform(-0.4, 0.4, ((2 if attention else 1) * u, C)),
         "ff_bias": rng.normal(scale=0.05, size=(C,))}
    if attention:
        w["scale"] = rng.uniform(-0.3, 0.3, (u,))
    w = {k: v.astype(np.float32) for k, v in w.items()}
    path = os.path.join(OUT, name + ".h5")
    with h5py.File(path, "w") as f:
        f.attrs["keras_version"] = b"2.5.0"
        f.attrs["backend"] = b"tensorflow"
        cfg = json.dumps(config(T, u, C, attention))
        f.attrs["model_config"] = cfg if vlen_config else cfg.encode("utf8")   # str -> variable-length utf-8 attribute
        f.attrs["training_config"] = json.dumps({"loss": "categorical_crossentropy"}).encode("utf8")
        g = f.create_group("model_weights")
        names = [l["name"] for l in config(T, u, C, attention)["config"]["layers"]]
        g.attrs["layer_names"] = [n.encode("utf8") for n in names]
        g.attrs["backend"] = b"tensorflow"
        g.attrs["keras_version"] = b"2.5.0"
        tensors = {"BGRU": [("BGRU/gru_cell/kernel:0", w["kernel"]), ("BGRU/gru_cell/recurrent_kernel:0", w["recurrent_kernel"]),
                            ("BGRU/gru_cell/bias:0", w["bias"])],
                   "FF": [("FF/kernel:0", w["ff_kernel"]), ("FF/bias:0", w["ff_bias"])]}
        if attention:
            tensors["additive_attention"] = [("additive_attention/scale:0", w["scale"])]
        for n in names:
            lg = g.create_group(n)
            items = tensors.get(n, [])
            lg.attrs["weight_names"] = [wn.encode("utf8") for wn, _ in items]
            for wn, arr in items:
                ds = lg.create_dataset(wn, arr.shape, dtype=arr.dtype)
                if arr.shape:
                    ds[:] = arr
                else:
                    ds[()] = arr
        og = f.create_group("optimizer_weights")
        og.attrs["weight_names"] = [b"training/RMSprop/iter:0"]
        og.create_dataset("training/RMSprop/iter:0", data=np.int64(1234))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), T=T, u=u, C=C, attention=attention, **w)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--check":
        with h5py.File(sys.argv[2], "r") as f:
            cfg = json.loads(f.attrs["model_config"])
            print("h5py reads it:", [l["name"] for l in cfg["config"]["layers"]][:4], list(f["model_weights"].keys())[:4])
            for k in ("BGRU/BGRU/gru_cell/kernel:0", "FF/FF/bias:0"):
                a = f["model_weights"][k][()]
                print(k, a.shape, a.dtype, float(np.abs(a).sum()))
            print("weight_names", list(f["model_weights/BGRU"].attrs["weight_names"]))
        sys.exit(0)
    os.makedirs(OUT, exist_ok=True)
    write("model_u8_T20", 20, 8, 5, False, 1)
    write("model_u60_T342_att", 342, 60, 5, True, 2)
    write("model_u16_T30_att_vlen", 30, 16, 3, True, 3, vlen_config=True)

"""What a TWO-pass split would cost: the library in place is built with -DDGRP_SPLIT_DROP=1 (no U_lo.h_hi pass) or =2 (no
U_hi.h_lo pass) -- or unchanged -- and its class probabilities are compared with the fp32 yardstick on windows spread over
a synthetic chromosome, with the benchmark's fitted model:   python tools/twopass_probe.py [Mbp] [windows] [tag]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd import synthetic
from deepgrp_amd.pipeline import DeviceModel, upload_sequence

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 20
windows = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
w = synthetic.trained_weights()
m = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, 200)
st, d_idx = upload_sequence(synthetic.synthetic_chromosome(int(mbp * 1e6)))
acc = m.check_accuracy(d_idx, 50, windows, level=1)
print(json.dumps({"tag": sys.argv[3] if len(sys.argv) > 3 else "", **{k: (round(v, 9) if isinstance(v, float) else v) for k, v in acc.items()}}))

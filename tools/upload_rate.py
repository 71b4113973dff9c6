"""Upload rate of a FASTA file in /dev/shm into HBM through the pinned slabs of deepgrp_amd.fasta (`_upload_file`).
    python tools/upload_rate.py [MB] [old_module.py]      (a second argument: another fasta.py to time beside the package's)"""
import importlib.util, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd import fasta as new
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 254
mods = [("package", new)]
if len(sys.argv) > 2:
    spec = importlib.util.spec_from_file_location("deepgrp_amd.fasta_old", sys.argv[2])
    old = importlib.util.module_from_spec(spec); old.__package__ = "deepgrp_amd"; spec.loader.exec_module(old)
    mods.append(("other", old))
path = f"/dev/shm/dgrp_upload_{os.getpid()}.bin"
data = np.random.default_rng(0).integers(0, 256, mb << 20, dtype=np.uint8)
data.tofile(path)
dev = torch.device("cuda", 0)
try:
    for name, m in mods:
        for rep in range(4):
            torch.cuda.synchronize(); t = time.perf_counter()
            d = m._upload_file(path, data.size, dev)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
            ok = bool((d.cpu().numpy() == data).all()) if rep == 0 else True
            print(f"{name}: {dt * 1e3:7.1f} ms  {data.size / dt / 1e9:5.1f} GB/s  identical={ok}", flush=True)
            del d
finally:
    os.unlink(path)

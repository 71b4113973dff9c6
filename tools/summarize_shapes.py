"""profiles/<tag>_shape_kernels.csv from the shape passes of tools/profile_r03.sh: per model shape and kernel the summed duration
(rocprofv3 --kernel-trace --stats pass) and the HBM bytes of all its dispatches (separate --pmc FETCH_SIZE / WRITE_SIZE passes of the same
command; gfx950 correction 2*FETCH + WRITE, MI355X_MICROARCH.md), against 8 TB/s; plus the SQ-counter summaries of tools/sq_shape.sh.
    python tools/summarize_shapes.py gpurun_out/r03 r03"""
import csv, glob, os, shutil, sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
newest = lambda pattern: max(glob.glob(pattern, recursive=True), key=os.path.getmtime)
clean = lambda k: k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


def counter(dirname, name):
    acc = defaultdict(float)
    with open(newest(os.path.join(src, dirname, "**", "*_counter_collection.csv"))) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == name:
                acc[clean(r["Kernel_Name"])] += float(r["Counter_Value"])
    return acc


rows = []
for namefile in sorted(glob.glob(os.path.join(src, "shape*_name.txt"))):
    i = os.path.basename(namefile)[5:-9]
    shape = open(namefile).read().strip()
    dur, calls = defaultdict(float), defaultdict(int)
    with open(newest(os.path.join(src, f"shape{i}_stats", "**", "*_kernel_trace.csv"))) as f:
        for r in csv.DictReader(f):
            k = clean(r["Kernel_Name"])
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            calls[k] += 1
    fetch, write = counter(f"shape{i}_fetch", "FETCH_SIZE"), counter(f"shape{i}_write", "WRITE_SIZE")
    for k, ns in sorted(dur.items(), key=lambda kv: -kv[1]):
        if ns < 0.01 * max(dur.values()):
            continue
        hbm = (2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024
        rows.append([shape + " (tools/bench_shapes.py, default precision, two passes)", k, calls[k], round(ns / 1e6, 3), round(hbm / 1e9, 2), round(hbm / ns, 1),
                     round(hbm / ns / 8000, 3)])
with open(os.path.join(out, f"{tag}_shape_kernels.csv"), "w", newline="") as g:
    w = csv.writer(g)
    w.writerow(["shape", "kernel", "dispatches", "total_ms", "hbm_GB=2*FETCH+WRITE (all dispatches)", "GB_per_s", "frac_of_8TBps"])
    w.writerows(rows)
print(open(os.path.join(out, f"{tag}_shape_kernels.csv")).read())
for name, dst in (("sq_defaults", "wave_sq_counters_defaults_toml"), ("sq_u36", "wave_sq_counters_u36_attention"), ("sq_cfg5", "stream64_sq_counters_cfg5")):
    p = os.path.join(src, name, "summary.txt")
    if os.path.exists(p):
        shutil.copyfile(p, os.path.join(out, f"{tag}_{dst}.txt"))
for name in ("shapes.txt", "mss_cliff.txt", "fp8_probe.txt", "l2_stream.txt", "bench_2ranks_one_gpu_gloo.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copyfile(p, os.path.join(out, f"{tag}_{name}"))

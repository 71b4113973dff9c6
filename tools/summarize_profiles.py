"""Turn the rocprofv3 output directories of tools/_prof.sh (kernel stats + one FETCH_SIZE and one WRITE_SIZE
pass) into the committed summaries under profiles/:  python tools/summarize_profiles.py gpurun_out r01"""
import csv, glob, json, os, sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")

newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)     # gpurun_out/ keeps earlier runs too
stats = newest(os.path.join(src, "prof_stats", "*", "*_kernel_stats.csv"))
with open(stats) as f, open(os.path.join(out, f"{tag}_bench50mbp_kernel_stats.csv"), "w") as g:
    g.write(f.read())

rows = []
agg = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    path = newest(os.path.join(src, "prof_" + counter.split("_")[0].lower(), "*", "*_counter_collection.csv"))
    acc = defaultdict(list)
    meta = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
            meta[r["Kernel_Name"]] = (r["VGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"])
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        rows.append((counter, k, len(v), round(sum(v) / len(v), 3)) + meta[k])
        for tagname, mode0 in (("gru_fused_kernel", "gru_fused_kernel<4, 0"), ("gru_split_kernel", "gru_split_kernel<4, 0"),
                               ("gru_split2_kernel", "gru_split2_kernel<0>")):
            if mode0 in k and len(v) >= 1:        # MODE 0 = forward + merge (MODE 1 launches belong to the accuracy check)
                # the forward of the whole chromosome is the one launch with ~1 M windows; the accuracy check of
                # bench.py adds short launches of the same kernels, so take the LARGEST dispatch of each kernel
                agg.setdefault(tagname, {})[counter] = max(v)
                agg[tagname]["kernel"] = k.replace("void ", "").replace("(gru_params)", "")
with open(os.path.join(out, f"{tag}_bench50mbp_pmc_summary.csv"), "w", newline="") as g:
    w = csv.writer(g)
    w.writerow(["counter", "kernel", "dispatches", "avg_value_KB", "vgpr", "lds_bytes", "workgroup"])
    w.writerows(rows)

windows = 999596
res = {"windows_per_launch": windows,
       "correction": "gfx950: FETCH_SIZE counts wide coalesced reads at half their bytes (MI355X_MICROARCH.md, HBM section) -> 2*FETCH_SIZE + WRITE_SIZE, KB*1024",
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --mbp 50 --steps 1 --warmup 0 --no-cpu-baseline (separate passes; the largest dispatch of each kernel = the whole-chromosome launch)",
       "kernels": {}}
for tagname, a in agg.items():
    if "FETCH_SIZE" not in a or "WRITE_SIZE" not in a:
        continue
    hbm = (2 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024
    res["kernels"][tagname] = {"kernel": a["kernel"], "FETCH_SIZE_KB": a["FETCH_SIZE"], "WRITE_SIZE_KB": a["WRITE_SIZE"],
                               "hbm_bytes_per_launch": hbm, "hbm_bytes_per_window": hbm / windows}
json.dump(res, open(os.path.join(out, f"{tag}_gru_traffic.json"), "w"), indent=1)
print(open(os.path.join(out, f"{tag}_gru_traffic.json")).read())

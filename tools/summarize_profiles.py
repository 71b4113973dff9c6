"""Turn the rocprofv3 output directories of tools/profile_r02.sh into the committed summaries under profiles/:
    python tools/summarize_profiles.py gpurun_out/r02 r02
  <tag>_bench50mbp_kernel_stats.csv   per-kernel statistics of one bench run (rocprofv3 --kernel-trace --stats)
  <tag>_bench50mbp_pmc_summary.csv    FETCH_SIZE / WRITE_SIZE per kernel (separate passes), raw KB
  <tag>_gru_traffic.json              HBM bytes per launch of the recurrent kernel, gfx950 correction applied (read by bench.py)
  <tag>_streaming_kernels.csv         the HBM-bound kernels: duration (kernel trace), 2*FETCH+WRITE bytes, fraction of 8 TB/s
  <tag>_split2_sq_counters.csv        SQ / GRBM counters of the recurrent kernel per wave-step (tools/sq_counters.sh)"""
import csv, glob, json, os, sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
newest = lambda pattern: max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def counter_table(dirname, counter):
    acc, meta = defaultdict(list), {}
    with open(newest(os.path.join(src, dirname, "**", "*_counter_collection.csv"))) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
                meta[r["Kernel_Name"]] = (r["VGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"])
    return acc, meta


def durations(dirname):
    acc = defaultdict(list)
    with open(newest(os.path.join(src, dirname, "**", "*_kernel_trace.csv"))) as f:
        for r in csv.DictReader(f):
            acc[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r.get("Grid_Size") or r["Grid_Size_X"])))
    return acc


with open(newest(os.path.join(src, "prof_stats", "**", "*_kernel_stats.csv"))) as f, open(os.path.join(out, f"{tag}_bench50mbp_kernel_stats.csv"), "w") as g:
    g.write(f.read())

rows, agg = [], {}
for counter, d in (("FETCH_SIZE", "prof_fetch"), ("WRITE_SIZE", "prof_write")):
    acc, meta = counter_table(d, counter)
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        rows.append((counter, k, len(v), round(sum(v) / len(v), 3)) + meta[k][:3])
        for tagname, mode0 in (("gru_fused_kernel", "gru_fused_kernel<4, 0"), ("gru_split_kernel", "gru_split_kernel<4, 0"),
                               ("gru_split2_kernel", "gru_split2_kernel<0")):
            if mode0 in k and v:
                # the forward of the whole chromosome is the launch with ~1 M windows: take the LARGEST dispatch of each kernel
                agg.setdefault(tagname, {})[counter] = max(v)
                agg[tagname]["kernel"] = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
with open(os.path.join(out, f"{tag}_bench50mbp_pmc_summary.csv"), "w", newline="") as g:
    w = csv.writer(g)
    w.writerow(["counter", "kernel", "dispatches", "avg_value_KB", "vgpr", "lds_bytes", "workgroup"])
    w.writerows(rows)

windows = 999596
res = {"windows_per_launch": windows,
       "correction": "gfx950: FETCH_SIZE counts wide coalesced reads at half their bytes (MI355X_MICROARCH.md, HBM section) -> 2*FETCH_SIZE + WRITE_SIZE, KB*1024",
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --mbp 50 --steps 3 --warmup 1 --no-cpu-baseline --no-extras (separate passes; the largest dispatch of each kernel = the whole-chromosome launch)",
       "kernels": {}}
for tagname, a in agg.items():
    if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
        hbm = (2 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024
        res["kernels"][tagname] = {"kernel": a["kernel"], "FETCH_SIZE_KB": a["FETCH_SIZE"], "WRITE_SIZE_KB": a["WRITE_SIZE"],
                                   "hbm_bytes_per_launch": hbm, "hbm_bytes_per_window": hbm / windows}
json.dump(res, open(os.path.join(out, f"{tag}_gru_traffic.json"), "w"), indent=1)
print(open(os.path.join(out, f"{tag}_gru_traffic.json")).read())

# ---- streaming kernels: duration from the kernel trace of the --stats pass, bytes from the two PMC passes, largest dispatch of each
if glob.glob(os.path.join(src, "stream_stats")):
    dur = durations("stream_stats")
    fetch, _ = counter_table("stream_fetch", "FETCH_SIZE")
    write, _ = counter_table("stream_write", "WRITE_SIZE")
    # the post-processing kernels of the bench passes as well (scores, MSS block statistics, segments)
    dur_b = durations("prof_stats")
    fetch_b, _ = counter_table("prof_fetch", "FETCH_SIZE")
    write_b, _ = counter_table("prof_write", "WRITE_SIZE")
    with open(os.path.join(out, f"{tag}_streaming_kernels.csv"), "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["kernel", "source", "dispatches", "median_us_largest_grid", "FETCH_SIZE_KB", "WRITE_SIZE_KB", "hbm_bytes=2*FETCH+WRITE",
                    "GB_per_s", "frac_of_8TBps"])
        for source, D, F, W_ in (("tools/bench_streaming.py", dur, fetch, write), ("bench.py --mbp 50", dur_b, fetch_b, write_b)):
            for k, v in sorted(D.items()):
                if not any(s in k for s in ("encode_kernel", "onehot_kernel", "windows_kernel", "get_max_kernel", "scores_kernel",
                                            "mss_blockstat", "seg_count", "seg_emit", "mss_vote")):
                    continue
                big = max(g_ for _d, g_ in v)
                ds = sorted(d for d, g_ in v if g_ == big)
                med = ds[len(ds) // 2]
                f_, w__ = max(F.get(k, [0])), max(W_.get(k, [0]))
                hbm = (2 * f_ + w__) * 1024
                w.writerow([k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0], source, len(v), round(med / 1e3, 2), round(f_, 1), round(w__, 1), int(hbm),
                            round(hbm / med, 1), round(hbm / med / 8000, 3)])
    print(open(os.path.join(out, f"{tag}_streaming_kernels.csv")).read())

sq = os.path.join(src, "sq", "summary.txt")
if os.path.exists(sq):
    with open(sq) as f, open(os.path.join(out, f"{tag}_split2_sq_counters.txt"), "w") as g:
        g.write(f.read())
for name in ("bench250.json", "bench50.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        with open(p) as f, open(os.path.join(out, f"{tag}_{name}"), "w") as g:
            g.write(f.read())

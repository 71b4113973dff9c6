"""Summarise the counter_collection.csv files of tools/sq_shape.sh: per kernel, every counter summed over the kernel's dispatches of the
LAST pass of the shape (bench_shapes.py runs two), next to SQ_WAVE_CYCLES (SQ counters tick once per 4 clock cycles on gfx950: x 4 = cycles).
usage: python tools/sq_shape_summary.py <dir>"""
import csv, glob, os, sys
d = sys.argv[1]
per = {}
for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        ent = per.setdefault(name, {})
        ent.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["Grid_Size"])))
for name, ent in sorted(per.items(), key=lambda kv: -sum(v for _d, v, _g in kv[1].get("SQ_WAVE_CYCLES", [(0, 0, 0)]))):
    if "SQ_WAVE_CYCLES" not in ent:
        continue
    def total(cn):
        vals = sorted(ent.get(cn, []))
        vals = vals[len(vals) // 2:]                 # second pass of the shape
        return sum(v for _d, v, _g in vals), sum(g for _d, _v, g in vals) // 64
    wc, waves = total("SQ_WAVE_CYCLES")
    if wc < 1e5:
        continue
    print(f"\n{name}: {waves} waves in the pass; counters per wave, and relative to the wave's lifetime")
    for cn in sorted(ent):
        v, _w = total(cn)
        print(f"  {cn:28s} {v / max(waves, 1):14.1f}  {v / wc:8.3f}")

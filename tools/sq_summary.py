"""Summarise the counter_collection.csv files of tools/sq_counters.sh: per launch of the largest GRU dispatch and per
wave-step (one wave's step over its row tiles).  usage: python tools/sq_summary.py <dir> [T] [windows per workgroup-wave]"""
import csv, glob, os, sys
d = sys.argv[1]
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = {}
for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gru_" not in r["Kernel_Name"] and "lstm_" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]), int(r["Workgroup_Size"]))
        ent = rows.setdefault(key, {})
        c = ent.setdefault(r["Counter_Name"], [])
        c.append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
if not rows:
    sys.exit("no GRU dispatch found")
key = max(rows, key=lambda k: k[1])
name, grid, wg = key
waves = grid // 64
print(f"kernel {name}  grid {grid} threads = {grid // wg} workgroups x {wg // 64} waves; T = {T}")
print(f"{'counter':28s} {'per launch':>16s} {'per wave-step':>14s}   ns")
for cn, vals in sorted(rows[key].items()):
    v, ns = vals[-1]                       # last = the timed launch (first is the warm-up)
    print(f"{cn:28s} {v:16.0f} {v / waves / T:14.2f}   {ns}")

#!/bin/bash
# Power / clock of the GPU while the recurrent kernel runs back to back (rocm-smi sampled every 0.25 s), real weights and all-zero
# weights (same instruction stream):  bash tools/power_trace.sh <outdir>
out=${1:-gpurun_out/power}; mkdir -p $out
for mode in real zero; do
  if [ $mode = zero ]; then export GRU_ONLY_ZERO=1; else unset GRU_ONLY_ZERO; fi
  timeout -k 10 200 python tools/gru_only.py 250 12 > $out/gru_$mode.log 2>&1 &
  pid=$!
  : > $out/smi_$mode.log
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks --showtemp --json >> $out/smi_$mode.log 2>/dev/null; echo >> $out/smi_$mode.log
    sleep 0.25
  done
  wait $pid
  tail -1 $out/gru_$mode.log
done
python3 - "$out" <<'PY'
import json, sys, re
out = sys.argv[1]
for mode in ("real", "zero"):
    pw, sclk = [], []
    for line in open(f"{out}/smi_{mode}.log"):
        line = line.strip()
        if not line.startswith("{"):
            continue
        try:
            d = json.loads(line)
        except ValueError:
            continue
        c = d.get("card0", {})
        for k, v in c.items():
            if "Power" in k and "(W)" in k:
                try: pw.append(float(v))
                except ValueError: pass
            if k.startswith("sclk clock speed"):
                m = re.search(r"(\d+)Mhz", str(v))
                if m: sclk.append(int(m.group(1)))
    busy = [p for p in pw if p > 0.5 * max(pw)] if pw else []
    print(f"{mode}: {len(pw)} samples, power under load mean {sum(busy)/max(len(busy),1):.0f} W (max {max(pw) if pw else 0:.0f}), sclk samples {sorted(set(sclk))[-5:] if sclk else None}")
PY

#!/bin/bash
# All randomised sweeps, logging to gpurun_out/ (a long run behind a pipe looks hung to gpurun): tools/fuzz_all.sh [seconds each] [seed]
secs=${1:-120}; seed=${2:-1}
mkdir -p gpurun_out
for f in fuzz_parity fuzz_mss fuzz_cli fuzz_ingest; do
  echo "== $f" | tee -a gpurun_out/fuzz_all.log
  timeout -k 10 $((secs + 120)) python tools/$f.py "$secs" "$seed" >> gpurun_out/fuzz_all.log 2>&1 || { echo "FAILED: $f" | tee -a gpurun_out/fuzz_all.log; exit 1; }
  tail -1 gpurun_out/fuzz_all.log
done

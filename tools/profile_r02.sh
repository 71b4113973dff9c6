#!/bin/bash
# The rocprofv3 passes behind profiles/r02_* (one MI355X): bash tools/profile_r02.sh ; python tools/summarize_profiles.py gpurun_out/r02 r02
# (program directly behind `--`; counters in passes of their own, without other trace domains)
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/r02
mkdir -p $out
python bench.py > $out/bench250.json 2> $out/bench250.err; tail -c 400 $out/bench250.json; echo
python bench.py --mbp 50 > $out/bench50.json 2> $out/bench50.err
B50="bench.py --mbp 50 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_stats -- python3 $B50 > $out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/prof_fetch -- python3 $B50 > $out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/prof_write -- python3 $B50 > $out/prof_write.log 2>&1
echo "bench passes done" >> $out/progress.log
# the streaming kernels (API-parity entry points): sliding one-hot encoder, get_max, encoder; scores via the bench passes above
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stream_stats -- python3 tools/bench_streaming.py > $out/stream_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/stream_fetch -- python3 tools/bench_streaming.py > $out/stream_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/stream_write -- python3 tools/bench_streaming.py > $out/stream_write.log 2>&1
cat $out/stream_stats.log | tail -8
echo "streaming passes done" >> $out/progress.log
bash tools/sq_counters.sh r02/sq 50

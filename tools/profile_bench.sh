set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -1 gpurun_out/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --mbp 50 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --mbp 50 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- python3 bench.py --mbp 50 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_write.log 2>&1
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -name "*.csv" | head -20

cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_att -- python3 tools/bench_shapes.py defaults > gpurun_out/prof_att.log 2>&1
cat gpurun_out/prof_att/*/*_kernel_stats.csv | cut -c1-160 | head -8

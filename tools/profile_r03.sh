#!/bin/bash
# The rocprofv3 passes behind profiles/r03_* (one MI355X):  bash tools/profile_r03.sh ; python tools/summarize_profiles.py gpurun_out/r03 r03 ;
# python tools/summarize_shapes.py gpurun_out/r03 r03      (program directly behind `--`; counters in passes of their own)
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/r03
mkdir -p $out
python bench.py > $out/bench250.json 2> $out/bench250.err; tail -c 300 $out/bench250.json; echo
python bench.py --mbp 50 > $out/bench50.json 2> $out/bench50.err
B50="bench.py --mbp 50 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_stats -- python3 $B50 > $out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/prof_fetch -- python3 $B50 > $out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/prof_write -- python3 $B50 > $out/prof_write.log 2>&1
echo "bench passes done" >> $out/progress.log
# model shapes of the reference (defaults.toml, its hyper-parameter space) and BASELINE configs[4]: kernel times + HBM bytes per kernel
# (per-kernel passes with the library's lanes off -- DGRP_LANE_CHUNK=0: one stream, every kernel alone on the chip; shapes.txt below is
# the product path, lanes and all)
export SHAPES_SPLIT_ONLY=1 DGRP_LANE_CHUNK=0
i=0
for shape in "defaults.toml" "u=36  T=200 s=50 attention" "u=128 T=200 s=50 attention" "cfg5"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/shape${i}_stats -- python3 tools/bench_shapes.py "$shape" > $out/shape${i}_stats.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/shape${i}_fetch -- python3 tools/bench_shapes.py "$shape" > $out/shape${i}_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/shape${i}_write -- python3 tools/bench_shapes.py "$shape" > $out/shape${i}_write.log 2>&1
    echo "$shape" > $out/shape${i}_name.txt
    echo "shape $i done" >> $out/progress.log
done
unset SHAPES_SPLIT_ONLY DGRP_LANE_CHUNK
python tools/bench_shapes.py 2>&1 | grep -v amdgpu.ids > $out/shapes.txt
echo "shapes done" >> $out/progress.log
export DGRP_LANE_CHUNK=0
bash tools/sq_shape.sh r03/sq_defaults "defaults.toml" > /dev/null 2>&1
bash tools/sq_shape.sh r03/sq_u36 "u=36  T=200 s=50 attention" > /dev/null 2>&1
bash tools/sq_shape.sh r03/sq_cfg5 "cfg5" > /dev/null 2>&1
bash tools/sq_counters.sh r03/sq 50 > /dev/null 2>&1 || true
unset DGRP_LANE_CHUNK
echo "sq done" >> $out/progress.log
MSS_CLIFF_TRACE=1 DGRP_MSS_TRACE=1 python tools/mss_cliff.py 10 check 2>&1 | grep -v amdgpu.ids > $out/mss_cliff.txt
python tools/fp8_probe.py 50 4096 2>&1 | grep -v amdgpu.ids > $out/fp8_probe.txt
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/l2_stream tools/ubench/l2_stream.hip > /dev/null 2>&1 && /tmp/l2_stream 200 > $out/l2_stream.txt 2>&1 || true
python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --backend gloo --mbp 20 --steps 3 --warmup 1 --cpu-sample-bp 100000 > $out/bench_2ranks_one_gpu_gloo.json 2> $out/bench_2ranks.err || true
echo "all done" >> $out/progress.log

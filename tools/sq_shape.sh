#!/bin/bash
# SQ / GRBM counters of every kernel of one model shape of tools/bench_shapes.py (default precision): three rocprofv3 --pmc passes, then
# one table per kernel (tools/sq_shape_summary.py).   usage: bash tools/sq_shape.sh <outdir under gpurun_out> "<shape substring>"
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp SHAPES_SPLIT_ONLY=1
out=gpurun_out/$1; shape="$2"
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p1 -- python3 tools/bench_shapes.py "$shape" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out/p2 -- python3 tools/bench_shapes.py "$shape" > $out/p2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $out/p3 -- python3 tools/bench_shapes.py "$shape" > $out/p3.log 2>&1 || true
python3 tools/sq_shape_summary.py $out | tee $out/summary.txt

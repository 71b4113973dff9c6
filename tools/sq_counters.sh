#!/bin/bash
# SQ / GRBM counters of the dominant GRU kernel: three rocprofv3 --pmc passes over tools/gru_only.py (one launch each), then
# a per-wave-step table.  usage: bash tools/sq_counters.sh <outdir under gpurun_out> [Mbp]
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/$1; mbp=${2:-50}
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p1 -- python3 tools/gru_only.py $mbp 1 > $out/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out/p2 -- python3 tools/gru_only.py $mbp 1 > $out/p2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out/p3 -- python3 tools/gru_only.py $mbp 1 > $out/p3.log 2>&1 || true
python3 tools/sq_summary.py $out | tee $out/summary.txt

#!/bin/bash
# Build a variant of libdeepgrp_hip.so with another phase schedule for same-box A/B (tools/ab.sh):
#   tools/build_variant.sh <name> "<gen_split2_schedule.py arguments>" ["extra hipcc flags"]   ->  build/ab/<name>.so
# The checked-in gru_split2_phase.inc is left untouched (the variant's schedule goes to build/ab/<name>.inc).
set -e
cd "$(dirname "$0")/.."
name=$1; gen=$2; extra=$3
mkdir -p build/ab/inc_$name
python tools/gen_split2_schedule.py $gen -o build/ab/inc_$name/gru_split2_phase.inc
cd deepgrp_amd/csrc
make -s api.o seq_kernels.o gru_kernel.o rnn_stream.o post_kernels.o mss_kernels.o fasta_kernels.o eval_kernels.o ref_kernels.o
# -I first: the variant's .inc shadows the checked-in one (the source includes it by quoted name, so copy the source next to it)
cp gru_split2.hip gru_shared.h dgrp_model.h dgrp_common.h ../../build/ab/inc_$name/
sed -i 's#"../../include/deepgrp_hip.h"#"'$(pwd)'/../../include/deepgrp_hip.h"#' ../../build/ab/inc_$name/dgrp_common.h
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -fvisibility=hidden -std=c++17 -Wall -Wno-unused-function -fno-slp-vectorize $extra \
    -c ../../build/ab/inc_$name/gru_split2.hip -o ../../build/ab/inc_$name/gru_split2.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|VGPRs:|Scratch" | head -2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ab/$name.so api.o seq_kernels.o gru_kernel.o ../../build/ab/inc_$name/gru_split2.o rnn_stream.o post_kernels.o mss_kernels.o fasta_kernels.o eval_kernels.o ref_kernels.o
echo "built build/ab/$name.so"

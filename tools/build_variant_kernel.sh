#!/bin/bash
# Build build/ab/<name>.so = the library with ONE object recompiled with extra flags, for same-box A/B (tools/ab_shapes.sh):
#   tools/build_variant_kernel.sh <name> <object, e.g. gru_kernel> "<extra hipcc flags>"
set -e
cd "$(dirname "$0")/../deepgrp_amd/csrc"
name=$1; obj=$2; extra=$3
make -s all
mkdir -p ../../build/ab/$name
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -fvisibility=hidden -std=c++17 -Wall -Wno-unused-function $extra -c $obj.hip -o ../../build/ab/$name/$obj.o
objs=""
for o in api seq_kernels gru_kernel gru_split2 gru_wave rnn_stream post_kernels mss_kernels fasta_kernels eval_kernels ref_kernels; do
    if [ $o = $obj ]; then objs="$objs ../../build/ab/$name/$o.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ab/$name.so $objs
echo "built build/ab/$name.so"

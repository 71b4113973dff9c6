#!/bin/bash
# build/ab/<name>.so = the library with gru_wave.o rebuilt with extra flags (through the Makefile's patched-assembly rule)
#   tools/build_variant_wave.sh <name> "<extra hipcc flags>"
set -e
cd "$(dirname "$0")/../deepgrp_amd/csrc"
name=$1; extra=$2
make -s all
mkdir -p ../../build/ab/$name
cp gru_wave.o ../../build/ab/$name/base.o
touch gru_wave.hip
make -s gru_wave.o GRU_EXTRA="$extra"
mv gru_wave.o ../../build/ab/$name/gru_wave.o
cp ../../build/ab/$name/base.o gru_wave.o; touch gru_wave.o
objs=""
for o in api seq_kernels gru_kernel gru_split2 gru_wave rnn_stream post_kernels mss_kernels fasta_kernels eval_kernels ref_kernels; do
    if [ $o = gru_wave ]; then objs="$objs ../../build/ab/$name/$o.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ab/$name.so $objs
echo "built build/ab/$name.so"

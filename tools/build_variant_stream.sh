#!/bin/bash
# build/ab/<name>.so = the library with rnn_stream.o rebuilt with extra flags (through the Makefile's patched-assembly rule)
#   tools/build_variant_stream.sh <name> "<extra hipcc flags>"
set -e
cd "$(dirname "$0")/../deepgrp_amd/csrc"
name=$1; extra=$2
make -s all
mkdir -p ../../build/ab/$name
cp rnn_stream.o ../../build/ab/$name/base.o
touch rnn_stream.hip
make -s rnn_stream.o GRU_EXTRA="$extra" > /dev/null
mv rnn_stream.o ../../build/ab/$name/rnn_stream.o
cp ../../build/ab/$name/base.o rnn_stream.o; touch rnn_stream.o
objs=""
for o in api seq_kernels gru_kernel gru_split2 gru_wave rnn_stream post_kernels mss_kernels fasta_kernels eval_kernels ref_kernels; do
    if [ $o = rnn_stream ]; then objs="$objs ../../build/ab/$name/$o.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ab/$name.so $objs
echo "built build/ab/$name.so"

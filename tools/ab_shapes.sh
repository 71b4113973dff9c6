#!/bin/bash
# like tools/ab.sh, for tools/bench_shapes.py <filter>: tools/ab_shapes.sh <filter> a.so b.so ...
flt="$1"; shift
for f in "$@"; do cp "$f" deepgrp_amd/libdeepgrp_hip.so; echo "== $f"; timeout -k 10 200 python tools/bench_shapes.py "$flt" 2>&1 | grep -v amdgpu.ids || exit 1; done

#!/bin/bash
# A/B timing of prebuilt library variants on ONE box: tools/ab.sh build/ab/a.so build/ab/b.so ... (each run twice, interleaved)
for rep in 1 2; do for f in "$@"; do cp "$f" deepgrp_amd/libdeepgrp_hip.so; echo -n "$f: "; timeout -k 10 120 python tools/gru_only.py 50 4 2>/dev/null | tail -1 || exit 1; done; done

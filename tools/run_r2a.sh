set -x
mkdir -p gpurun_out/r2a
cp build/ab/new.so deepgrp_amd/libdeepgrp_hip.so
timeout -k 10 120 python tools/_diag.py > gpurun_out/r2a/diag.log 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_api.py tests/test_gpu_batch.py -x -q -m gpu > gpurun_out/r2a/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2a/tests.log
tail -5 gpurun_out/r2a/tests.log
cat gpurun_out/r2a/diag.log

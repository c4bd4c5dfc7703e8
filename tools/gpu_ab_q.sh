#!/bin/bash
mkdir -p gpurun_out/abq
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
echo "== pipelined"; bash tools/abn.sh 4 "--steps 20 --warmup 5 --no-fill-legs --no-self-check" build/var_qold.so build/var_qnew.so
echo "== in line"; bash tools/abn.sh 2 "--steps 20 --warmup 5 --no-fill-legs --no-self-check --no-pipeline" build/var_qold.so build/var_qnew.so

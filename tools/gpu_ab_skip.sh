#!/bin/bash
# A/B: the renormalisation test skipped on steps whose predecessor showed both state-0 metrics <= 147 (FOA_TEST_SKIP)
mkdir -p gpurun_out/skip
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "viterbi or conv or decode or soft" 2>&1 | tail -3
echo "== pipelined"; bash tools/abn.sh 3 "--steps 20 --warmup 5 --no-fill-legs --no-self-check" build/var_skip0.so build/var_skip1.so
echo "== in line"; bash tools/abn.sh 2 "--steps 20 --warmup 5 --no-fill-legs --no-self-check --no-pipeline" build/var_skip0.so build/var_skip1.so
echo "== 1000 frames, in line"; bash tools/abn.sh 2 "--steps 20 --warmup 5 --no-fill-legs --no-self-check --no-pipeline --frames 1000" build/var_skip0.so build/var_skip1.so

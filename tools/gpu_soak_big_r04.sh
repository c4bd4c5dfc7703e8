#!/bin/bash
# one long session of the random checkers on the final build of round 4 (about twenty GPU-minutes): profiles/r04_soak.txt
mkdir -p gpurun_out/soakbig
( time timeout 3000 python3 tests/manual/stress_collide.py gpu 400000 440000 ) > gpurun_out/soakbig/collide_gpu.txt 2>&1
( time timeout 1500 python3 tests/manual/stress_viterbi.py 200000 220000 ) > gpurun_out/soakbig/viterbi.txt 2>&1
( time timeout 1500 python3 tests/manual/stress_decode.py 300000 320000 ) > gpurun_out/soakbig/decode.txt 2>&1
( time timeout 1500 python3 tests/manual/stress_stream.py 20000 30000 ) > gpurun_out/soakbig/stream.txt 2>&1
( time timeout 2400 python3 tests/manual/stress_sync.py 100000 130000 ) > gpurun_out/soakbig/sync.txt 2>&1
for f in gpurun_out/soakbig/*.txt; do echo "== $f"; grep -E "seeds|differ|Traceback|Error|real" $f | tail -3; done

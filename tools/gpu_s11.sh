#!/bin/bash
# full GPU suite on the build with the cross-check forward pass, then more seeds of the random checkers (profiles/r04_soak.txt)
mkdir -p gpurun_out/s11
timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -5
( time timeout 900 python3 tests/manual/stress_viterbi.py 80000 82000 --forward4 ) > gpurun_out/s11/viterbi_fwd4.txt 2>&1
( time timeout 900 python3 tests/manual/stress_viterbi.py 82000 84000 ) > gpurun_out/s11/viterbi.txt 2>&1
( time timeout 1500 python3 tests/manual/stress_collide.py gpu 300000 303000 ) > gpurun_out/s11/collide_gpu.txt 2>&1
( time timeout 900 python3 tests/manual/stress_stream.py 7000 8500 ) > gpurun_out/s11/stream.txt 2>&1
( time timeout 900 python3 tests/manual/stress_decode.py 94000 97000 ) > gpurun_out/s11/decode.txt 2>&1
tail -n 5 gpurun_out/s11/*.txt

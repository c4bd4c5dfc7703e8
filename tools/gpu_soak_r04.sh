#!/bin/bash
# round-4 soak: the random differential checkers at length (tests/manual/), results appended to profiles/r04_soak.txt by hand
mkdir -p gpurun_out/soak4
( time timeout 1500 python3 tests/manual/stress_collide.py gpu 100000 102000 ) > gpurun_out/soak4/collide_gpu.txt 2>&1
( time timeout 900 python3 tests/manual/stress_collide.py gpu 200000 200150 --cpp ) > gpurun_out/soak4/collide_gpu_cpp.txt 2>&1
( time timeout 900 python3 tests/manual/stress_stream.py 5000 6500 ) > gpurun_out/soak4/stream.txt 2>&1
( time timeout 900 python3 tests/manual/stress_viterbi.py 70000 73000 ) > gpurun_out/soak4/viterbi.txt 2>&1
( time timeout 900 python3 tests/manual/stress_decode.py 90000 93000 ) > gpurun_out/soak4/decode.txt 2>&1
( time timeout 900 python3 tests/manual/stress_sync.py 30000 36000 ) > gpurun_out/soak4/sync.txt 2>&1
tail -n 4 gpurun_out/soak4/*.txt

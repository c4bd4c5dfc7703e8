#!/bin/bash
mkdir -p gpurun_out/s10
( time timeout 1500 python3 tests/manual/stress_collide.py gpu 100000 102000 ) > gpurun_out/s10/collide_gpu.txt 2>&1
tail -5 gpurun_out/s10/collide_gpu.txt
( time timeout 900 python3 tests/manual/stress_sync.py 36000 40000 ) > gpurun_out/s10/sync.txt 2>&1
tail -4 gpurun_out/s10/sync.txt
timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | tail -6

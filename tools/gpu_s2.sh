#!/bin/bash
mkdir -p gpurun_out/s2
timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/s2/tests.txt
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s2/bench.json 2> gpurun_out/s2/bench.err
timeout 1500 python3 tools/bench_latency.py 60000 > gpurun_out/s2/latency.jsonl 2> gpurun_out/s2/latency.err
cat /sys/fs/cgroup/cpu.max > gpurun_out/s2/cpu_max.txt 2>&1; nproc >> gpurun_out/s2/cpu_max.txt
tail -30 gpurun_out/s2/tests.txt
cut -c1-300 gpurun_out/s2/latency.jsonl | head -8

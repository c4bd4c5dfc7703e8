#!/bin/bash
# GPU session 1 of round 4: tests, the bench line, process_samples call size / latency
mkdir -p gpurun_out/s1
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/s1/tests.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s1/bench.json 2> gpurun_out/s1/bench.err
timeout 1500 python3 tools/bench_latency.py 60000 > gpurun_out/s1/latency.jsonl 2> gpurun_out/s1/latency.err
tail -5 gpurun_out/s1/tests.txt
head -c 1500 gpurun_out/s1/bench.json; echo
cat gpurun_out/s1/latency.jsonl | cut -c1-400

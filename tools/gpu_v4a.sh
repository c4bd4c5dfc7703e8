#!/bin/bash
# bring-up of the four-states-per-lane forward pass (option "forward" = 4 through FOA_FORWARD): parity first, then the kernel's own time
mkdir -p gpurun_out/v4a
export FOA_FORWARD=4
( timeout 1200 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "viterbi or conv or decode" ) > gpurun_out/v4a/parity.txt 2>&1
tail -15 gpurun_out/v4a/parity.txt
( timeout 900 python3 tests/manual/stress_viterbi.py 70000 70300 ) > gpurun_out/v4a/stress.txt 2>&1
tail -3 gpurun_out/v4a/stress.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-fill-legs --no-sync-leg --no-self-check"
for k in 3 4 3 4; do
  FOA_FORWARD=$k timeout 600 python3 bench.py $B --no-pipeline > gpurun_out/v4a/bench_inline_$k.json 2> gpurun_out/v4a/bench_inline_$k.err
  python3 - <<PY
import json
try:
    d = json.loads(open("gpurun_out/v4a/bench_inline_$k.json").read().strip().splitlines()[-1])
    print("forward", $k, "in line: ms_per_step", d["ms_per_step"], {k: v for k, v in d.get("kernel_ms", {}).items()})
except Exception as e:
    print("forward", $k, "failed", e)
PY
done

#!/bin/bash
# the pipelined step with viterbi_v4.h's forward pass and a chain-back of viterbi_v3.h's cost (FOA_FORWARD=5: see launch_forward), against the product
mkdir -p gpurun_out/v4d
B="--steps 20 --warmup 30 --no-cpu-baseline --no-extra-legs --no-fill-legs --no-sync-leg --no-self-check"
for rep in 1 2 3; do
for k in 3 5; do
  for d in -1 3; do
  FOA_FORWARD=$k timeout 600 python3 bench.py $B --depth $d > gpurun_out/v4d/bench_${k}_${d}_$rep.json 2> gpurun_out/v4d/bench_${k}_${d}_$rep.err
  python3 - <<PY
import json
try:
    d = json.loads(open("gpurun_out/v4d/bench_${k}_${d}_$rep.json").read().strip().splitlines()[-1])
    print("forward", $k, "depth", $d, "ms_per_step", d["ms_per_step"], "runs", d.get("runs_ms_per_step"), d.get("kernel_ms"))
except Exception as e:
    print("forward", $k, "failed", e)
PY
  done
done
done

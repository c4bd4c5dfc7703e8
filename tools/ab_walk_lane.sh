#!/bin/bash
# A/B of option walk_lane (the chain-back walk on the call's lane vs on the second stream with the finish), interleaved on one lease.
# usage: tools/ab_walk_lane.sh <tag>   -> gpurun_out/ab_walk_<tag>/*.json
out=gpurun_out/ab_walk_$1; mkdir -p $out
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-sync-leg"
for rep in 1 2 3; do
  for wl in 1 0; do
    timeout 300 python3 bench.py $Q --walk-lane $wl > $out/wl${wl}_r${rep}.json 2> $out/wl${wl}_r${rep}.err < /dev/null
  done
done
for f in $out/*.json; do python3 - "$f" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['repeats']['ms_per_step'], d['repeats']['forward_ms_live'], d['config']['psdu_bit_exact'])
PY
done

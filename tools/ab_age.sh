#!/bin/bash
# usage (GPU box): tools/ab_age.sh <tag>   -- how much host delay does the pipelined loop absorb, by how far back the timings are read
out=gpurun_out/age_$1; mkdir -p $out
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-sync-leg"
for age in 2 4; do for jit in 0 300 1000 3000; do
  python3 bench.py $Q --timing-age $age --host-jitter-us $jit > $out/age${age}_jit${jit}.json 2> $out/age${age}_jit${jit}.err
done; done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$out/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["repeats"]
        print("%-16s value %8.1f  ms/step %s  fwd live/alone %s" % (os.path.basename(f)[:-5], d["value"], r["ms_per_step"], r.get("forward_live_over_alone")))
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e, open(f[:-4] + "err").read()[-400:])
PY

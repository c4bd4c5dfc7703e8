#!/bin/bash
# usage (GPU box): tools/trace_hold.sh "<bench args A>" "<bench args B>" ...   kernel-trace excerpt (three steps) of bench.py per argument set -> gpurun_out/tr_<i>.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for a in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_$i -o kt -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sync-leg --no-extra-legs $a > gpurun_out/tr_$i.json 2> gpurun_out/tr_$i.log
  f=$(find gpurun_out/tr_$i -name "*kernel_trace.csv" | head -1)
  echo "== $a"; python3 tools/trace_excerpt.py $f gpurun_out/tr_$i.csv 8 3
  rm -rf gpurun_out/tr_$i
done

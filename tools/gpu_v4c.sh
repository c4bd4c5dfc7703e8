#!/bin/bash
# forward pass alone at 2, 4 and 8 times config 2's frame count (waves per SIMD: v3 9.8 / 19.5 / 39, v4 4.9 / 9.8 / 19.5)
mkdir -p gpurun_out/v4c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for fr in 20000 40000; do
for k in 3 4; do
  export FOA_FORWARD=$k
  rocprofv3 --kernel-trace --stats -d gpurun_out/v4c/kp_${k}_$fr -o x -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-fill-legs --no-sync-leg --no-self-check --no-pipeline --frames $fr > /dev/null 2>&1
  echo "== forward $k, $fr frames"; python3 tools/rocpd_stats.py gpurun_out/v4c/kp_${k}_$fr/x_results.db | grep -E "fwd" | cut -c1-150
done
done

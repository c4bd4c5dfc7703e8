#!/bin/bash
# forward pass alone, per kernel (rocprofv3 kernel trace): viterbi_v3.h against viterbi_v4.h, 10 000 and 1 000 frames, calls in line
mkdir -p gpurun_out/v4b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for fr in 10000 1000; do
for k in 3 4; do
  export FOA_FORWARD=$k
  rocprofv3 --kernel-trace --stats -d gpurun_out/v4b/kp_${k}_$fr -o x -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --no-fill-legs --no-sync-leg --no-self-check --no-pipeline --frames $fr > /dev/null 2>&1
  echo "== forward $k, $fr frames"; python3 tools/rocpd_stats.py gpurun_out/v4b/kp_${k}_$fr/x_results.db | grep -E "kernel|fwd|dec4|tb_|data_symbols" | cut -c1-150
done
done

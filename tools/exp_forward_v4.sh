#!/bin/bash
# viterbi_v4.h (four states per lane, cross-check build) against viterbi_v3.h, as measured for profiles/r04_exp_forward_four_states_per_lane.txt.
#   gpurun -- "bash tools/exp_forward_v4.sh parity|alone|saturated|pipelined"
# FOA_FORWARD (read by the cross-check build only) selects the pass: 3, 4, or 5 = viterbi_v4.h's pass beside a chain-back of viterbi_v3.h's cost
# (launch_forward in csrc/foa_rx.hip: the first calls fill every work set with valid decisions, then the conversion kernel is left out).
export FOA_LIB=$PWD/fun_ofdm_amd/csrc/libfun_ofdm_amd_xcheck.so
out=gpurun_out/exp_forward_v4; mkdir -p $out
Q="--no-cpu-baseline --no-extra-legs --no-fill-legs --no-sync-leg --no-self-check"
line() {     # line <json file> <label>: ms_per_step and the per-kernel times of a bench line
python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms_per_step", d["ms_per_step"], d.get("kernel_ms"))
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
}
trace() {    # trace <forward> <frames>: per-kernel averages of calls in line (rocprofv3 kernel trace)
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  FOA_FORWARD=$1 rocprofv3 --kernel-trace --stats -d $out/kp_$1_$2 -o x -- python3 bench.py --steps 4 --warmup 1 $Q --no-pipeline --frames $2 > /dev/null 2>&1
  echo "== forward $1, $2 frames per call"; python3 tools/rocpd_stats.py $out/kp_$1_$2/x_results.db | grep -E "kernel|fwd|dec4" | cut -c1-150
}
case "$1" in
parity)
  ( timeout 900 python3 tests/manual/stress_viterbi.py 70000 70300 --forward4 ) 2>&1 | tail -2
  timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k four_states 2>&1 | tail -2
  ;;
alone)
  for fr in 10000 1000; do for k in 3 4; do trace $k $fr; done; done
  ;;
saturated)
  for fr in 20000 40000; do for k in 3 4; do trace $k $fr; done; done
  ;;
pipelined)
  for rep in 1 2 3; do for k in 3 5; do for d in -1 3; do
    FOA_FORWARD=$k timeout 600 python3 bench.py --steps 20 --warmup 30 $Q --depth $d > $out/bench_${k}_${d}_$rep.json 2> $out/bench_${k}_${d}_$rep.err
    line $out/bench_${k}_${d}_$rep.json "forward $k depth $d:"
  done; done; done
  ;;
*) echo "usage: $0 parity|alone|saturated|pipelined"; exit 2;;
esac

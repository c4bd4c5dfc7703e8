#!/bin/bash
# usage (GPU box): tools/ab_priorities.sh  -- does spreading the library's six streams over the three stream priorities (a pool of hardware
# queues per priority: profiles/r05_probe_queues.txt) make its speed independent of GPU_MAX_HW_QUEUES?  bench.py's default leg at three
# grid sizes, with the runtime's default of 4 queues and with 8, streams at one priority ("-"), spread with lanes 3-4 low (1), spread with
# all lanes normal (2).  Interleaved rounds on one box.  (Ran against the build that still had the FOA_EXP_PRIO switch; layout 1 is what
# foa_rx_create does since.  Today the script compares queue counts only.)
out=${1:-gpurun_out/ab_prio.txt}
: > $out
for round in 1 2; do
  for q in 4 8; do
    for p in - 1 2; do
      for f in 10000 1000 300; do
        if [ $p = - ]; then unset FOA_EXP_PRIO; else export FOA_EXP_PRIO=$p; fi
        v=$(GPU_MAX_HW_QUEUES=$q python3 bench.py --frames $f --steps $((f == 10000 ? 60 : 300)) --warmup 10 --no-cpu-baseline --no-sync-leg --no-extra-legs --no-self-check 2>/dev/null \
            | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
        echo "round $round queues $q prio $p frames $f : $v" | tee -a $out
      done
    done
  done
done
unset FOA_EXP_PRIO

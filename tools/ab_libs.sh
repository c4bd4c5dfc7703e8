#!/bin/bash
# usage (GPU box): tools/ab_libs.sh <tag> <rounds> <lib> [<lib> ...]   -- alternates builds of the library (FOA_LIB) over quick bench runs: config 2
# pipelined and in line, and a 1 000-frame batch in line (one forward-pass wave per SIMD at most: a wave's own latency)
out=gpurun_out/ablibs_$1.txt; : > $out; rounds=$2; shift; shift
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-sync-leg"
for round in $(seq $rounds); do for lib in "$@"; do for mode in "" "--no-pipeline" "--no-pipeline --frames 1000"; do
  FOA_LIB=$PWD/$lib python3 bench.py $Q $mode 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print('%-28s %-28s ms/step %s  hdr %.3f sym %.3f fwd %.3f finish %.3f  bit-exact %s' % ('$lib', '$mode' or 'pipelined', d['repeats']['ms_per_step'], k['header'], k['symbols'], k['viterbi_fwd'], k['viterbi_finish'], d['config']['psdu_bit_exact']))" >> $out
done; done; done
cat $out

#!/bin/bash
# usage (GPU box): tools/ab_trees.sh <tag> <rounds> <tree> [<tree> ...]   -- alternates whole source trees (each with its own bench.py, package and built
# library; "." = this one) over quick bench runs on ONE box: config 2 pipelined and in line.  For changes that touch the ABI, where
# tools/ab_libs.sh (FOA_LIB: one Python tree, several libraries) cannot be used.  Build the other tree in the dev container first.
out=$PWD/gpurun_out/abtrees_$1.txt; : > $out; rounds=$2; shift; shift
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-sync-leg"
for round in $(seq $rounds); do for tree in "$@"; do for mode in "" "--no-pipeline"; do
  (cd $tree && python3 bench.py $Q $mode 2>/dev/null) | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print('%-18s %-14s ms/step %s  hdr %.4f scan %.4f sym %.4f fwd %.4f finish %.4f  bit-exact %s' % ('$tree', '$mode' or 'pipelined', d['repeats']['ms_per_step'], k['header'], k['scan'], k['symbols'], k['viterbi_fwd'], k['viterbi_finish'], d['config']['psdu_bit_exact']))" >> $out
done; done; done
cat $out

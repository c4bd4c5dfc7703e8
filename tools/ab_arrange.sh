#!/bin/bash
# usage (GPU box): tools/ab_arrange.sh <tag> <lib> [<lib> ...]  -- pipeline arrangements (lanes, fe_hold, depth) per library build
out=gpurun_out/arrange_$1.txt; : > $out; shift
Q="--steps 20 --warmup 8 --no-cpu-baseline --no-extra-legs --no-sync-leg --no-self-check"
for lib in "$@"; do for args in "" "--lanes 0" "--lanes 0 --fe-hold 0" "--lanes 0 --fe-hold 2" "--lanes 0 --fe-hold 0 --depth 2" "--depth 3" "--lanes 0 --fe-hold 0 --depth 3"; do
  FOA_LIB=$PWD/$lib python3 bench.py $Q $args 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print('%-22s %-34s ms/step %s  hdr %.3f scan %.3f sym %.3f fwd %.3f finish %.3f  %s' % ('$lib', '$args' or 'default', d['repeats']['ms_per_step'], k['header'], k['scan'], k['symbols'], k['viterbi_fwd'], k['viterbi_finish'], d['config']['psdu_bit_exact']))" >> $out
done; done
cat $out

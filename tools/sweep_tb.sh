#!/bin/bash
# usage: tools/sweep_tb.sh  -> kernel times of the v3 chain-back for several (segment, overlap) settings
for sl in "576 96" "768 96" "960 96" "1152 96" "1440 96" "1920 96" "960 0" "960 192" "480 96"; do
  set -- $sl
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --tb-segment $1 --tb-overlap $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['kernel_ms']
print('S=$1 L=$2', 'finish(walk+stitch)=%.3f fwd=%.3f total=%.3f exact=%s' % (k['viterbi_finish'], k['viterbi_fwd'], k['total'], d['config']['psdu_bit_exact']))"
done

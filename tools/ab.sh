#!/bin/bash
# usage: tools/ab.sh <rounds> <lib A> <lib B> [bench args]   alternates bench.py over two builds on one box (ms per step, forward-pass ms)
n=$1; a=$2; b=$3; shift 3
for i in $(seq $n); do
  for lib in $a $b; do
    FOA_LIB=$PWD/$lib python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['kernel_ms']['viterbi_fwd'], d['config']['psdu_bit_exact'])"
  done
done

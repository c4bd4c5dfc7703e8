#!/bin/bash
# usage: tools/abn.sh <rounds> "<bench args>" <lib> [<lib> ...]   like ab.sh for any number of builds; prints ms per step and the per-kernel times
n=$1; args=$2; shift 2
for i in $(seq $n); do
  for lib in "$@"; do
    FOA_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-sync-leg --no-extra-legs $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$lib', d['ms_per_step'], 'hdr %.3f scan %.3f sym %.3f fwd %.3f fin %.3f' % (k['header'],k['scan'],k['symbols'],k['viterbi_fwd'],k['viterbi_finish']), d['config']['psdu_bit_exact'])"
  done
done

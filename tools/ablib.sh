#!/bin/bash
# usage: tools/ablib.sh <rounds> "<lib>|<bench args>" ...   alternates bench.py over (build, argument set) pairs on one box
n=$1; shift
for i in $(seq $n); do
  for spec in "$@"; do
    lib=${spec%%|*}; args=${spec#*|}
    FOA_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-sync-leg --no-extra-legs $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('[$spec]', d['ms_per_step'], 'hdr %.3f scan %.3f sym %.3f fwd %.3f fin %.3f' % (k['header'],k['scan'],k['symbols'],k['viterbi_fwd'],k['viterbi_finish']), d['config']['psdu_bit_exact'])"
  done
done

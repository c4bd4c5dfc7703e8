#!/bin/bash
# usage: tools/kprof.sh <lib.so|-> <frames> [bench args]  -> per-kernel avg us via rocprofv3 kernel trace
lib=$1; frames=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$(basename $lib .so)_$frames
[ "$lib" != "-" ] && export FOA_LIB=$PWD/$lib
rocprofv3 --kernel-trace --stats -d gpurun_out/kp_$tag -o x -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --frames $frames "$@" > /dev/null 2>&1
echo "== $tag"; python3 tools/rocpd_stats.py gpurun_out/kp_$tag/x_results.db | grep -E "fwd|finish|tb_|data_symbols" | cut -c1-120

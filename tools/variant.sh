#!/bin/bash
# usage: tools/variant.sh <name> <hipcc -D flags...>   builds build/var_<name>.so from fun_ofdm_amd/csrc with extra compiler flags;
# run with FOA_LIB=$PWD/build/var_<name>.so (A/B timing of two builds on one box: tools/ab_libs.sh)
set -e
name=$1; shift
src=fun_ofdm_amd/csrc; obj=build/var_$name.o.d
rm -rf $obj build/var_$name.so          # never link a stale object of an earlier build of this variant name
mkdir -p $obj
pids=()
for u in rx_handle rx_decode rx_sync rx_stage rx_tx rx_stream; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c -o $obj/$u.o $src/$u.hip &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "variant $name: a unit failed to compile" >&2; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var_$name.so $obj/*.o

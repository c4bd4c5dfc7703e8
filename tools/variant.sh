#!/bin/bash
# usage: tools/variant.sh <name> <hipcc -D flags...>   builds build/var_<name>.so; run with FOA_LIB=build/var_<name>.so (A/B timing on one box)
mkdir -p build
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function "$@" -o build/var_$name.so fun_ofdm_amd/csrc/foa_rx.hip

#!/bin/bash
# usage: tools/ablate.sh <mask> ...   builds build/abl_<mask>.so with -DFOA_ABL=<mask> (timing experiments; results are wrong by design)
mkdir -p build
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -DFOA_ABL=$m -o build/abl_$m.so fun_ofdm_amd/csrc/foa_rx.hip &
done
wait
ls -la build/

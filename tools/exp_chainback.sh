#!/bin/bash
# Bounds on a chain-back fused into the forward pass, measured instead of argued:
#   gain  <= what the step gains when the decision words never touch HBM at all: forward pass without its decision stores (FOA_ABL=16)
#            and no walk kernel (FOA_EXP=1) -- results wrong by design;
#   cost  >= what k extra VALU instructions per forward step cost the step (FOA_EXP=2: one, 4: two; results stay right).
# usage: tools/exp_chainback.sh build   (dev container: builds build/var_*.so)
#        tools/exp_chainback.sh run <tag>   (GPU box: alternates the builds, three rounds; gpurun_out/exp_chainback_<tag>.txt)
if [ "$1" = build ]; then
  mkdir -p build
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function"
  /opt/rocm/bin/hipcc $F -DFOA_ABL=16 -DFOA_EXP=1 -o build/var_nodec.so fun_ofdm_amd/csrc/foa_rx.hip &
  /opt/rocm/bin/hipcc $F -DFOA_EXP=1 -o build/var_nowalk.so fun_ofdm_amd/csrc/foa_rx.hip &
  /opt/rocm/bin/hipcc $F -DFOA_EXP=2 -o build/var_plus1.so fun_ofdm_amd/csrc/foa_rx.hip &
  /opt/rocm/bin/hipcc $F -DFOA_EXP=4 -o build/var_plus2.so fun_ofdm_amd/csrc/foa_rx.hip &
  wait; ls -la build/var_*.so; exit 0
fi
out=gpurun_out/exp_chainback_$2.txt; : > $out
for round in 1 2 3; do
  for lib in fun_ofdm_amd/csrc/libfun_ofdm_amd.so build/var_nowalk.so build/var_nodec.so build/var_plus1.so build/var_plus2.so; do
    for mode in "" "--no-pipeline"; do
      FOA_LIB=$PWD/$lib python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sync-leg --no-extra-legs $mode 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print('%-42s %-14s ms/step %s  fwd %.3f finish %.3f  bit-exact %s' % ('$lib', '$mode' or 'pipelined', d['repeats']['ms_per_step'], k['viterbi_fwd'], k['viterbi_finish'], d['config']['psdu_bit_exact']))" >> $out
    done
  done
done
cat $out

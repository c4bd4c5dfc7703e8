#!/bin/bash
# usage (GPU box): tools/soak_r06.sh [scale [seed offset]]  -- the random differential checkers of tests/manual/ over fresh seed ranges, run side by side
# (six processes share the GPU), results in gpurun_out/soak_r06.txt.  scale 1 = about 15 minutes.
s=${1:-1}
o=${2:-0}
out=gpurun_out/soak_r06.txt
mkdir -p gpurun_out/soak
run() { name=$1; shift; ( "$@" 2>&1 | grep -v "amdgpu.ids" | tail -4 > gpurun_out/soak/$name.txt ) & }
run collide   python3 tests/manual/stress_collide.py gpu $((900000 + o)) $((900000 + o + 12000 * s))
run decode    python3 tests/manual/stress_decode.py $((800000 + o)) $((800000 + o + 6000 * s))
run stream    python3 tests/manual/stress_stream.py $((90000 + o)) $((90000 + o + 4000 * s))
run sync      python3 tests/manual/stress_sync.py $((600000 + o)) $((600000 + o + 10000 * s))
run viterbi   python3 tests/manual/stress_viterbi.py $((700000 + o)) $((700000 + o + 8000 * s))
run tags      python3 tests/manual/stress_tags.py gpu $((60000 + o)) $((60000 + o + 6000 * s))
wait
run chain_cpp python3 tests/manual/stress_chain_cpp.py $((5000 + o)) $((5000 + o + 150 * s))
run stages    python3 tests/manual/stress_stages.py $((8000 + o)) $((8000 + o + 1500 * s))
run tx        python3 tests/manual/stress_tx.py $((8000 + o)) $((8000 + o + 1500 * s))
run pipeline  python3 tools/soak_pipeline.py $((400 * s))
wait
for f in collide decode stream sync viterbi tags chain_cpp stages tx pipeline; do echo "== $f"; cat gpurun_out/soak/$f.txt; done > $out
cat $out

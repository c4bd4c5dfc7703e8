#!/bin/bash
mkdir -p gpurun_out/s9
timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/s9/tests.txt
cat gpurun_out/s9/tests.txt
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s9/bench.json 2> gpurun_out/s9/bench.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/s9/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['bench_wall_s'], d['config']['psdu_bit_exact'])
print(d['legs']['process_samples_api'])
PY

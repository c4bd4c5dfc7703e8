#!/bin/bash
# final build of round 4: the driver's command on a cold box, then more seeds of the random checkers
mkdir -p gpurun_out/final_r04
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_r04/driver_cmd.json 2> gpurun_out/final_r04/driver_cmd.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/final_r04/driver_cmd.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "bench_wall_s")}, d["roofline"]["frac"], d["roofline"].get("frac_at_step_rate"), d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
( time timeout 900 python3 tests/manual/stress_sync.py 40000 46000 ) > gpurun_out/final_r04/sync.txt 2>&1
( time timeout 900 python3 tests/manual/stress_viterbi.py 84000 88000 ) > gpurun_out/final_r04/viterbi.txt 2>&1
( time timeout 900 python3 tests/manual/stress_viterbi.py 88000 90000 --forward4 ) > gpurun_out/final_r04/viterbi_fwd4.txt 2>&1
( time timeout 1500 python3 tests/manual/stress_collide.py gpu 303000 308000 ) > gpurun_out/final_r04/collide_gpu.txt 2>&1
( time timeout 900 python3 tests/manual/stress_stream.py 8500 10500 ) > gpurun_out/final_r04/stream.txt 2>&1
tail -n 4 gpurun_out/final_r04/*.txt

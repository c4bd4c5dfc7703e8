#!/bin/bash
# usage (on the GPU box, from the repo root, as the FIRST thing of a fresh lease):  tools/cold_ab.sh <tag> [quick]
# 1. the driver's exact command as the first GPU process of the lease (what BENCH_rNN.json is), 2. the same again (warm),
# 3. A/B of the pipeline arrangement in quick runs (no extra legs), 4. a kernel trace of the default.  Everything under
# gpurun_out/cold_<tag>/; copy what is to be judged into profiles/.
tag=$1
out=gpurun_out/cold_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
{ date; hostname; nproc; grep -m1 'model name' /proc/cpuinfo; rocm-smi --showuniqueid --showperflevel --showmaxpower --showcomputepartition --showmemorypartition --showpids; } > $out/box.txt 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/driver_cmd_1.json 2> $out/driver_cmd_1.err
rocm-smi --showclocks --showpower --showtemp >> $out/box.txt 2>&1
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-sync-leg"
run() { name=$1; shift; "$@" > $out/$name.json 2> $out/$name.err; }
run default_a python3 bench.py $Q
run lanes0_hq8 python3 bench.py $Q --lanes 0
GPU_MAX_HW_QUEUES=4 run lanes0_hq4 python3 bench.py $Q --lanes 0
GPU_MAX_HW_QUEUES=4 run lanes1_hq4 python3 bench.py $Q
run no_pipeline python3 bench.py $Q --no-pipeline
run warmup12 python3 bench.py $Q --warmup 12
run default_b python3 bench.py $Q
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py $Q > $out/under_rocprof.json 2> $out/kt.log
python3 tools/trace_excerpt.py $out/kt/kt_kernel_trace.csv $out/kernel_trace_excerpt.csv > $out/kernel_trace_excerpt.txt 2>&1
# keep the merge small: the raw trace of ~2000 launches is fine, drop anything else rocprof wrote
find $out/kt -type f ! -name '*.csv' -delete
if [ "$2" != quick ]; then
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/driver_cmd_2.json 2> $out/driver_cmd_2.err
fi
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$out/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("repeats", {})
        print("%-22s value %8.1f  ms/step %s  fwd live %s alone %s ratio %s" % (os.path.basename(f)[:-5], d["value"], r.get("ms_per_step"),
              r.get("forward_ms_live"), r.get("forward_ms_alone"), r.get("forward_live_over_alone")))
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY

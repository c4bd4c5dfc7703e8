#!/bin/bash
# round-4 profile set: rocprofv3 passes, probes, bench lines, process_samples tables
tools/profile_all.sh r04 > gpurun_out/profile_all_r04.log 2>&1
python3 tools/bench_latency.py 60000 > gpurun_out/prof_r04/r04_process_samples_latency.jsonl 2> gpurun_out/prof_r04/latency.err
ls gpurun_out/prof_r04 | head -50
python3 -c "
import json
d=json.load(open('gpurun_out/prof_r04/r04_bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_at_step_rate']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['parallel_efficiency'])
print(json.load(open('gpurun_out/prof_r04/r04_pmc_sq.json'))['per_launch']['k_viterbi_fwd3'])
"

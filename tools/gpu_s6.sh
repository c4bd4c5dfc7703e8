#!/bin/bash
mkdir -p gpurun_out/s6
timeout 1500 python3 -m pytest tests/test_gpu_stream.py tests/test_gpu_cpp_adaptors.py tests/test_gpu_fuzz.py -q 2>&1 | tail -6 > gpurun_out/s6/tests.txt
cat gpurun_out/s6/tests.txt
timeout 1500 python3 tools/bench_latency.py 60000 > gpurun_out/s6/latency.jsonl 2> gpurun_out/s6/latency.err
python3 - <<'PY'
import json
for l in open('gpurun_out/s6/latency.jsonl'):
    d=json.loads(l); print(d["mode"][:76].ljust(78), d.get("Msamples_per_s"), d.get("spread",""), d.get("latency_ms"))
PY

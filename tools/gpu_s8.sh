#!/bin/bash
mkdir -p gpurun_out/s8
timeout 1500 python3 -m pytest tests/test_gpu_stream.py -q 2>&1 | tail -5
timeout 1500 python3 tools/bench_latency.py 60000 > gpurun_out/s8/latency.jsonl 2> gpurun_out/s8/latency.err
python3 - <<'PY'
import json
for l in open('gpurun_out/s8/latency.jsonl'):
    d=json.loads(l); print(d["mode"][:84].ljust(86), d.get("Msamples_per_s"), d.get("spread",""), d.get("latency_ms"), d.get("error","")[:200])
PY

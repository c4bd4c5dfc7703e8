#!/bin/bash
# usage: tools/pmc_cmd.sh <kernel substring> <tag> <python script and args...>   -> SQ / LDS / clock counters per launch of that kernel for ANY
# command (tools/pmc_kernel.sh does the same for bench.py), each counter set in its own rocprofv3 --pmc run; prints and writes gpurun_out/<tag>.txt
pat=$1; tag=$2; shift 2
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p gpurun_out
: > gpurun_out/$tag.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); out=gpurun_out/pmcc_$(echo $tag | tr / _)_$i; rm -rf $out
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p -- python3 "$@" > /dev/null 2> $out.log || tail -3 $out.log
  python3 - "$out" "$pat" <<'PY' | tee -a gpurun_out/$tag.txt
import csv, glob, os, sys
from collections import defaultdict
out, pat = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0.0, set()])
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        low = {k.lower(): v for k, v in row.items()}
        if pat in low["kernel_name"]:
            a = acc[low["counter_name"]]; a[0] += float(low["counter_value"]); a[1].add(low.get("dispatch_id"))
dur = []
for path in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        low = {k.lower(): v for k, v in row.items()}
        if pat in low["kernel_name"]:
            dur.append((int(low["end_timestamp"]) - int(low["start_timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    print("%-28s %16.0f   (%d launches)" % (k, v[0] / max(1, len(v[1])), len(v[1])))
if dur:
    print("%-28s %16.1f   us per launch under the counters (%d launches)" % ("duration", sum(dur) / len(dur), len(dur)))
PY
done

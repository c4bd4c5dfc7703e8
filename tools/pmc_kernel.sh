#!/bin/bash
# usage: tools/pmc_kernel.sh <kernel substring> <bench args...>   -> a few SQ/LDS/TA counters per launch of that kernel
pat=$1; shift
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1)); out=gpurun_out/pmck_$i; rm -rf $out
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $out -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $out.log || tail -3 $out.log
  python3 - "$out" "$pat" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, pat = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0.0, set()])
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        low = {k.lower(): v for k, v in row.items()}
        if pat in low["kernel_name"]:
            a = acc[low["counter_name"]]; a[0] += float(low["counter_value"]); a[1].add(low.get("dispatch_id"))
for k, v in sorted(acc.items()):
    print("%-28s %16.0f" % (k, v[0] / max(1, len(v[1]))))
PY
done

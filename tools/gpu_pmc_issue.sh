#!/bin/bash
# where do the forward pass's issue slots go when its SIMDs are saturated?  SQ counters of k_viterbi_fwd3 at 40 000 frames per call (calls in line)
mkdir -p gpurun_out/pmci
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/pmci/counters.txt 2>&1
grep -c . gpurun_out/pmci/counters.txt
B="python3 bench.py --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-extra-legs --no-fill-legs --no-sync-leg --no-self-check --no-pipeline --frames 40000"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmci/p$i -o p -- $B > /dev/null 2> gpurun_out/pmci/p$i.log
  python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmci/p$i/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "viterbi_fwd3" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(acc.items()): print("%-32s %16.0f per launch (%d rows)" % (k, v / max(n, 1) , n))
PY
done

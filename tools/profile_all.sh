#!/bin/bash
# usage (on the GPU box, from the repo root):  tools/profile_all.sh <tag>
# Everything profiles/<tag>_* is made of: the rocprofv3 passes of profile_round.sh, the micro-benchmarks (binaries built in the
# dev container: build/probe_issue.bin, build/probe_renorm.bin), the bench lines (default, calls in line, two ranks on one
# device over gloo) and the process_samples() API figures.  Results under gpurun_out/prof_<tag>/; copy <tag>_* into profiles/.
tag=$1
out=gpurun_out/prof_$tag
tools/profile_round.sh $tag > gpurun_out/profile_round_$tag.log 2>&1
mkdir -p $out
for p in probe_issue probe_renorm; do
  [ -x build/$p.bin ] && timeout 300 build/$p.bin > $out/${tag}_$p.txt 2> $out/$p.err
done
[ -s $out/${tag}_probe_issue.txt ] && python3 tools/reduce_probe.py $out/${tag}_probe_issue.txt $out/${tag}_probe_issue.json > /dev/null
# bench.py reads the probe and counter files from profiles/: the lines below are written with this round's
cp $out/${tag}_pmc_sq.json $out/${tag}_pmc_hbm.json $out/${tag}_pmc_lds_l2.json profiles/ 2>/dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_default.json 2> $out/bench_default.err
FOA_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > $out/${tag}_bench_forced_rccl_world1.json 2> $out/bench_forced_rccl.err
python3 bench.py --no-pipeline --no-extra-legs > $out/${tag}_bench_no_pipeline.json 2> $out/bench_no_pipeline.err
FOA_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --no-extra-legs > $out/${tag}_bench_gpus2_one_device.json 2> $out/bench_gpus2.err
python3 tools/bench_stream.py 60000 > $out/${tag}_process_samples_api.jsonl 2> $out/bench_stream.err
ls -la $out

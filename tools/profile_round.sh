#!/bin/bash
# usage (on the GPU box, from the repo root):  tools/profile_round.sh <tag> [bench args]
# Takes the three rocprofv3 passes the judged numbers come from and writes their summaries under gpurun_out/prof_<tag>/:
#   kernel trace + stats of a default bench run, one --pmc pass for FETCH_SIZE, one for WRITE_SIZE (never combined
#   with traces), one for SQ_* issue counters, one for the LDS counters and one for the L2 hit/miss counters.
#   Copy <tag>_* from there into profiles/.
tag=$1; shift
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
# (--no-sync-leg: the extra leg with the device pre-sync launches the same kernels again under different sharing; left in,
#  they would enter the per-kernel averages that are compared with bench.py's live HIP-event times.  The second trace is
#  the DRIVER'S command -- `python3 bench.py --gpus 1 --steps 20 --warmup 5`, every leg -- so that its per-kernel averages can be compared like for
#  like with a BENCH_rNN record.)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sync-leg --no-extra-legs --no-self-check "$@" > $out/${tag}_bench_under_rocprof.json 2> $out/kt.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kf -o kf -- python3 bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $out/${tag}_driver_cmd_under_rocprof.json 2> $out/kf.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf -o pf -- python3 bench.py --steps 3 --warmup 1 --reps 1 --no-self-check --no-cpu-baseline --no-extra-legs "$@" > /dev/null 2> $out/pf.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw -o pw -- python3 bench.py --steps 3 --warmup 1 --reps 1 --no-self-check --no-cpu-baseline --no-extra-legs "$@" > /dev/null 2> $out/pw.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $out/ps -o ps -- python3 bench.py --steps 3 --warmup 1 --reps 1 --no-self-check --no-cpu-baseline --no-extra-legs "$@" > /dev/null 2> $out/ps.log
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $out/pl -o pl -- python3 bench.py --steps 3 --warmup 1 --reps 1 --no-self-check --no-cpu-baseline --no-extra-legs "$@" > /dev/null 2> $out/pl.log
# effective shader clock of the forward pass: GRBM_GUI_ACTIVE (cycles the GPU was busy) over the launch's own duration
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/pg -o pg -- python3 bench.py --steps 3 --warmup 1 --reps 1 --no-self-check --no-cpu-baseline --no-extra-legs --no-sync-leg "$@" > /dev/null 2> $out/pg.log
cp $out/kt/kt_kernel_stats.csv $out/${tag}_bench_kernel_stats.csv
python3 tools/trace_excerpt.py $out/kt/kt_kernel_trace.csv $out/${tag}_kernel_trace_excerpt.csv > $out/${tag}_kernel_trace_excerpt.txt
cp $out/kf/kf_kernel_stats.csv $out/${tag}_bench_with_sync_leg_kernel_stats.csv
python3 tools/reduce_pmc.py $out $tag "$@"
# L2 hit rate last and under its own timeout: a TA/TCP/TCC counter set hung a box once; everything above is already reduced
timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pc -o pc -- python3 bench.py --steps 3 --warmup 1 --reps 1 --no-self-check --no-cpu-baseline --no-extra-legs "$@" > /dev/null 2> $out/pc.log && python3 tools/reduce_pmc.py $out $tag "$@" > /dev/null
ls -la $out

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for h in 1 2; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_h$h -o kt -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sync-leg --no-extra-legs --fe-hold $h > gpurun_out/tr_h$h.json 2> gpurun_out/tr_h$h.log
  f=$(find gpurun_out/tr_h$h -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_excerpt.py $f gpurun_out/tr_h$h.csv 8 3
  rm -rf gpurun_out/tr_h$h
done

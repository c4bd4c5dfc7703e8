mkdir -p gpurun_out/r06b
ROWS=1 tools/latency_quick.sh 6000 2>&1 | grep batch
S=3520
: > gpurun_out/r06b/lat_ab.txt
for rep in 1 2 3; do
for mode in "6 0" "12 0" "12 2000"; do
set -- $mode
for BC in "4096 4096" "8192 4096" "8192 8192"; do
  set -- $mode $BC
  echo "== bufs $1 spin_us $2 batch $3 chunk $4" >> gpurun_out/r06b/lat_ab.txt
  FOA_STREAM_BUFS=$1 FOA_STREAM_SPIN_US=$2 FOA_STREAM_STATS=1 /tmp/foa_sim_lat /tmp/stream_lat.fc32 --format fc32 --preload --latency $((S+160)) 80 $S --chunk $4 --device-batch $3 --narrow-threads 2 --pace 20 2>&1 | grep -i "payload latency\|caller ms" | sed 's/(5999 payloads.*//; s/; 0 tasks.*//' >> gpurun_out/r06b/lat_ab.txt
done
done
done
cat gpurun_out/r06b/lat_ab.txt

#!/bin/bash
# usage (GPU box): tools/latency_stages.sh [frames] [runs]  -- where a small batch's time goes, and what each of the stream engine's small-batch
# settings is worth: process_samples in device mode at 20 Msample/s in calls of 4096 samples, batches of 4 Ki and 8 Ki samples, with
#   default                     copies on the pre-sync's stream, twelve buffers, polling submitter
#   FOA_STREAM_COPY_STREAM=1    upload + carry copy on a copy stream of their own (rounds 3-5)
#   FOA_STREAM_SPIN_US=0        the submitter sleeps between polls
#   FOA_STREAM_BUFS=6           six buffers in rotation
# Prints the engine's own stage medians (FOA_STREAM_STATS) and the payload latency percentiles of every run (profiles/r06_latency_stages.txt).
n=${1:-20000}; runs=${2:-3}
ROWS=1 tools/latency_quick.sh $n > /dev/null 2>&1          # builds /tmp/stream_lat.fc32 and /tmp/foa_sim_lat
S=3520                                                      # samples of a 1024-byte frame at 54 Mbps
for r in $(seq 1 $runs); do
for mode in "default" "FOA_STREAM_COPY_STREAM=1" "FOA_STREAM_SPIN_US=0" "FOA_STREAM_BUFS=6"; do
for B in 4096 8192; do
  echo "== $mode, batch $B"
  ( [ "$mode" != default ] && export $mode; FOA_STREAM_STATS=1 /tmp/foa_sim_lat /tmp/stream_lat.fc32 --format fc32 --preload --latency $((S+160)) 80 $S --chunk 4096 --device-batch $B --narrow-threads 2 --pace 20 2>&1 \
      | grep -i "payload latency\|way through" | sed 's/ (.*payloads; from.*//; s/foa_stream: a batch.s way through the submitter, //' )
done
done
done

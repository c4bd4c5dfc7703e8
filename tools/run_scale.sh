#!/bin/bash
# usage (on a multi-GPU node, from the repo root):  tools/run_scale.sh [steps] [warmup]
# BASELINE config 4's curve: bench.py at N = 1, 2, 4, 8 ranks (as many as the node has devices), each started the way the driver starts it --
# python -m torch.distributed.run, one rank per GPU over RCCL -- back to back; one JSON line per N into gpurun_out/scale_N.json and the
# efficiency table (value_N / (N x value_1): weak scaling, 10 000 frames per GPU) on stdout.
steps=${1:-20}; warmup=${2:-5}
ndev=$(python3 -c "import torch; print(torch.cuda.device_count())")
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 1 2 4 8; do
  [ "$n" -gt "$ndev" ] && break
  port=$((29500 + n))
  if [ "$n" = 1 ]; then python3 bench.py --gpus 1 --steps $steps --warmup $warmup --no-extra-legs > gpurun_out/scale_$n.json 2> gpurun_out/scale_$n.err
  else python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --steps $steps --warmup $warmup \
         --no-extra-legs > gpurun_out/scale_$n.json 2> gpurun_out/scale_$n.err; fi
done
python3 - <<'PY'
import glob, json
rows = []
for f in sorted(glob.glob("gpurun_out/scale_*.json"), key=lambda p: int(p.split("_")[-1].split(".")[0])):
    try:
        d = json.loads([ln for ln in open(f) if ln.startswith("{")][0])
        rows.append((d["n_gpus"], d["value"], d["ms_per_step"], d["config"]["psdu_bit_exact"], d["config"].get("collective", {}).get("backend")))
    except Exception as e:
        print(f, "unreadable:", e)
if rows:
    v1 = rows[0][1] / rows[0][0]
    print("%4s %14s %10s %10s %9s %s" % ("GPUs", "Msamples/s", "ms/step", "efficiency", "bit-exact", "collective"))
    for n, v, ms, ok, be in rows:
        print("%4d %14.1f %10.4f %10.3f %9s %s" % (n, v, ms, v / (n * v1), ok, be or "-"))
PY

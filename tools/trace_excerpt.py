#!/usr/bin/env python3
"""A <= 60-line excerpt of a rocprofv3 kernel trace of bench.py: the launches of a few consecutive timed steps with their
queue, start and end (us, relative), so that the tracked evidence shows what DESIGN.md 4 describes -- forward passes on
alternating queues overlapping at their ends (a launch lasts longer than a step), everything else riding along.
usage: tools/trace_excerpt.py <kt_kernel_trace.csv> <out.csv> [first forward pass to show, default 12] [how many, default 6]"""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 12
count = int(sys.argv[4]) if len(sys.argv) > 4 else 6
rows = []
with open(src, newline="") as fh:
    for r in csv.DictReader(fh):
        low = {k.lower(): v for k, v in r.items()}
        name = low["kernel_name"].split("(")[0].split("<")[0].split("::")[-1].strip()
        if name.startswith("k_"):
            rows.append((int(low["start_timestamp"]), int(low["end_timestamp"]), name, low.get("queue_id", "?")))
rows.sort()
fwd = [i for i, r in enumerate(rows) if "viterbi_fwd" in r[2]]
lo, hi = fwd[first], fwd[min(first + count, len(fwd) - 1)]
t0 = rows[lo][0]
sel = [r for r in rows[max(0, lo - 4):hi + 1]]
keep = [r for r in sel if "scan_sums" not in r[2] and "scan_apply" not in r[2]]        # the three scan launches show as k_scan_blocks
with open(dst, "w") as out:
    out.write("# rocprofv3 --kernel-trace of `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sync-leg --no-extra-legs --no-self-check`; times in us relative to the first row's forward pass\n")
    out.write("kernel,queue,start_us,end_us,duration_us\n")
    for s, e, n, q in keep[:58]:
        out.write("%s,%s,%.1f,%.1f,%.1f\n" % (n, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
f = [r for r in rows[lo:hi + 1] if "viterbi_fwd" in r[2]]
gaps = [(b[0] - a[0]) / 1e3 for a, b in zip(f, f[1:])]
over = [(a[1] - b[0]) / 1e3 for a, b in zip(f, f[1:])]
print("forward passes shown: %d; start-to-start %.0f..%.0f us; durations %.0f..%.0f us; overlap with the next %.0f..%.0f us; queues %s"
      % (len(f), min(gaps), max(gaps), min((r[1] - r[0]) / 1e3 for r in f), max((r[1] - r[0]) / 1e3 for r in f), min(over), max(over), sorted(set(r[3] for r in f))))

#!/usr/bin/env python3
"""A rocprofv3 kernel trace (CSV) as a timeline: for a window of the run, every launch with its queue, start (us from the window's start) and
duration -- where the short kernels land against the long ones.   python tools/trace_timeline.py <kernel_trace.csv> [skip fraction] [window us]"""
import csv
import sys


def main():
    path = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
    win = float(sys.argv[3]) if len(sys.argv) > 3 else 3000.0
    rows = []
    for r in csv.DictReader(open(path, newline="")):
        low = {k.lower(): v for k, v in r.items()}
        rows.append((int(low["start_timestamp"]), int(low["end_timestamp"]), low["kernel_name"].split("(")[0].replace("foa::", "").replace("void ", "")[:34], low.get("queue_id", "?")))
    rows.sort()
    t0 = rows[0][0] + (rows[-1][1] - rows[0][0]) * skip
    queues = sorted({q for _, _, _, q in rows})
    print("queues:", queues)
    for s, e, name, q in rows:
        if s < t0 or s > t0 + win * 1e3:
            continue
        print("%9.1f us  +%8.1f us  q%-3s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, queues.index(q), "    " * queues.index(q), name))


if __name__ == "__main__":
    main()

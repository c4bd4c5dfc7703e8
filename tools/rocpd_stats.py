#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite file (kernel-trace) as a per-kernel table (text, for profiles/)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), "
                  "max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("%-60s %6s %12s %12s %12s %12s %6s %5s %5s %7s %9s %5s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds", "grid", "wg"))
for r in rows:
    name = r[0].split("(")[0][-60:]
    print("%-60s %6d %12.1f %12.2f %12.2f %12.2f %6.2f %5d %5d %7d %9d %5d" % (name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10]))

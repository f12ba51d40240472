#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, min, max in microseconds) of a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- ...` writes DIR/NAME_results.db on this ROCm): the text the
summaries under profiles/ are made of.   python tools/rocpd_stats.py DB [top_n] [name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                  "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot:.1f} us over {sum(r[1] for r in rows)} dispatches")
print("name,calls,total_us,percent,avg_us,min_us,max_us")
for r in [r for r in rows if flt in r[0]][:top]:
    print('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f' % (r[0][:110], r[1], r[2], 100 * r[2] / tot, r[3], r[4], r[5]))

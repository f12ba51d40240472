#!/usr/bin/env python3
"""Kernel statistics of the LAST `seconds` of a rocpd database (the timed region of a bench run sits at the end, behind
set-up and pre-roll): per-kernel calls / total / average, plus the busy fraction = sum of kernel durations / window.
    python tools/rocpd_window.py DB seconds [top_n]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) * 1e9
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t1 = db.execute("select max(end) from kernels").fetchone()[0]
t0 = t1 - win
rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3 from kernels where start >= ? "
                  "group by name order by 3 desc", (t0,)).fetchall()
tot = sum(r[2] for r in rows)
print(f"window {win / 1e9:.3f} s: {sum(r[1] for r in rows)} dispatches, kernel time {tot / 1e3:.1f} ms = {100 * tot * 1e3 / win:.1f} % of the window (streams overlap: can exceed 100)")
print("name,calls,total_us,percent,avg_us,min_us")
for r in rows[:top]:
    print('"%s",%d,%.1f,%.2f,%.2f,%.2f' % (r[0][:120], r[1], r[2], 100 * r[2] / tot, r[3], r[4]))

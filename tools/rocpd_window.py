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
# union of the kernel intervals (time with at least one kernel resident), the idle gaps between them, and the
# concurrency profile (share of the window with exactly k kernels in flight)
ev = []
for st, en in db.execute("select start, end from kernels where start >= ? order by start", (t0,)):
    ev.append((st, 1)); ev.append((en, -1))
ev.sort()
depth, last, hist, gaps = 0, t0, {}, []
for t, dlt in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    if depth == 0 and t > last:
        gaps.append(t - last)
    depth += dlt
    last = t
busy = sum(v for k, v in hist.items() if k > 0)
gaps.sort(reverse=True)
print(f"at least one kernel resident: {100 * busy / win:.1f} % of the window; idle {100 * hist.get(0, 0) / win:.1f} % in {len(gaps)} gaps "
      f"(longest {gaps[0] / 1e3 if gaps else 0:.0f} us, mean {sum(gaps) / max(1, len(gaps)) / 1e3:.1f} us); "
      "kernels in flight -> share: " + ", ".join(f"{k}: {100 * v / win:.1f} %" for k, v in sorted(hist.items())[:8]))
print("name,calls,total_us,percent,avg_us,min_us")
for r in rows[:top]:
    print('"%s",%d,%.1f,%.2f,%.2f,%.2f' % (r[0][:120], r[1], r[2], 100 * r[2] / tot, r[3], r[4]))

# Optional 4th argument: a substring of a kernel name -> the same window split by launch grid (one line per grid size), so
# that ONE shape of an instantiation shared by several shapes can be read off (e.g. the dominant GEMM: its grid is
# ceil(M / BM) * ceil(N / BN) tiles + the prefetch blocks).
if len(sys.argv) > 4:
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    gcol = next((c for c in ("grid_x", "grid_size_x", "grid_size", "grid") if c in cols), None)
    wcol = next((c for c in ("workgroup_x", "workgroup_size_x", "workgroup_size") if c in cols), None)
    print(f"# launches of kernels matching '{sys.argv[4]}' by grid ({gcol} / {wcol}); columns available: {cols}")
    if gcol:
        sel = f"{gcol}" + (f", {wcol}" if wcol else "")
        for r in db.execute(f"select name, {sel}, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where start >= ? and name like ? "
                            f"group by name, {sel} order by 4 desc", (t0, f"%{sys.argv[4]}%")).fetchall():
            print(",".join(str(x)[:100] if i == 0 else (f"{x:.2f}" if isinstance(x, float) else str(x)) for i, x in enumerate(r)))

#!/usr/bin/env python3
"""Reads the rocpd databases tools/pmc_probe.sh left under /tmp/pmc_* and prints the per-dispatch means of every counter
for the gemm_bf16_kernel dispatches, plus the derived figures bench.py's `roofline.traffic` uses (HBM-side bytes per
launch = 2 x FETCH_SIZE (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md §HBM) + WRITE_SIZE, both in KiB
units of the counter) and the MFMA utilisation."""
import glob, json, sqlite3, sys
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
M, N, K = 768 * B, 4096, 1024
out = {"shape": [M, N, K], "command": "rocprofv3 --pmc <group> --kernel-trace -- python3 tools/dominant_kernel.py %d (one group per run)" % B,
       "per_dispatch_mean": {}, "kernel": None}
for f in glob.glob("/tmp/pmc_*/*.db"):
    db = sqlite3.connect(f)
    rows = db.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events where name like '%gemm_bf16_kernel%' "
                      "group by name, counter_name").fetchall()
    for name, cn, n, avg in rows:
        out["kernel"] = name[:90]
        out["per_dispatch_mean"][cn] = avg
        out["dispatches"] = n
    d = db.execute("select avg(end-start)/1e3 from kernels where name like '%gemm_bf16_kernel%'").fetchone()[0]
    out.setdefault("kernel_us_under_pmc", {})[f.split("/")[2]] = d
c = out["per_dispatch_mean"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    out["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    out["algorithmic_bytes_per_launch"] = 2.0 * (M * K + N * K + M * N)
if "TCC_HIT_sum" in c:
    out["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
    # rocprofv3 reports SQ counters once per shader engine (32 rows per dispatch, each a per-SE partial sum) while
    # GRBM_GUI_ACTIVE / FETCH_SIZE / WRITE_SIZE / TCC_* rows repeat the device total: the mean over rows of an SQ counter is
    # 1/32 of the device total
    out["mfma_utilisation_all_1024_simds"] = 32.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 1024.0)
    w = c.get("SQ_WAVE_CYCLES", 0.0)
    if w:
        out["wave_time_split"] = {"parked_waitcnt_or_barrier": c["SQ_WAIT_ANY"] / w,
                                  "issuing": c["SQ_ACTIVE_INST_ANY"] / w,
                                  "issue_stall": (c["SQ_WAIT_INST_ANY"] - 0.0) / w}
    out["note"] = ("per_dispatch_mean is the mean over rocprofv3's rows: SQ_* device totals are 32 x the value listed "
                   "(per-shader-engine partial sums), the other counters are device totals")
print(json.dumps(out, indent=1))

#!/bin/bash
# PMC passes (one counter group per run, --kernel-trace only, as the guide prescribes) over any python tool, summarised
# per kernel-name substring:   tools/pmc_kernel.sh <tag> <kernel substring> <python script> [args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; KSUB=$2; shift 2
mkdir -p gpurun_out/pmc
rm -rf /tmp/pmck_*
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU" \
           "TCP_REQ_sum TCP_REQ_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmck_$i -o p -- python3 "$@" > gpurun_out/pmc/run_${TAG}_$i.log 2>&1
done
python3 - "$KSUB" > gpurun_out/pmc/$TAG.json <<'PY'
import glob, json, sqlite3, sys
ks = sys.argv[1]
out = {"kernel_substring": ks, "per_dispatch_mean_of_row_sums": {}, "note": "SQ_* rows are per shader engine (summed here over the rows of a dispatch); TCC_/FETCH/WRITE rows repeat the device total (max taken)"}
for f in sorted(glob.glob("/tmp/pmck_*/*.db")):
    db = sqlite3.connect(f)
    try:
        rows = db.execute("select dispatch_id, counter_name, sum(counter_value), max(counter_value), count(*) from pmc_events "
                          "where name like ? group by dispatch_id, counter_name", (f"%{ks}%",)).fetchall()
    except Exception as e:
        out.setdefault("errors", []).append(f"{f}: {e}")
        continue
    acc = {}
    for did, cn, s, mx, n in rows:
        v = mx if cn.startswith(("TCC_", "FETCH", "WRITE", "GRBM")) else s
        acc.setdefault(cn, []).append(v)
    for cn, v in acc.items():
        out["per_dispatch_mean_of_row_sums"][cn] = sum(v) / len(v)
        out["dispatches"] = len(v)
    d = db.execute("select avg(end-start)/1e3, count(*) from kernels where name like ?", (f"%{ks}%",)).fetchone()
    out.setdefault("kernel_us_under_pmc", []).append(d[0])
print(json.dumps(out, indent=1))
PY
cat gpurun_out/pmc/$TAG.json

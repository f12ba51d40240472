#!/bin/bash
# HBM-side traffic of the network stages (FETCH_SIZE and WRITE_SIZE, one counter per run as the guide prescribes):
#   tools/pmc_stage_traffic.sh ["enc4 dec4 dec1"]   -> gpurun_out/pmc/stage_traffic.json   (stages: enc<B> / dec<B>)
# FETCH_SIZE / WRITE_SIZE rows repeat the device total per dispatch (max over a dispatch's rows is taken); gfx950 tallies
# 128-byte fetch requests as 64 B, hence fetch x 2 (MI355X_MICROARCH.md, HBM section).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
N=6
STAGES=${1:-"enc4 dec4 dec1"}
rm -rf /tmp/pst_*
for st in $STAGES; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pst_${st}_$c -o p -- python3 tools/stage_profile.py $st $N > gpurun_out/pmc/run_stage_${st}_$c.log 2>&1
  done
done
python3 - $N $STAGES > gpurun_out/pmc/stage_traffic.json <<'PY'
import glob, json, sqlite3, sys
N = float(sys.argv[1])
out = {"method": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python3 tools/stage_profile.py <stage> %d; per-dispatch device totals summed over every kernel of the process except the weight upload (copyBuffer / fill kernels before the first stage call are included in 'all', excluded in 'mslam')" % int(N)}
stages = sys.argv[2:]
for st, frames in ((st, int(st[3:]) if len(st) > 3 else 1) for st in stages):
    rec = {"frames_per_call": frames}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"/tmp/pst_{st}_{c}/*.db") + glob.glob(f"/tmp/pst_{st}_{c}/*/*.db")
        if not f:
            rec[c] = None
            continue
        db = sqlite3.connect(f[0])
        rows = db.execute("select name, dispatch_id, max(counter_value) from pmc_events where counter_name = ? group by dispatch_id", (c,)).fetchall()
        rec[c + "_kib_per_call_mslam"] = sum(v for n, d, v in rows if "mslam" in n) / N
        rec[c + "_kib_per_call_all"] = sum(v for n, d, v in rows) / N
        rec["dispatches_per_call_mslam"] = sum(1 for n, d, v in rows if "mslam" in n) / N
    if rec.get("FETCH_SIZE_kib_per_call_mslam") is not None and rec.get("WRITE_SIZE_kib_per_call_mslam") is not None:
        mb = (2.0 * rec["FETCH_SIZE_kib_per_call_mslam"] + rec["WRITE_SIZE_kib_per_call_mslam"]) * 1024.0 / 1e6
        rec["hbm_mb_corrected_per_call"] = mb
        rec["hbm_mb_corrected_per_frame"] = mb / frames
    out[st] = rec
encs = [s for s in stages if s.startswith("enc") and "hbm_mb_corrected_per_frame" in out.get(s, {})]
decs = [s for s in stages if s.startswith("dec") and "hbm_mb_corrected_per_frame" in out.get(s, {})]
if encs and decs:   # a tracked frame = its share of the first encoder stage + of the first decoder stage listed
    out["tracked_frame_mb_%s_%s" % (encs[0], decs[0])] = out[encs[0]]["hbm_mb_corrected_per_frame"] + out[decs[0]]["hbm_mb_corrected_per_frame"]
print(json.dumps(out, indent=1))
PY
cat gpurun_out/pmc/stage_traffic.json

#!/usr/bin/env python3
"""Per-kernel MFMA utilisation and wave-time split from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE):
pmc_util.py <counter_collection.csv> -> CSV on stdout.  MFMA utilisation = busy cycles / (kernel cycles x 1024 SIMDs),
kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("mslam::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
tot_cyc = sum(v["GRBM_GUI_ACTIVE"] for v in acc.values())
print("kernel,dispatches,share_of_gpu_cycles,mfma_util,parked_waitcnt_barrier,issue_stall,issuing")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    wc = max(v["SQ_WAVE_CYCLES"], 1.0)
    print(f'"{k}",{len(disp[k])},{v["GRBM_GUI_ACTIVE"] / tot_cyc:.4f},{v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024):.4f},'
          f'{v["SQ_WAIT_ANY"] / wc:.3f},{v["SQ_WAIT_INST_ANY"] / wc:.3f},{v["SQ_ACTIVE_INST_ANY"] / wc:.3f}')

#!/usr/bin/env python3
"""Sums one PMC counter of a rocprofv3 --pmc run per kernel: pmc_sum.py <counter_collection.csv> <calls> -> CSV on stdout
(Kernel_Name, Dispatches, Sum_<counter>_raw per call)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
calls = float(sys.argv[2])
tot, cnt = collections.Counter(), collections.Counter()
name = rows[0]["Counter_Name"]
for r in rows:
    tot[r["Kernel_Name"]] += float(r["Counter_Value"])
    cnt[r["Kernel_Name"]] += 1
print(f"Kernel_Name,Dispatches_per_call,Sum_{name}_raw_per_call")
for k, v in tot.most_common():
    print(f'"{k}",{cnt[k] / calls:.1f},{v / calls:.1f}')
print(f'"TOTAL",{sum(cnt.values()) / calls:.1f},{sum(tot.values()) / calls:.1f}')

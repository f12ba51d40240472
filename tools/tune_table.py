#!/usr/bin/env python3
"""Tabulates a tools/gemm_tune.py log: one row per shape, one column per configuration."""
import collections, sys
d, cfgs = collections.OrderedDict(), []
for ln in open(sys.argv[1]):
    p = ln.split()
    if len(p) != 5:
        continue
    c, M, N, K, us = p
    if c not in cfgs:
        cfgs.append(c)
    d.setdefault((int(M), int(N), int(K)), {})[c] = float(us)
print("shape".ljust(20) + "".join(c.rjust(8) for c in cfgs) + "   best")
for k, v in d.items():
    best = min((x for x in v if x != "auto"), key=lambda c: v[c])
    print(str(k).ljust(20) + "".join(f"{v.get(c, float('nan')):8.1f}" for c in cfgs) +
          f"   {best} {2 * k[0] * k[1] * k[2] / v[best] * 1e-6:.0f}TF")

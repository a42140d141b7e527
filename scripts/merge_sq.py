#!/usr/bin/env python
"""Fold the SQ counter summary of scripts/pmc_sq.sh into a profile's traffic.json (per read, summed over the kernels of one step):
    python scripts/merge_sq.py <traffic.json> <sq.txt> <waves_per_simd>
bench.py reads `sq_per_read`, `l1_to_l2_read_requests_per_read` and `tcc_miss_per_read` from the committed copy under profiles/."""
import json
import sys

traffic_path, sq_path, wps = sys.argv[1], sys.argv[2], int(sys.argv[3])
t = json.load(open(traffic_path))
n = t["reads_per_gpu"]
sq = {"note": "scripts/pmc_sq.sh on the same workload, per read, summed over the kernels of one step; *_CYCLES / WAIT / ACTIVE in quad-cycles summed over waves"}
for line in open(sq_path):
    if line.startswith(" ") or ":" not in line:
        continue
    name, val = line.split(":")
    try:
        sq[name.strip()] = round(float(val) / n, 2)
    except ValueError:
        pass
sq["waves_per_simd"] = wps
t["sq_per_read"] = sq
if "TCP_TCC_READ_REQ_sum" in t:
    t["l1_to_l2_read_requests_per_read"] = t["TCP_TCC_READ_REQ_sum"] / n
if "TCC_MISS_sum" in t:
    t["tcc_miss_per_read"] = t["TCC_MISS_sum"] / n
json.dump(t, open(traffic_path, "w"), indent=1)
print(json.dumps({k: v for k, v in t.items() if k != "sq_per_read"}))

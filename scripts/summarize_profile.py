#!/usr/bin/env python
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small text summary for profiles/."""
import csv
import glob
import json
import os
import sys


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def main(out):
    print(f"# rocprofv3 summary for {os.path.basename(out)}")
    for name in ("bench.json", "trace.json"):
        p = os.path.join(out, name)
        if os.path.exists(p) and os.path.getsize(p):
            line = open(p).read().strip().splitlines()[-1]
            try:
                j = json.loads(line)
                print(f"{name}: value={j['value']:.2f} {j['unit']} ms_per_step={j['ms_per_step']:.3f} kernel_ms_avg={j['roofline']['kernel_ms_avg']:.3f} "
                      f"achieved={j['roofline']['achieved']:.2f} GB/s frac={j['roofline']['frac']:.5f}")
            except Exception as e:  # noqa: BLE001
                print(f"{name}: unparsable ({e})")
    for f in find(os.path.join(out, "trace"), "*kernel_stats.csv"):
        print(f"\n## kernel stats ({os.path.relpath(f, out)})")
        rows = list(csv.DictReader(open(f)))
        cols = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"]
        print(",".join(cols))
        for r in rows[:12]:
            print(",".join(str(r.get(c, ""))[:110] for c in cols))
    for tag in ("pmc_fetch", "pmc_write", "pmc_l2"):
        for f in find(os.path.join(out, tag), "*counter_collection.csv"):
            print(f"\n## {tag} ({os.path.relpath(f, out)}) — per-dispatch counter values of the alignment kernel")
            agg = {}
            for r in csv.DictReader(open(f)):
                if "nb_align" not in r.get("Kernel_Name", ""):
                    continue
                key = (r["Kernel_Name"][:60], r["Counter_Name"])
                agg.setdefault(key, []).append(float(r["Counter_Value"]))
            for (k, c), v in agg.items():
                print(f"{k} {c}: dispatches={len(v)} mean={sum(v) / len(v):.1f} min={min(v):.1f} max={max(v):.1f}")


if __name__ == "__main__":
    main(sys.argv[1])

#!/usr/bin/env python
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small text summary for profiles/."""
import csv
import glob
import json
import os
import sys


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def main(out, cfg="c2"):
    print(f"# rocprofv3 summary for {os.path.basename(out)} (bench.py --config {cfg})")
    traffic = {"config": cfg}
    for name in ("bench.json", "trace.json"):
        p = os.path.join(out, name)
        if os.path.exists(p) and os.path.getsize(p):
            line = open(p).read().strip().splitlines()[-1]
            try:
                j = json.loads(line)
                if name == "bench.json":
                    traffic.update(reads_per_gpu=j["config"]["reads_per_gpu"], read_len=j["config"]["read_len"], n_seqs=j["config"]["n_seqs"],
                                   k=j["config"]["k"], kernel=j["roofline"]["kernel"], kernel_ms_avg=j["roofline"]["kernel_ms_avg"])
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
        # one step of the bench = one nb_align_batch call = several nb:: launches (the anchor kernel twice: all reads, then the stragglers'
        # retry list): bench.py's kernel_ms_avg is their sum per step, to be compared with the total printed here
        nb = [r for r in rows if "nb::" in r.get("Name", "")]
        if nb:
            calls = min(int(r["Calls"]) for r in nb)
            print(f"# nb:: kernels, per step (steps incl. warm-up = {calls}, the launch count of the kernels that run once per step):")
            tot = 0.0
            for r in nb:
                per = float(r["TotalDurationNs"]) / calls / 1e6
                tot += per
                print(f"#   {r['Name'][:80]}: {int(r['Calls']) / calls:g} launches, {per:.3f} ms")
            print(f"#   all nb:: kernels: {tot:.3f} ms per step")
    for tag in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_l1"):
        for f in find(os.path.join(out, tag), "*counter_collection.csv"):
            print(f"\n## {tag} ({os.path.relpath(f, out)}) — per-dispatch counter values of the alignment kernel")
            # one step of the bench = every nb:: kernel of one nb_align_batch call (lane-per-read path: forward pass, reverse
            # worklist, gather, wave-per-read kernel over the redo list, scatter): per-step value = sum over those kernels / steps,
            # steps = dispatches of a kernel that runs once per step
            agg, total = {}, {}
            for r in csv.DictReader(open(f)):
                if "nb::" not in r.get("Kernel_Name", ""):
                    continue
                key = (r["Kernel_Name"][:60], r["Counter_Name"])
                agg.setdefault(key, []).append(float(r["Counter_Value"]))
            steps = min((len(v) for v in agg.values()), default=1)      # (the anchor kernel is launched twice per step, gather / scatter once)
            for (k, c), v in agg.items():
                print(f"{k} {c}: dispatches={len(v)} mean={sum(v) / len(v):.1f} min={min(v):.1f} max={max(v):.1f}")
                total[c] = total.get(c, 0.0) + sum(v)
            for c, v in total.items():
                traffic[c] = v / steps
                print(f"ALL nb:: kernels {c}: per step = {v / steps:.1f}")
    if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic:
        # rocprofv3 reports both in KB per dispatch. FETCH_SIZE = TCC_EA0_RDREQ x 64 B (guide, MI355X_MICROARCH §HBM): for this
        # kernel's scattered 16-B probes / 8-B gathers each request is one 64-B fabric read, so no x2 streaming correction is
        # applied; the x2 figure is kept as an upper bound. Infinity-Cache hits are counted, so this is L2<->fabric traffic.
        traffic["traffic_bytes_per_launch"] = (traffic["FETCH_SIZE"] + traffic["WRITE_SIZE"]) * 1024
        traffic["traffic_bytes_per_launch_upper_x2_fetch"] = (2 * traffic["FETCH_SIZE"] + traffic["WRITE_SIZE"]) * 1024
        print(f"\ntraffic per launch: {traffic['traffic_bytes_per_launch'] / 1e9:.2f} GB (fetch as reported) .. "
              f"{traffic['traffic_bytes_per_launch_upper_x2_fetch'] / 1e9:.2f} GB (x2 fetch upper bound)")
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:3])

#!/bin/bash
# individual kernel durations of the timed steps (run on the GPU box): bash scripts/r03_calls.sh <tag> [bench args...]
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 0 "$@" > "$OUT/trace.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
f=$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY' | tee "$OUT/calls.txt"
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "nb::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows[-24:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - t0) / 1e3 if t0 else 0.0
    print(f'{r["Kernel_Name"].replace("void nb::","")[:58]:58s} {(e - s) / 1e3:9.1f} us   gap {gap:7.1f} us   grid {r.get("Grid_Size","?")} wg {r.get("Workgroup_Size","?")}')
    t0 = e
PY
rm -rf "$OUT/trace"

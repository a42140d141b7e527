#!/bin/bash
# per-kernel times of one bench configuration: bash scripts/r03_trace.sh <tag> [bench args...]
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --parity-sample 0 "$@" > "$OUT/trace.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
f=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY' | tee "$OUT/kernel_stats.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if r["Name"].startswith(("nb", "void nb")) or "nb::" in r["Name"]:
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:10.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} {r["Percentage"]:>6s}%')
PY
rm -rf "$OUT/trace"

#!/bin/bash
# kernel-trace stats + SQ counters of every nb_* kernel of one bench run (run on the GPU box): bash scripts/lane_prof.sh <tag> [bench args...]
set -e
TAG=${1:-lane}; shift || true
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
Q="--steps 3 --warmup 1 --no-cpu-baseline --parity-sample 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o b -- python3 bench.py $Q > "$OUT/trace.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 bench.py $Q > "$OUT/$name.json" 2> "$OUT/$name.err" || { tail -5 "$OUT/$name.err"; exit 1; }
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
run l2 TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("## kernel stats")
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith(("void nb::", "nb::")):
            print(f"{r['Name'][:70]:70s} calls={r['Calls']:>4s} avg_ms={float(r['AverageNs'])/1e6:9.3f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f}")
for tag in ("sq1", "sq2", "l2"):
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        agg = {}
        for r in csv.DictReader(open(f)):
            if "nb::" in r["Kernel_Name"]:
                agg.setdefault((r["Kernel_Name"][:48], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print(f"{k:48s} {c:22s} mean={sum(v)/len(v):.4g} n={len(v)}")
PY
# keep the merged-back output small (gpurun copies gpurun_out/ back only below 64 MiB): the per-dispatch traces are not needed once summarised
# raw rocprofv3 output is summarised above; only the summaries travel back (gpurun merges gpurun_out/ only below 64 MiB)
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + 2>/dev/null || true

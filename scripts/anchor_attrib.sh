#!/bin/bash
# Which array costs what (run on the GPU box): the PROF build with NB_ANCHOR_DBG redirects the sparse-table / table-entry / text loads of the
# anchor kernel to a few hot lines (results are WRONG on purpose) and reports time and L2 misses per launch for each variant.
set -e
export TMPDIR=/tmp NB_EXTRA_HIPCC_FLAGS=-DNB_ANCHOR_PROF
python -c "from nimble_amd import _build; _build.build(force=True)" > /dev/null 2>&1
for dbg in 0 1 2 4 7; do
  OUT=gpurun_out/attrib_$dbg; rm -rf $OUT; mkdir -p $OUT
  NB_ANCHOR_DBG=$dbg rocprofv3 --kernel-trace --pmc TCC_MISS_sum TCC_HIT_sum --output-format csv -d $OUT/t -o p -- python3 bench.py ${BENCH_ARGS:-} --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 0 > $OUT/out.json 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 1; }
  python3 - $OUT $dbg <<'PY'
import csv, glob, json, sys
out, dbg = sys.argv[1], sys.argv[2]
agg = {}
for f in glob.glob(out + "/t/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nb_anchor_kernel" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
j = json.loads([l for l in open(out + "/out.json") if l.startswith("{")][-1])
print("dbg=%s kernel_ms=%.3f main-kernel TCC_MISS=%.1f M TCC_HIT=%.1f M" % (dbg, j["roofline"]["kernel_ms_avg"], sum(agg["TCC_MISS_sum"]) / len(agg["TCC_MISS_sum"]) / 1e6, sum(agg["TCC_HIT_sum"]) / len(agg["TCC_HIT_sum"]) / 1e6))
PY
  rm -rf $OUT/t
done

#!/bin/bash
# L2 hit/miss counters of the alignment kernel (run on the GPU box): bash scripts/pmc_l2.sh <tag>
set -e
TAG=${1:-l2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/l2" -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --parity-sample 0 > "$OUT/l2.json" 2> "$OUT/l2.err" || { tail -5 "$OUT/l2.err"; exit 1; }
python - "$OUT" <<'PY'
import csv,glob,sys,os
out=sys.argv[1]
for f in sorted(glob.glob(os.path.join(out,'l2','**','*counter_collection.csv'),recursive=True)):
    agg={}
    for r in csv.DictReader(open(f)):
        if 'nb_align' in r['Kernel_Name']: agg.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
    for k,v in agg.items(): print(f"{k}: {sum(v)/len(v):.4g}")
PY
rm -rf "$OUT/l2"

#!/bin/bash
# instruction counts of the alignment kernel per launch (run on the GPU box): bash scripts/pmc_insts.sh <tag>
set -e
TAG=${1:-insts}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$OUT/p" -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --parity-sample 0 > "$OUT/p.json" 2> "$OUT/p.err" || { tail -5 "$OUT/p.err"; exit 1; }
python - "$OUT" <<'PY'
import csv,glob,sys,os
out=sys.argv[1]
for f in sorted(glob.glob(os.path.join(out,'p','**','*counter_collection.csv'),recursive=True)):
    agg={}
    for r in csv.DictReader(open(f)):
        if 'nb_align' in r['Kernel_Name']: agg.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
    print(" ".join(f"{k}={sum(v)/len(v)/1e7:.1f}" for k,v in agg.items()), "(per read)")
PY
rm -rf "$OUT/p"

#!/bin/bash
# instruction counts per dispatch of the alignment kernel: bash scripts/pmc_insts.sh <tag>
set -e
TAG=${1:-insts}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS --output-format csv -d "$OUT/i" -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --parity-sample 0 > "$OUT/i.json" 2> "$OUT/i.err" || { tail -5 "$OUT/i.err"; exit 1; }
python - "$OUT" <<'PY'
import csv,glob,sys,os,json
out=sys.argv[1]
j=json.loads(open(os.path.join(out,'i.json')).read().strip().splitlines()[-1]); n=j['config']['reads_per_gpu']
for f in sorted(glob.glob(os.path.join(out,'i','**','*counter_collection.csv'),recursive=True)):
    agg={}
    for r in csv.DictReader(open(f)):
        if 'nb_align' in r['Kernel_Name']: agg.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
    print(' '.join(f"{k.replace('SQ_INSTS_','')}={sum(v)/len(v)/n:.1f}" for k,v in sorted(agg.items())), 'per read; kernel_ms', round(j['roofline']['kernel_ms_avg'],2))
PY

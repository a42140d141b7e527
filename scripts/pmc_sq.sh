#!/bin/bash
# SQ counters of the alignment kernels (run on the GPU box): bash scripts/pmc_sq.sh <tag> [reads] [config]
set -e
TAG=${1:-sq}; READS=${2:-10000000}; CFG=${3:-c2}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
run() { name=$1; shift
  timeout -k 10 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 bench.py --config $CFG --steps 2 --warmup 1 --reads $READS --no-cpu-baseline --parity-sample 0 > "$OUT/$name.json" 2> "$OUT/$name.err" || { tail -5 "$OUT/$name.err"; exit 1; }
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
run sq3 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE
python - "$OUT" <<'PY'
import csv,glob,sys,os
out=sys.argv[1]
for f in sorted(glob.glob(os.path.join(out,'sq*','**','*counter_collection.csv'),recursive=True)):
    agg={}; per={}
    for r in csv.DictReader(open(f)):
        if 'nb::' in r['Kernel_Name']:
            agg.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
            per.setdefault((r['Kernel_Name'][:44], r['Counter_Name']),[]).append(float(r['Counter_Value']))
    steps=min(len(v) for v in per.values())          # (the anchor kernel is launched twice per step, gather and scatter once)
    for k,v in agg.items(): print(f"{k}: {sum(v)/steps:.4g}")          # per step: summed over the kernels of one nb_align_batch call
    for (kn,c),v in sorted(per.items()): print(f"  {kn:44s} {c}: {sum(v)/len(v):.4g}")
PY
# raw rocprofv3 output is summarised above; only the summaries travel back (gpurun merges gpurun_out/ only below 64 MiB)
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + 2>/dev/null || true

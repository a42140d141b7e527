#!/bin/bash
# SQ + L2 counters per kernel of one configuration (run on the GPU box): bash scripts/r03_pmc.sh <tag> [bench args...]
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
run() { name=$1; shift; pmc="$1"; shift
  timeout -k 10 180 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$OUT/$name" -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --parity-sample 0 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || { tail -5 "$OUT/$name.err"; return 1; }
}
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "$@" || exit 1
run sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "$@" || exit 1
run sq3 "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" "$@" || exit 1
run l2 "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "$@" || exit 1
run fetch "FETCH_SIZE" "$@" || exit 1
run write "WRITE_SIZE" "$@" || exit 1
python3 - "$OUT" <<'PY' | tee "$OUT/pmc.txt"
import csv,glob,sys,os
out=sys.argv[1]
per={}
for f in sorted(glob.glob(os.path.join(out,'*','**','*counter_collection.csv'),recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'nb::' in r['Kernel_Name']:
            per.setdefault((r['Kernel_Name'].replace('void nb::','')[:40], r['Counter_Name']),[]).append(float(r['Counter_Value']))
ks=sorted({k for k,_ in per}); cs=sorted({c for _,c in per})
for k in ks:
    print(k)
    for c in cs:
        if (k,c) in per:
            v=per[(k,c)]; print(f"   {c:28s} {sum(v)/len(v):14.5g}  (x{len(v)})")
PY
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + 2>/dev/null || true

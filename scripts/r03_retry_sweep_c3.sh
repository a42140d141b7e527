#!/bin/bash
# hand-over thresholds for the 91-bp reads of c3 and for c5: bash scripts/r03_retry_sweep_c3.sh
mkdir -p gpurun_out
one() { label=$1; shift
  env $ENVX timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 2000 "$@" 2> gpurun_out/sweep.err | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$label', 'kernel_ms', round(j['roofline']['kernel_ms_avg'],3), 'value', round(j['value'],1), 'ok', j.get('oracle_self_consistency_ok'))" || { echo "$label failed"; tail -3 gpurun_out/sweep.err; }
}
ENVX="NB_ANCHOR_RETRY_LANES=0" one "c3 no second pass" --config c3
for rounds in 4 5 6 7 8; do for lanes in 8 16 24; do
  ENVX="NB_ANCHOR_RETRY_LANES=$lanes NB_ANCHOR_RETRY_ROUNDS=$rounds" one "c3 rounds>=$rounds lanes<=$lanes" --config c3
done; done

#!/bin/bash
# thresholds of the stragglers' hand-over (state-preserving second pass), c2 and pairs: bash scripts/r03_retry_sweep.sh
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_anchor_gpu.py -x -q -m gpu > gpurun_out/retry_tests.log 2>&1; rc=$?
tail -n 3 gpurun_out/retry_tests.log
[ $rc -ne 0 ] && { echo "tests failed rc=$rc"; exit $rc; }
one() { label=$1; shift
  env $ENVX timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --parity-sample 2000 "$@" 2> gpurun_out/sweep.err | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$label', 'kernel_ms', round(j['roofline']['kernel_ms_avg'],3), 'value', round(j['value'],1), 'ok', j.get('oracle_self_consistency_ok'))" || { echo "$label failed"; tail -3 gpurun_out/sweep.err; }
}
ENVX="NB_ANCHOR_RETRY_LANES=0" one "no second pass"
for rounds in 6 7 8 9 10 12; do for lanes in 6 12 20 28 36; do
  ENVX="NB_ANCHOR_RETRY_LANES=$lanes NB_ANCHOR_RETRY_ROUNDS=$rounds" one "rounds>=$rounds lanes<=$lanes"
done; done

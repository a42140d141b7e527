#!/bin/bash
# A/B of the pipelined anchor kernels against the lock-step kernel on the GPU box: parity first, then c2 timing both ways.
#   bash scripts/r03_ab.sh [reads]
READS=${1:-10000000}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_anchor_gpu.py -x -q -m gpu > gpurun_out/ab_anchor_tests.log 2>&1; rc=$?
tail -n 5 gpurun_out/ab_anchor_tests.log
[ $rc -ne 0 ] && { echo "anchor tests failed rc=$rc"; exit $rc; }
for pipe in 1 0 1 0; do
  NB_ANCHOR_PIPE=$pipe timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --reads $READS 2> gpurun_out/ab_bench_$pipe.err | tee gpurun_out/ab_bench_$pipe.json | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('pipe', $pipe, 'value', round(j['value'],1), 'kernel_ms', round(j['roofline']['kernel_ms_avg'],3), 'ok', j.get('oracle_self_consistency_ok'), 'classes', j['classes'], j['roofline']['kernels_per_step'])" || { echo "bench pipe=$pipe failed"; tail -5 gpurun_out/ab_bench_$pipe.err; exit 1; }
done

#!/bin/bash
# quick parity + timing on the GPU box: anchor parity tests, then bench lines (c2 default, all-foreign batch): bash scripts/r03_check.sh
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_anchor_gpu.py tests/test_align_gpu.py tests/test_general_gpu.py -x -q -m gpu > gpurun_out/check_tests.log 2>&1; rc=$?
tail -n 4 gpurun_out/check_tests.log
[ $rc -ne 0 ] && { echo "tests failed rc=$rc"; exit $rc; }
one() { label=$1; shift
  env $ENVX timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2> gpurun_out/check_bench.err | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$label', 'kernel_ms', round(j['roofline']['kernel_ms_avg'],3), 'value', round(j['value'],1), 'ok', j.get('oracle_self_consistency_ok'), 'classes', j['classes'])" || { echo "$label failed"; tail -5 gpurun_out/check_bench.err; exit 1; }
}
one c2
ENVX=NB_ANCHOR_DBG=16 one c2_nofilter_in_tail
one c2_foreign --foreign 1.0
one c2_foreign50 --foreign 0.5
one c3 --config c3 --steps 3
one c2_mm2 --mismatches 2
ENVX=NB_ANCHOR_MM=0 one c2_mm2_general --mismatches 2 --reads 2000000
one c5 --config c5 --steps 3
bash scripts/r03_trace.sh check_trace
bash scripts/r03_trace.sh check_trace_foreign --foreign 1.0

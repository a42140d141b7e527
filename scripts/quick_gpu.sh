#!/bin/bash
# quick parity + timing check on the GPU box: bash scripts/quick_gpu.sh [reads]
set -e
READS=${1:-10000000}
timeout -k 10 600 python -m pytest tests/test_align_gpu.py -x -q -m gpu 2>&1 | tail -2
python bench.py --steps 5 --no-cpu-baseline --reads $READS 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('value', round(j['value'],1), 'kernel_ms', round(j['roofline']['kernel_ms_avg'],3), 'parity', j['parity_sample_ok'], 'classes', j['classes'])"

#!/bin/bash
# A/B runs of the anchor-extend kernel on the GPU box: bash scripts/anchor_sweep.sh <out.txt> "<ENV=.. ENV=..>" ["<...>" ...]
# every quoted argument is one environment for one `bench.py` run (c2, no CPU baseline); prints kernel ms and M reads/s per setting
out=$1; shift
: > "$out"
for envs in "$@"; do
  line=$(env $envs timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --parity-sample 0 ${BENCH_ARGS:-} 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('kernel_ms=%.3f value=%.1f classes=%s called=%.6f kernels=%s' % (j['roofline']['kernel_ms_avg'], j['value'], j.get('classes'), j.get('called_frac', 0), ','.join(j['roofline']['kernels_per_step'])))")
  echo "[$envs] $line" | tee -a "$out"
done

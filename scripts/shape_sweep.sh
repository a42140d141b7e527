#!/bin/bash
# bench.py (c2 library, 10 M x 150 bp) over the make-up of the batch: share of foreign reads, substitution rate.  bash scripts/shape_sweep.sh <out.txt>
out=$1; : > "$out"
for spec in "0.0 0.01" "0.1 0.01" "0.5 0.01" "0.95 0.01" "1.0 0.01" "0.1 0.0" "0.1 0.03" "0.1 0.08"; do
  set -- $spec
  line=$(timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 20000 --foreign $1 --err $2 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('kernel_ms=%.3f value=%.1f classes=%s called=%.4f parity=%s' % (j['roofline']['kernel_ms_avg'], j['value'], j.get('classes'), j.get('called_frac', 0), j.get('oracle_self_consistency_ok')))")
  echo "[foreign $1, substitutions $2] $line" | tee -a "$out"
done

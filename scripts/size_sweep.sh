#!/bin/bash
# bench.py over library sizes and read lengths (10 M reads per batch): which kernels run and how long a batch takes — looks for cliffs between
# the BASELINE configs. bash scripts/size_sweep.sh <out.txt>
out=$1; : > "$out"
for spec in "5 50 150" "10 50 150" "30 50 150" "50 50 150" "80 60 150" "100 70 150" "140 70 150" "20 50 50" "20 50 100" "20 50 250" "20 50 339" "60 60 91"; do
  set -- $spec
  line=$(timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 20000 --loci $1 --alleles $2 --read-len $3 --reads 10000000 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('kernel_ms=%.3f value=%.1f classes=%s called=%.4f parity=%s kernels=%s' % (j['roofline']['kernel_ms_avg'], j['value'], j.get('classes'), j.get('called_frac', 0), j.get('oracle_self_consistency_ok'), ','.join(k.split('(')[0] for k in j['roofline']['kernels_per_step'])))")
  echo "[$1 loci x $2 alleles, $3 bp] $line" | tee -a "$out"
done

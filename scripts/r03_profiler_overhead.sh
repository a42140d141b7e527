#!/bin/bash
# What rocprofv3 --kernel-trace adds per dispatch: the same bench line with and without the profiler, at two batch sizes (run on the GPU box)
export TMPDIR=/tmp; mkdir -p gpurun_out/ovh
for reads in 10000000 1000000; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --parity-sample 0 --reads $reads 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('reads $reads plain   : ms_per_step %.3f (wall), HIP events around nb_align_batch %.3f' % (j['ms_per_step'], j['roofline']['kernel_ms_avg']))"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ovh/t$reads -o b -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --parity-sample 0 --reads $reads 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('reads $reads profiled: ms_per_step %.3f (wall), HIP events around nb_align_batch %.3f' % (j['ms_per_step'], j['roofline']['kernel_ms_avg']))"
  f=$(find gpurun_out/ovh/t$reads -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'nb::' in r['Name']]
calls=min(int(r['Calls']) for r in rows)
print('          rocprofv3 kernel durations, per step: %.3f ms over %d dispatches' % (sum(float(r['TotalDurationNs']) for r in rows)/calls/1e6, sum(int(r['Calls']) for r in rows)//calls))
PY
done
rm -rf gpurun_out/ovh

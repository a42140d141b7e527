#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: bench line + rocprofv3 kernel stats + HBM traffic counters.
# Outputs under gpurun_out/<tag>/; copy the summaries you want judged into profiles/.
set -e
TAG=${1:-r01}
READS=${2:-10000000}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== bench (unprofiled) =="
python bench.py --steps 5 --warmup 1 --reads $READS > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
cat "$OUT/bench.json"
echo "== rocprofv3 kernel trace + stats =="
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 bench.py --steps 5 --warmup 1 --reads $READS --no-cpu-baseline --parity-sample 0 > "$OUT/trace.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
echo "== rocprofv3 PMC: FETCH_SIZE =="
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- python3 bench.py --steps 2 --warmup 1 --reads $READS --no-cpu-baseline --parity-sample 0 > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err" || { tail -5 "$OUT/pmc_fetch.err"; exit 1; }
echo "== rocprofv3 PMC: WRITE_SIZE =="
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- python3 bench.py --steps 2 --warmup 1 --reads $READS --no-cpu-baseline --parity-sample 0 > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err" || { tail -5 "$OUT/pmc_write.err"; exit 1; }
echo "== rocprofv3 PMC: L2 hit/miss =="
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -o bench -- python3 bench.py --steps 2 --warmup 1 --reads $READS --no-cpu-baseline --parity-sample 0 > "$OUT/pmc_l2.json" 2> "$OUT/pmc_l2.err" || { tail -5 "$OUT/pmc_l2.err"; exit 1; }
python scripts/summarize_profile.py "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"

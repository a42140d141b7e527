#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: bench line + rocprofv3 kernel stats + PMC passes for one config.
#   bash scripts/profile_gpu.sh <tag> <config c2|c3|c5> [steps]
# Outputs under gpurun_out/<tag>/; summary.txt and traffic.json are what gets copied into profiles/.
set -e
TAG=${1:-r01}; CFG=${2:-c2}; STEPS=${3:-5}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== bench (unprofiled) =="
python bench.py --config $CFG --steps $STEPS --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
cut -c1-400 "$OUT/bench.json"
Q="--config $CFG --warmup 1 --no-cpu-baseline --parity-sample 0"
echo "== rocprofv3 kernel trace + stats =="
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 bench.py --steps $STEPS $Q --warmup 3 > "$OUT/trace.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
for pass in "pmc_fetch FETCH_SIZE" "pmc_write WRITE_SIZE" "pmc_l2 TCC_HIT_sum TCC_MISS_sum"; do
  set -- $pass; name=$1; shift
  echo "== rocprofv3 PMC: $* =="
  timeout -k 10 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o bench -- python3 bench.py --steps 2 $Q > "$OUT/$name.json" 2> "$OUT/$name.err" || { tail -5 "$OUT/$name.err"; exit 1; }
done
# optional pass (counter names differ between ROCm releases; a missing counter must not lose the passes above): L1 -> L2 requests
echo "== rocprofv3 PMC: TCP_TCC_READ_REQ_sum (optional) =="
timeout -k 10 900 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum --output-format csv -d "$OUT/pmc_l1" -o bench -- python3 bench.py --steps 2 $Q > "$OUT/pmc_l1.json" 2> "$OUT/pmc_l1.err" || echo "(TCP_TCC_READ_REQ_sum not collected)"
python scripts/summarize_profile.py "$OUT" "$CFG" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
# keep what is merged back small (gpurun copies gpurun_out/ back only below 64 MiB)
# raw rocprofv3 output is summarised above; only the summaries travel back (gpurun merges gpurun_out/ only below 64 MiB)
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + 2>/dev/null || true

#!/bin/bash
# Random-gather ceilings of the chip at the access shapes of the k-mer table lookups (run on the GPU box):
#   bash scripts/run_gather_bench.sh gpurun_out/r02_gather.jsonl
# Table sizes: 0.125 MB (L1/LDS scale), 3 MB (L2-resident compact table), 49 MB / 446 MB (round-1 tables of c2 / c5:
# Infinity Cache / HBM); request widths 16 / 32 / 64 B; 5 and 6 waves per SIMD.
set -euo pipefail
out=${1:-gpurun_out/gather.jsonl}
mkdir -p "$(dirname "$out")" build
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o build/gather_bench scripts/gather_bench.hip
: > "$out"
for mb in 0.125 3 49 446; do
  for req in 16 32 64; do
    for bpc in 5 6 8; do
      ./build/gather_bench $mb $req $bpc 4096 5 >> "$out"
    done
  done
done
cat "$out"

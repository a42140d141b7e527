#!/bin/bash
# One config's complete profile set for profiles/ (run on the GPU box): bash scripts/profile_round.sh <tag> <config> [steps]
#   -> gpurun_out/<tag>_<config>_{bench.json,summary.txt,traffic.json,sq.txt}
set -e
TAG=$1; CFG=$2; STEPS=${3:-5}
bash scripts/profile_gpu.sh ${TAG}_${CFG} $CFG $STEPS > gpurun_out/${TAG}_${CFG}_profile.log 2>&1
READS=$(python -c "import json;print(json.load(open('gpurun_out/${TAG}_${CFG}/traffic.json'))['reads_per_gpu'])")
bash scripts/pmc_sq.sh ${TAG}_${CFG}_sq $READS $CFG > gpurun_out/${TAG}_${CFG}_sq.txt 2>&1
cp gpurun_out/${TAG}_${CFG}/bench.json gpurun_out/${TAG}_${CFG}_bench.json
cp gpurun_out/${TAG}_${CFG}/summary.txt gpurun_out/${TAG}_${CFG}_summary.txt
cp gpurun_out/${TAG}_${CFG}/traffic.json gpurun_out/${TAG}_${CFG}_traffic.json
python scripts/merge_sq.py gpurun_out/${TAG}_${CFG}_traffic.json gpurun_out/${TAG}_${CFG}_sq.txt ${WAVES_PER_SIMD:-4} | cut -c1-600

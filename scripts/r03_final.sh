#!/bin/bash
# end-of-round check on the GPU box: smoke(), the whole GPU suite, the driver's bench command: bash scripts/r03_final.sh
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 gpurun_out/final_smoke.log | cut -c1-300
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/final_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -n 3 gpurun_out/final_gpu_suite.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; cut -c1-700 gpurun_out/final_bench.json

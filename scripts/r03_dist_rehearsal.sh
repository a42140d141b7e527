#!/bin/bash
# The several-rank path of bench.py on a one-GPU box: one rank under torchrun, RCCL communicator, barriers, end-of-job reduce through its
# collectives (all-gather of digests, all-reduce of the count vector). Not a scaling number: it shows that the path runs on RCCL at all and what
# the reduce costs per job.   bash scripts/r03_dist_rehearsal.sh
mkdir -p gpurun_out
for steps in 5 20; do
NB_BENCH_REHEARSE_DIST=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 1 --steps $steps --warmup 2 > gpurun_out/dist_rehearsal_$steps.json 2> gpurun_out/dist_rehearsal_$steps.err; rc=$?
echo "steps=$steps rc=$rc"
python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/dist_rehearsal_$steps.json").read().strip().splitlines()[-1])
    print({k: j.get(k) for k in ("value", "ms_per_step", "steps", "end_of_job_reduce_ms", "timed_region_ms", "classes", "oracle_self_consistency_ok")})
except Exception as e:
    print("no line:", e)
PY
tail -n 5 gpurun_out/dist_rehearsal_$steps.err
[ $rc -ne 0 ] && exit $rc
done
# is it the environment torchrun sets (OMP_NUM_THREADS=1)?
OMP_NUM_THREADS=1 timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('single process, OMP_NUM_THREADS=1:', j['value'], j['ms_per_step'])"
# reference point: the same steps in one process without the communicator
timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('single process:', j['value'], j['ms_per_step'])"

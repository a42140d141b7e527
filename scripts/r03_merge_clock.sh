#!/bin/bash
# stage clock of the end-of-job reduce (one rank through the collectives) + of the read-back that follows it in bench.py
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29529 NB_BENCH_REHEARSE_DIST=1 NB_MERGE_TIMING=1 NB_TIMING=1
timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>&1 >/dev/null | grep "merge\|nb_classes\|rows_to"

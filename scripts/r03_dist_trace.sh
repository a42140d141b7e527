#!/bin/bash
# kernel trace of the several-rank path with one rank (bench.py started directly with the rendezvous variables set, so that the program itself
# follows `--` under rocprofv3):   bash scripts/r03_dist_trace.sh
mkdir -p gpurun_out
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29521 NB_BENCH_REHEARSE_DIST=1
timeout -k 10 300 python bench.py --steps 20 --warmup 2 > gpurun_out/dist_direct.json 2> gpurun_out/dist_direct.err; echo "direct rc=$?"
python -c "
import json; j = json.loads(open('gpurun_out/dist_direct.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['timed_region_ms'])"
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/dist_trace
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dist_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/dist_trace.json 2> $GRAFT_REPO_ROOT/gpurun_out/dist_trace.err; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/dist_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last_nb = max(i for i, r in enumerate(rows) if "nb::" in r["Kernel_Name"] and "anchor" in r["Kernel_Name"])
t_ref = int(rows[last_nb]["End_Timestamp"])
print("kernels after the last anchor-path launch (start offset us, duration us, name):")
for r in rows[last_nb - 3:last_nb + 8]:
    print(f'{(int(r["Start_Timestamp"]) - t_ref) / 1e3:10.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f}  {r["Kernel_Name"][:110]}')
for f in glob.glob("gpurun_out/dist_trace/**/*memory_copy_trace.csv", recursive=True):
    cps = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    print("memory copies from 1 ms before the last anchor-path launch ended (start offset us, duration us, bytes, direction):")
    for r in cps:
        off = (int(r["Start_Timestamp"]) - t_ref) / 1e3
        if -1000 < off < 80000:
            print(f'{off:10.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} {r.get("Bytes", r.get("Size", "?")):>12} {r.get("Direction", "")}')
PY

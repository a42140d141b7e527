#!/bin/bash
# HIP API trace of the several-rank rehearsal: which calls precede the slow synchronize after the end-of-job reduce
mkdir -p gpurun_out
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29527 NB_BENCH_REHEARSE_DIST=1
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/dist_hip
timeout -k 10 400 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dist_hip -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --parity-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/dist_hip.json 2> $GRAFT_REPO_ROOT/gpurun_out/dist_hip.err; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/dist_hip/**/*hip_api_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the slow call: the longest hipStreamSynchronize after the last anchor kernel launch
k = sorted(csv.DictReader(open(glob.glob("gpurun_out/dist_hip/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
t_ref = max(int(r["End_Timestamp"]) for r in k if "nb_anchor" in r["Kernel_Name"])
after = [r for r in rows if int(r["Start_Timestamp"]) > t_ref]
slow = max((r for r in after[:3000] if "Synchronize" in r["Function"]), key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
i = after.index(slow)
print("slow call:", slow["Function"], (int(slow["End_Timestamp"]) - int(slow["Start_Timestamp"])) / 1e6, "ms, thread", slow["Thread_Id"])
print("the 60 calls before it (offset from the last anchor kernel's end in us, duration us, thread, function):")
for r in after[max(0, i - 60):i + 3]:
    print(f'{(int(r["Start_Timestamp"]) - t_ref) / 1e3:10.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} {r["Thread_Id"]} {r["Function"]}')
print("calls on OTHER threads during the slow call:")
for r in rows:
    if r["Thread_Id"] != slow["Thread_Id"] and int(r["End_Timestamp"]) > int(slow["Start_Timestamp"]) and int(r["Start_Timestamp"]) < int(slow["End_Timestamp"]):
        print(f'{(int(r["Start_Timestamp"]) - t_ref) / 1e3:10.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} {r["Thread_Id"]} {r["Function"]}')
PY

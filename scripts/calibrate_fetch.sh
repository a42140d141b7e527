#!/bin/bash
# FETCH_SIZE calibration for scattered gathers (guide: "other access widths are uncalibrated: calibrate on a known byte count in
# your own access pattern"): the gather microbenchmark reads a KNOWN number of random 16 / 32 / 64-byte slots; FETCH_SIZE of the
# same launches tells what the counter reports per request for each width and table size. Run on the GPU box:
#   bash scripts/calibrate_fetch.sh gpurun_out/r02_fetch_calibration
set -euo pipefail
out=${1:-gpurun_out/fetch_calibration}; mkdir -p "$out" build; export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o build/gather_bench scripts/gather_bench.hip
for mb in 3 49 446; do
  for req in 16 32 64; do
    tag=t${mb}_r${req}
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/$tag" -o g -- ./build/gather_bench $mb $req 5 1024 2 > "$out/$tag.json" 2> "$out/$tag.err" || { tail -3 "$out/$tag.err"; exit 1; }
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$out/${tag}_l2" -o g -- ./build/gather_bench $mb $req 5 1024 2 > /dev/null 2> "$out/${tag}_l2.err" || { tail -3 "$out/${tag}_l2.err"; exit 1; }
  done
done
python3 - "$out" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
print("table_MB req_B requests/launch useful_GB FETCH_SIZE_GB FETCH/useful bytes_per_request  TCC_HIT TCC_MISS TCC_REQ  (per request)")
rows = []
for f in sorted(glob.glob(os.path.join(out, "t*_r*.json"))):
    tag = os.path.basename(f)[:-5]
    j = json.loads(open(f).read().strip().splitlines()[-1])
    vals = {}
    for d in (tag, tag + "_l2"):
        for c in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
            acc = {}
            for r in csv.DictReader(open(c)):
                if "gather_kernel" in r["Kernel_Name"]:
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for k, v in acc.items():
                vals[k] = sum(v[1:]) / max(1, len(v) - 1) if len(v) > 1 else v[0]      # skip the warm-up launch
    n = j["requests_per_launch"]; fetch = vals.get("FETCH_SIZE", 0) * 1024
    row = dict(table_MB=j["table_MB"], req_bytes=j["req_bytes"], requests_per_launch=n, useful_bytes=j["useful_bytes_per_launch"], fetch_bytes=fetch,
               fetch_over_useful=fetch / j["useful_bytes_per_launch"], fetch_bytes_per_request=fetch / n,
               tcc_hit_per_request=vals.get("TCC_HIT_sum", 0) / n, tcc_miss_per_request=vals.get("TCC_MISS_sum", 0) / n, tcc_req_per_request=vals.get("TCC_REQ_sum", 0) / n)
    rows.append(row)
    print(f"{row['table_MB']:8.1f} {row['req_bytes']:4d} {n:14.0f} {row['useful_bytes']/1e9:8.2f} {fetch/1e9:10.2f} {row['fetch_over_useful']:8.3f} {row['fetch_bytes_per_request']:8.1f}"
          f"   {row['tcc_hit_per_request']:.3f} {row['tcc_miss_per_request']:.3f} {row['tcc_req_per_request']:.3f}")
json.dump(rows, open(os.path.join(out, "calibration.json"), "w"), indent=1)
PY

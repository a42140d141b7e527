#!/bin/bash
# Runs the 10x leg of e2e_bench at 2 M pairs and keeps the per-read TSV (gzip) under gpurun_out/ for host-side profiling.
set -e
export TMPDIR=/tmp
python - <<'PY'
import os, sys, gzip, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_bench
real_remove = os.remove
os.remove = lambda p: None
sys.argv = ["e2e_bench.py", "2000000", "16"]
e2e_bench.main()
with open("/tmp/e2e_tenx.tsv", "rb") as f, gzip.open("gpurun_out/e2e_tenx.tsv.gz", "wb", compresslevel=6) as g:
    shutil.copyfileobj(f, g)
PY
ls -la gpurun_out/e2e_tenx.tsv.gz

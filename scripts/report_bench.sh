#!/bin/bash
# Times `report` (native) on the per-read table of the 10x leg of e2e_bench, for 1 and all threads.
set -e
export TMPDIR=/tmp
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_bench
real_remove = os.remove
os.remove = lambda p: None
sys.argv = ["e2e_bench.py", "8000000", "16"]
e2e_bench.main()
from nimble_amd.report import report
for thr in ("1", "4", "16"):
    os.environ["NB_REPORT_THREADS"] = thr
    t = time.time(); report("/tmp/e2e_tenx.tsv", "/tmp/e2e_counts2.tsv"); print(f"[report] threads={thr}: {time.time() - t:.2f} s")
PY

#!/usr/bin/env python
"""Host-side speed of the BAM reader alone (no GPU work). usage: python scripts/bam_bench.py [records] [threads]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import e2e_bench  # noqa: E402
from nimble_amd.align import BamReader  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
thr = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
rng = np.random.default_rng(3)
p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "bb.bam")
e2e_bench.write_bam_fixed(p, rng.integers(0, 4, (n, 91), dtype=np.uint8), rng.integers(0, 4, (n, 16), dtype=np.uint8), rng.integers(0, 4, (n, 12), dtype=np.uint8))
for t in (1, 4, thr):
    t0 = time.perf_counter(); r = BamReader(p, t); m = 0
    while True:
        b = r.next_batch(1 << 20, 339)
        if b["n"] == 0:
            break
        m += b["n"]
    r.close()
    dt = time.perf_counter() - t0
    print(f"[bam] {m} records, {t} threads: {dt:.2f} s = {m / dt / 1e6:.2f} M records/s ({os.path.getsize(p) / 1e6:.0f} MB)")
os.remove(p)

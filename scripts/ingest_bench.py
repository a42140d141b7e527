#!/usr/bin/env python
"""Host-side stage timing of the 10x ingest (no GPU work): R1 barcode extraction, R2 packing, whitelist correction.
usage: python scripts/ingest_bench.py [pairs] [threads]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import e2e_bench  # noqa: E402
from nimble_amd import synth  # noqa: E402
from nimble_amd.align import FastqReader, NativeWhitelist  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
thr = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
d = os.environ.get("TMPDIR", "/tmp")
rng = np.random.default_rng(3)
_, seqs, _ = synth.synth_library(20, 50)
wl = rng.integers(0, 4, size=(100000, 16), dtype=np.uint8)
cells = wl[rng.integers(0, 10000, n)]
err = rng.random(n) < 0.01
pos = rng.integers(0, 16, n)
cells[err, pos[err]] = (cells[err, pos[err]] + 1) & 3
umis = rng.integers(0, 4, size=(n, 12), dtype=np.uint8)
f1, f2, wlp = os.path.join(d, "ib_R1.fastq"), os.path.join(d, "ib_R2.fastq"), os.path.join(d, "ib_wl.txt")
e2e_bench.write_fastq(f1, np.concatenate([cells, umis], axis=1), b"q")
e2e_bench.write_fastq(f2, synth.synth_reads(seqs, n, 91, seed=3), b"q")
open(wlp, "wb").write(b"\n".join(synth.BASES[w].tobytes() for w in wl) + b"\n")
for rep in range(2):
    t = time.perf_counter(); r = FastqReader(f1, thr); cbs = []
    while True:
        b = r.next_batch(1 << 20, barcodes_only=True, cb_len=16, umi_len=12)
        if b["n"] == 0:
            break
        cbs.append((b["cb"].copy(), b["cbq"].copy()))
    r.close(); t1 = time.perf_counter() - t
    t = time.perf_counter(); r = FastqReader(f2, thr); m = 0
    while True:
        b = r.next_batch(1 << 20, 339)
        if b["n"] == 0:
            break
        m += b["n"]
    r.close(); t2 = time.perf_counter() - t
    w = NativeWhitelist(wlp, thr)
    t = time.perf_counter()
    for cb, cq in cbs:
        w.correct(cb, cq)
    t3 = time.perf_counter() - t
    w.close()
    print(f"[ingest] {n} pairs, {thr} threads: R1 barcodes {t1:.2f} s, R2 pack {t2:.2f} s, whitelist {t3:.2f} s")
for p in (f1, f2, wlp):
    os.remove(p)

#!/usr/bin/env python
"""End-to-end (FASTQ on disk -> TSV [-> report]) timing on the GPU box. Not the headline metric: host-bound.
usage: python scripts/e2e_bench.py [n_reads] [cores]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nimble_amd import library, synth  # noqa: E402
from nimble_amd.align import TenxOptions, align  # noqa: E402
from nimble_amd.report import report  # noqa: E402


def write_fastq(path, codes, prefix):
    n, L = codes.shape
    seqs = synth.BASES[codes]                      # [n, L] uint8 ascii
    with open(path, "wb") as f:
        q = b"I" * L
        for lo in range(0, n, 200000):
            hi = min(n, lo + 200000)
            f.write(b"".join(b"@%s%d\n%s\n+\n%s\n" % (prefix, i, seqs[i].tobytes(), q) for i in range(lo, hi)))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    cores = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    d = os.environ.get("TMPDIR", "/tmp")
    names, seqs, lineage = synth.synth_library(20, 50)
    csv = os.path.join(d, "e2e_lib.csv")
    with open(csv, "w") as f:
        f.write("name,sequence,lineage\n")
        for nm, s, l in zip(names, seqs, lineage):
            f.write(f"{nm},{synth.codes_to_str(s)},{l}\n")
    lib = os.path.join(d, "e2e_lib.json")
    library.generate(csv, None, lib)
    # --- c2-like: single-end bulk ---
    codes = synth.synth_reads(seqs, n, 150, seed=2)
    fq = os.path.join(d, "e2e_reads.fastq")
    write_fastq(fq, codes, b"r")
    t0 = time.time()
    rc = align(lib, os.path.join(d, "e2e_bulk.tsv"), [fq], cores, "unstranded", "", None)
    t = time.time() - t0
    print(f"[e2e] bulk single-end: rc={rc} {n} reads in {t:.2f} s = {n / t / 1e6:.2f} M reads/s (FASTQ {os.path.getsize(fq) / 1e6:.0f} MB)")
    # --- c3-like: 10x R1 barcodes + R2 91 bp ---
    rng = np.random.default_rng(3)
    wl = rng.integers(0, 4, size=(100000, 16), dtype=np.uint8)
    cells = wl[rng.integers(0, 10000, n)]
    err = rng.random(n) < 0.01
    pos = rng.integers(0, 16, n)
    cells[err, pos[err]] = (cells[err, pos[err]] + 1) & 3
    umis = rng.integers(0, 4, size=(n // 4 + 1, 12), dtype=np.uint8)[rng.integers(0, n // 4 + 1, n)]
    r1 = np.concatenate([cells, umis], axis=1)
    r2 = synth.synth_reads(seqs, n, 91, seed=3)
    f1, f2, wlp = os.path.join(d, "e2e_R1.fastq"), os.path.join(d, "e2e_R2.fastq"), os.path.join(d, "e2e_wl.txt")
    write_fastq(f1, r1, b"q"); write_fastq(f2, r2, b"q")
    with open(wlp, "wb") as f:
        f.write(b"\n".join(synth.BASES[w].tobytes() for w in wl) + b"\n")
    out = os.path.join(d, "e2e_tenx.tsv")
    t0 = time.time()
    rc = align(lib, out, [f1, f2], cores, "unstranded", "", None, tenx=TenxOptions(wlp, 16, 12))
    t = time.time() - t0
    print(f"[e2e] 10x pairs: rc={rc} {n} pairs in {t:.2f} s = {n / t / 1e6:.2f} M pairs/s (TSV {os.path.getsize(out) / 1e6:.0f} MB)")
    t0 = time.time()
    report(out, os.path.join(d, "e2e_counts.tsv"))
    t = time.time() - t0
    print(f"[e2e] report (native): {t:.2f} s, {sum(1 for _ in open(os.path.join(d, 'e2e_counts.tsv')))} output rows")
    for p in (csv, lib, fq, f1, f2, wlp, out, os.path.join(d, "e2e_bulk.tsv"), os.path.join(d, "e2e_counts.tsv")):
        os.remove(p)


if __name__ == "__main__":
    main()

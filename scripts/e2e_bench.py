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


def write_bam_fixed(path, codes, cells, umis):
    """Unaligned single-end BAM with CB:Z / UB:Z tags, fixed-width records assembled with numpy, BGZF blocks of 60 KB."""
    import struct
    import zlib
    n, L = codes.shape
    nib = np.array([1, 2, 4, 8, 15], dtype=np.uint8)[codes]                # A C G T N -> BAM 4-bit codes
    if L & 1:
        nib = np.concatenate([nib, np.zeros((n, 1), np.uint8)], axis=1)
    seq = (nib[:, 0::2] << 4) | nib[:, 1::2]
    name = np.frombuffer(b"".join(b"r%09d\x00" % i for i in range(n)), dtype=np.uint8).reshape(n, 11)
    cb = np.concatenate([np.tile(np.frombuffer(b"CBZ", np.uint8), (n, 1)), synth.BASES[cells], np.tile(np.frombuffer(b"-1\x00", np.uint8), (n, 1))], axis=1)
    ub = np.concatenate([np.tile(np.frombuffer(b"UBZ", np.uint8), (n, 1)), synth.BASES[umis], np.zeros((n, 1), np.uint8)], axis=1)
    body = struct.pack("<iiBBHHHIiii", -1, -1, 11, 0, 4680, 0, 4, L, -1, -1, 0)
    size = 32 + 11 + seq.shape[1] + L + cb.shape[1] + ub.shape[1]
    fixed = np.tile(np.frombuffer(struct.pack("<i", size) + body, np.uint8), (n, 1))
    recs = np.concatenate([fixed, name, seq, np.full((n, L), 30, np.uint8), cb, ub], axis=1)
    text = b"@HD\tVN:1.6\tSO:unsorted\n"
    raw = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", 0) + recs.tobytes()
    with open(path, "wb") as f:
        for i in range(0, len(raw) + 1, 60000):
            data = raw[i:i + 60000]
            comp = zlib.compressobj(1, zlib.DEFLATED, -15)
            cdata = comp.compress(data) + comp.flush()
            f.write(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(cdata) + 25) + cdata +
                    struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
            if not data:
                break
        else:
            f.write(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, 27) + b"\x03\x00" + struct.pack("<II", 0, 0))


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
    # --- the same reads as a gzip FASTQ (one deflate stream: inflate is serial, csrc/nb_inflate.cpp) ---
    import subprocess
    ngz = min(n, 4_000_000)
    fqz = os.path.join(d, "e2e_reads_gz.fastq.gz")
    with open(fq, "rb") as src, open(fqz, "wb") as dst:
        head = subprocess.Popen(["gzip", "-1", "-c"], stdin=subprocess.PIPE, stdout=dst)
        left = ngz * 4
        for line in src:
            head.stdin.write(line)
            left -= 1
            if left == 0:
                break
        head.stdin.close(); head.wait()
    t0 = time.time()
    rc = align(lib, os.path.join(d, "e2e_bulk_gz.tsv"), [fqz], cores, "unstranded", "", None)
    t = time.time() - t0
    print(f"[e2e] bulk single-end, gzip FASTQ: rc={rc} {ngz} reads in {t:.2f} s = {ngz / t / 1e6:.2f} M reads/s (gz {os.path.getsize(fqz) / 1e6:.0f} MB)")
    # --- and as a BGZF file (bgzip): the same reader inflates its blocks in parallel ---
    import ctypes as ct
    from nimble_amd import _lib
    fqb = os.path.join(d, "e2e_reads_bgzf.fastq.gz")
    with open(fq, "rb") as src, open(fqb, "wb") as dst:
        left = ngz * 4
        piece = []
        size = 0
        def flush():
            text = np.frombuffer(b"".join(piece), dtype=np.uint8)
            cap = text.shape[0] + (text.shape[0] >> 9) + 64 * ((text.shape[0] >> 20) + 1) + 64
            out, out_len = np.empty(cap, dtype=np.uint8), ct.c_int64()
            _lib.check(_lib.lib().nb_gzip_members(text.ctypes.data, text.shape[0], 1, cores, out.ctypes.data, cap, ct.byref(out_len)))
            dst.write(out[:out_len.value].tobytes())
        for line in src:
            piece.append(line); size += len(line); left -= 1
            if size > (64 << 20) or left == 0:
                flush(); piece, size = [], 0
            if left == 0:
                break
    t0 = time.time()
    rc = align(lib, os.path.join(d, "e2e_bulk_bgzf.tsv"), [fqb], cores, "unstranded", "", None)
    t = time.time() - t0
    print(f"[e2e] bulk single-end, BGZF FASTQ: rc={rc} {ngz} reads in {t:.2f} s = {ngz / t / 1e6:.2f} M reads/s (bgzf {os.path.getsize(fqb) / 1e6:.0f} MB)")
    for p in (fqz, os.path.join(d, "e2e_bulk_gz.tsv"), fqb, os.path.join(d, "e2e_bulk_bgzf.tsv")):
        os.remove(p)
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
    # --- the reference's main single-cell route: a BAM with CB/UB tags (what fastq-to-bam / cellranger produce) ---
    bam = os.path.join(d, "e2e.bam")
    write_bam_fixed(bam, r2, cells, umis)
    out_bam = os.path.join(d, "e2e_bam.tsv.gz")      # the reference's aligner writes a gzip table for BAM input
    t0 = time.time()
    rc = align(lib, out_bam, [bam], cores, "unstranded", "", None)
    t = time.time() - t0
    print(f"[e2e] BAM (CB/UB tags): rc={rc} {n} records in {t:.2f} s = {n / t / 1e6:.2f} M records/s (BAM {os.path.getsize(bam) / 1e6:.0f} MB, TSV {os.path.getsize(out_bam) / 1e6:.0f} MB)")
    for p in (bam, out_bam):
        os.remove(p)
    for p in (csv, lib, fq, f1, f2, wlp, out, os.path.join(d, "e2e_bulk.tsv"), os.path.join(d, "e2e_counts.tsv")):
        os.remove(p)


if __name__ == "__main__":
    main()

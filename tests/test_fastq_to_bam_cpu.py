"""`fastq-to-bam` (csrc/nb_ingest.cpp nb_fastq_to_bam) against the rules of the reference's process_pair
(/root/reference/nimble/fastq_barcode_processor.py:145-210), restated here in Python with barcodes.BarcodeCorrector, and an
independent BAM parser (struct + zlib)."""
import gzip
import struct
import zlib

import numpy as np

from tests import barcodes_twin as barcodes
from nimble_amd.__main__ import main as cli


def parse_bam(path):
    data = open(path, "rb").read()
    raw, o = b"", 0
    while o < len(data):                                   # BGZF members
        bsize = int.from_bytes(data[o + 16:o + 18], "little") + 1
        raw += zlib.decompress(data[o:o + bsize], 31)
        o += bsize
    assert data[-28:] == bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])   # EOF marker
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].decode()
    n_ref = struct.unpack_from("<i", raw, 8 + l_text)[0]
    p = 12 + l_text
    recs = []
    while p < len(raw):
        size = struct.unpack_from("<i", raw, p)[0]
        ref, pos, l_name, mapq, bin_, n_cig, flag, l_seq, nref, npos, tlen = struct.unpack_from("<iiBBHHHIiii", raw, p + 4)
        q = p + 36
        name = raw[q:q + l_name - 1].decode(); q += l_name + 4 * n_cig
        packed = raw[q:q + (l_seq + 1) // 2]; q += (l_seq + 1) // 2
        seq = "".join("=ACMGRSVTWYHKDBN"[(packed[i >> 1] >> (0 if i & 1 else 4)) & 15] for i in range(l_seq))
        qual = list(raw[q:q + l_seq]); q += l_seq
        tags = []
        while q < p + 4 + size:
            tag, ty = raw[q:q + 2].decode(), chr(raw[q + 2])
            assert ty == "Z"
            e = raw.index(b"\0", q + 3)
            tags.append((tag, raw[q + 3:e].decode())); q = e + 1
        recs.append(dict(name=name, flag=flag, seq=seq, qual=qual, tags=tags, ref=ref, pos=pos, mapq=mapq, bin=bin_, nref=nref, npos=npos, tlen=tlen))
        p += 4 + size
    return text, n_ref, recs


def test_fastq_to_bam_matches_the_reference_rules(built, tmp_path, capsys):
    rng = np.random.default_rng(4)
    wl = sorted({"".join("ACGT"[c] for c in rng.integers(0, 4, 16)) for _ in range(500)})
    (tmp_path / "wl.txt").write_text("\n".join(wl) + "\n")
    n = 70_000                                             # more than one 65 536-pair sub-batch of the converter
    r1, r2 = [], []
    for i in range(n):
        cb = list(wl[int(rng.integers(0, len(wl)))])
        for _ in range(int(rng.choice([0, 0, 0, 1, 2]))):
            cb[int(rng.integers(0, 16))] = "ACGTN"[int(rng.integers(0, 5))]
        kind = i % 97
        rest = "".join("ACGTN"[c] for c in rng.integers(0, 5, int(rng.integers(1, 40))))
        s1 = "".join(cb) + "".join("ACGT"[c] for c in rng.integers(0, 4, 12)) + rest
        if kind == 5:
            s1 = s1[:28]                                       # nothing after CB + UMI
        if kind == 6:
            s1 = s1[:20]                                       # too short
        q1 = "".join(chr(33 + int(x)) for x in rng.integers(2, 41, len(s1)))
        s2 = "".join("ACGTNRY"[c] for c in rng.integers(0, 7, int(rng.integers(20, 92))))
        q2 = "".join(chr(33 + int(x)) for x in rng.integers(2, 41, len(s2)))
        n1, n2 = f"read{i}/1 1:N:0:ACGT", f"read{i}/2 2:N:0:ACGT"
        if kind == 7:
            n2 = f"other{i}/2"
        if kind == 8:
            n1, n2 = f"plain{i}", f"plain{i} extra"
        r1.append((n1, s1, q1)); r2.append((n2, s2, q2))
    with gzip.open(tmp_path / "R1.fastq.gz", "wt") as f:
        f.write("".join(f"@{a}\n{s}\n+\n{q}\n" for a, s, q in r1))
    with open(tmp_path / "R2.fastq", "w") as f:
        f.write("".join(f"@{a}\n{s}\n+\n{q}\n" for a, s, q in r2) + "@tail\nACGT\n+\nIIII\n")   # one record more than R1: zip() stops
    out = tmp_path / "out.bam"
    assert cli(["fastq-to-bam", "--r1-fastq", str(tmp_path / "R1.fastq.gz"), "--r2-fastq", str(tmp_path / "R2.fastq"), "--map", str(tmp_path / "wl.txt"),
                "--output", str(out), "-c", "3"]) == 0
    printed = capsys.readouterr().out
    # the reference's rules, restated
    corr = barcodes.BarcodeCorrector(set(wl))
    want, stats = [], dict(name_mismatch=0, too_short=0, no_remaining_seq=0, none=0)
    for (a, s1, q1), (c, s2, q2) in zip(r1, r2):
        ida, idc = a.split()[0].removesuffix("/1"), c.split()[0].removesuffix("/2")
        if ida != idc:
            stats["name_mismatch"] += 1; continue
        if len(s1) < 28:
            stats["too_short"] += 1; continue
        if len(s1) == 28:
            stats["no_remaining_seq"] += 1; continue
        fixed = corr.correct(s1[:16], q1[:16])
        if fixed is None:
            stats["none"] += 1; continue
        tags = [("CB", fixed), ("UB", s1[16:28])]
        want.append(dict(name=ida, flag=77, seq=s1[28:].upper(), qual=[ord(x) - 33 for x in q1[28:]], tags=tags))
        want.append(dict(name=ida, flag=141, seq=s2.upper(), qual=[ord(x) - 33 for x in q2], tags=tags))
    text, n_ref, recs = parse_bam(out)
    assert text == "@HD\tVN:1.6\tSO:queryname\n@PG\tID:nimble-fastq-to-bam\tPN:nimble\tVN:1.2\tCL:whitelist-based CB correction\n" and n_ref == 0
    assert len(recs) == len(want) and len(want) > 2 * 0.6 * n
    for g, w in zip(recs, want):
        assert {k: g[k] for k in w} == w
        assert (g["ref"], g["pos"], g["mapq"], g["bin"], g["nref"], g["npos"], g["tlen"]) == (-1, -1, 0, 4680, -1, -1, 0)
    assert f"Total read pairs: {n}" in printed and f"Written pairs: {len(want) // 2}" in printed
    assert f"Name mismatch: {stats['name_mismatch']}" in printed and f"Too short: {stats['too_short']}" in printed
    assert f"No remaining sequence: {stats['no_remaining_seq']}" in printed and f"No valid correction: {stats['none']}" in printed
    assert min(stats.values()) > 0
    # and the engine's own BAM reader pairs it back up
    from nimble_amd import engine
    from nimble_amd.align import BamReader
    r = BamReader(str(out), 2)
    got = 0
    while True:
        b = r.next_batch(20_000, 200)
        if b["n"] == 0:
            break
        assert b["paired"]
        for i in range(0, b["n"], 501):
            w1, w2 = want[2 * (got + i)], want[2 * (got + i) + 1]
            norm = lambda s: "".join(ch if ch in "ACGT" else "N" for ch in s)
            assert engine.unpack_read(b["words"][i], int(b["lens"][i]), b["max_len"]) == norm(w1["seq"])
            assert engine.unpack_read(b["mate_words"][i], int(b["mate_lens"][i]), b["mate_max_len"]) == norm(w2["seq"])
            assert b["cb"][i].tobytes().rstrip(b"\0").decode() == w1["tags"][0][1] and b["umi"][i].tobytes().rstrip(b"\0").decode() == w1["tags"][1][1]
        got += b["n"]
    r.close()
    assert got == len(want) // 2

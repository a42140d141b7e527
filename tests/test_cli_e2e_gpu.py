"""End to end on the GPU: lib.json + FASTQ -> `align` -> TSV -> `report`, checked against the oracle's calls pushed
through the same host rules (docs/SPEC.md §5) and, for the reducer, through the golden-pinned report()."""
import gzip
import json
import os

import numpy as np
import pytest

from nimble_amd import engine, library, synth
from nimble_amd.__main__ import main as cli
from tests.helpers import library_strings

pytestmark = pytest.mark.gpu


def write_fastq(path, names, seqs, quals=None):
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wt") as f:
        for i, (n, s) in enumerate(zip(names, seqs)):
            q = quals[i] if quals else "I" * len(s)
            f.write(f"@{n}\n{s}\n+\n{q}\n")


def make_library(tmp_path, n_loci=5, alleles=8, name="hla", group_on="", **cfg):
    names, seqs, strs, lineage = library_strings(n_loci, alleles, locus_len=600, seed=77)
    csv_path = tmp_path / f"{name}.csv"
    with open(csv_path, "w") as f:
        f.write("name,sequence,lineage\n")
        for n, s, l in zip(names, strs, lineage):
            f.write(f"{n},{s},{l}\n")
    out = tmp_path / f"{name}.json"
    assert cli(["generate", "--file", str(csv_path), "--output_path", str(out)]) == 0
    doc = json.load(open(out))
    doc[0].update(cfg)
    doc[0]["group_on"] = group_on
    json.dump(doc, open(out, "w"), indent=2)
    return out, names, seqs, strs, lineage


def expected_bulk(lib_path, reads, strand="unstranded"):
    from oracle import oracle
    lib = library.Library(str(lib_path))
    c = lib.config
    cfg = engine.make_config(k=lib.k, score_threshold=c["score_threshold"], score_filter=c["score_filter"], score_percent=c["score_percent"],
                             num_mismatches=c["num_mismatches"], discard_multiple_matches=c["discard_multiple_matches"],
                             intersect_level=c["intersect_level"], require_valid_pair=c["require_valid_pair"], strand_filter=strand)
    ora = oracle.OracleIndex(lib.sequences, lib.k).align_strs(cfg, reads, n_threads=4)
    return lib, ora


@pytest.mark.parametrize("mm", [0, 2])
def test_bulk_single_end(built, tmp_path, mm):
    """(mm = the library's num_mismatches, /root/reference/nimble/types.py:15: 2 = the anchor kernels pass over bases, SPEC 3)"""
    lib_path, names, seqs, strs, lineage = make_library(tmp_path, max_hits_to_report=4, num_mismatches=mm)
    codes = synth.synth_reads(seqs, 4000, 120, seed=3, err=0.02)
    reads = [synth.codes_to_str(c) for c in codes]
    fq = tmp_path / "reads.fastq.gz"
    write_fastq(fq, [f"r{i}" for i in range(len(reads))], reads)
    out = tmp_path / "out.tsv"
    assert cli(["align", "--reference", str(lib_path), "--output", str(out), "--input", str(fq)]) == 0
    lib, ora = expected_bulk(lib_path, reads)
    want = {}
    for i in np.nonzero(ora["reason"] == 0)[0]:
        feats = lib.feature_list(engine.bits_to_members(ora["set_bits"][i]))
        if len(feats) <= 4:
            k = ",".join(feats); want[k] = want.get(k, 0) + 1
    lines = open(out).read().splitlines()
    assert lines[0] == "nimble_features\tnimble_score"
    got = {l.split("\t")[0]: int(l.split("\t")[1]) for l in lines[1:]}
    assert got == want and [l.split("\t")[0] for l in lines[1:]] == sorted(want)


@pytest.mark.parametrize("batch_reads", [1 << 20, 25_000])
def test_paired_fastq_ragged_lengths(built, tmp_path, batch_reads):
    """A biological pair from two FASTQ files with ragged read lengths: the readers pack each batch at the layout of its own
    longest read and end a batch early before a longer one, independently for R1 and R2 — mates must stay paired. With
    batch_reads = 25 000 the job has four outer batches and the early batch ends fall inside later ones: the top-up of the shorter
    side then runs while the PREVIOUS batch is still being uploaded (it used to overwrite that batch's pinned buffers)."""
    lib_path, names, seqs, strs, lineage = make_library(tmp_path, intersect_level=1)
    n = 90_000
    rng = np.random.default_rng(21)
    c1, c2 = synth.synth_reads(seqs, n, 140, seed=5), synth.synth_reads(seqs, n, 140, seed=6)
    l1, l2 = rng.integers(30, 100, n), rng.integers(30, 100, n)
    l1[40_000], l2[70_000] = 140, 133                       # late long reads: R1 and R2 batches end at different places
    r1 = [synth.codes_to_str(c[:l]) for c, l in zip(c1, l1)]
    r2 = [synth.codes_to_str(c[:l]) for c, l in zip(c2, l2)]
    f1, f2 = tmp_path / "R1.fastq", tmp_path / "R2.fastq.gz"
    write_fastq(f1, [f"p{i}" for i in range(n)], r1)
    write_fastq(f2, [f"p{i}" for i in range(n)], r2)
    out = tmp_path / "pairs.tsv"
    from nimble_amd.align import align
    assert align(str(lib_path), str(out), [str(f1), str(f2)], 4, "unstranded", "", None, batch_reads=batch_reads) == 0
    from oracle import oracle
    lib = library.Library(str(lib_path))
    c = lib.config
    cfg = engine.make_config(k=lib.k, score_threshold=c["score_threshold"], score_filter=c["score_filter"], score_percent=c["score_percent"],
                             num_mismatches=c["num_mismatches"], discard_multiple_matches=c["discard_multiple_matches"],
                             intersect_level=c["intersect_level"], require_valid_pair=c["require_valid_pair"], strand_filter="unstranded")
    ora = oracle.OracleIndex(lib.sequences, lib.k).align_strs(cfg, r1, r2, n_threads=8)
    want = {}
    for i in np.nonzero(ora["reason"] == 0)[0]:
        feats = lib.feature_list(engine.bits_to_members(ora["set_bits"][i]))
        if len(feats) <= c["max_hits_to_report"]:
            k = ",".join(feats); want[k] = want.get(k, 0) + 1
    lines = open(out).read().splitlines()[1:]
    assert {l.split("\t")[0]: int(l.split("\t")[1]) for l in lines} == want and len(want) > 10
    short = tmp_path / "R2short.fastq"
    write_fastq(short, [f"p{i}" for i in range(n - 1)], r2[:-1])
    assert cli(["align", "--reference", str(lib_path), "--output", str(out), "--input", str(f1), str(short)]) == 1   # record counts differ


def test_two_libraries_group_on_and_naming(built, tmp_path):
    """Several -r libraries in one pass: outputs named with append_path_string (__main__.py:182-189); group_on."""
    a, *_ = make_library(tmp_path, name="mhc_a")
    b, names, seqs, strs, lineage = make_library(tmp_path, name="mhc_b", group_on="lineage", discard_multi_hits=1)
    codes = synth.synth_reads(seqs, 2000, 150, seed=9)
    reads = [synth.codes_to_str(c) for c in codes]
    fq = tmp_path / "reads.fastq"
    write_fastq(fq, [f"r{i}" for i in range(len(reads))], reads)
    out = tmp_path / "res.tsv.gz"
    assert cli(["align", "--reference", f"{a},{b}", "--output", str(out), "--input", str(fq), "--strand_filter", "none"]) == 0
    assert os.path.exists(tmp_path / "res.mhc_a.tsv.gz") and os.path.exists(tmp_path / "res.mhc_b.tsv.gz")
    lib, ora = expected_bulk(b, reads, "none")
    want = {}
    for i in np.nonzero(ora["reason"] == 0)[0]:
        feats = lib.feature_list(engine.bits_to_members(ora["set_bits"][i]))
        if len(feats) <= 1:
            want[feats[0]] = want.get(feats[0], 0) + 1
    lines = gzip.open(tmp_path / "res.mhc_b.tsv.gz", "rt").read().splitlines()[1:]
    assert {l.split("\t")[0]: int(l.split("\t")[1]) for l in lines} == want and set(want) <= set(lineage)


def test_tenx_per_read_rows_then_report(built, tmp_path):
    lib_path, names, seqs, strs, lineage = make_library(tmp_path)
    rng = np.random.default_rng(5)
    n = 3000
    codes = synth.synth_reads(seqs, n, 91, seed=4)
    r2 = [synth.codes_to_str(c) for c in codes]
    wl = ["".join("ACGT"[c] for c in rng.integers(0, 4, 16)) for _ in range(40)]
    cells = [wl[i] for i in rng.integers(0, 8, n)]
    umis = ["".join("ACGT"[c] for c in rng.integers(0, 4, 12)) for _ in range(n // 4)]
    umi_of = [umis[i] for i in rng.integers(0, len(umis), n)]
    raw_cb = []
    for c in cells:
        if rng.random() < 0.05:
            j = int(rng.integers(0, 16)); c = c[:j] + "ACGT"[("ACGT".index(c[j]) + 1) % 4] + c[j + 1:]
        raw_cb.append(c)
    r1 = [c + u + "TTTTTTTT" for c, u in zip(raw_cb, umi_of)]
    (tmp_path / "wl.txt").write_text("\n".join(wl) + "\n")
    write_fastq(tmp_path / "R1.fastq.gz", [f"q{i}/1" for i in range(n)], r1)
    write_fastq(tmp_path / "R2.fastq.gz", [f"q{i}/2" for i in range(n)], r2)
    out = tmp_path / "per_read.tsv"
    assert cli(["align", "--reference", str(lib_path), "--output", str(out), "--input", str(tmp_path / "R1.fastq.gz"),
                str(tmp_path / "R2.fastq.gz"), "--tenx", "--map", str(tmp_path / "wl.txt"), "--report-output", str(tmp_path / "fused.tsv"),
                "--report-threshold", "0.1"]) == 0
    lib, ora = expected_bulk(lib_path, r2)
    from tests import barcodes_twin as barcodes
    corr = barcodes.BarcodeCorrector(set(wl))
    want = []
    for i in range(n):
        cb = corr.correct(raw_cb[i], "I" * 16)
        if ora["reason"][i] == 0 and cb is not None:
            feats = lib.feature_list(engine.bits_to_members(ora["set_bits"][i]))
            if len(feats) <= 10:
                want.append("\t".join([",".join(feats), "1", cb, umi_of[i]] + [str(x) for x in ora["cov"][i]]))
    lines = open(out).read().splitlines()
    assert lines[0].split("\t")[:4] == ["nimble_features", "nimble_score", "r1_CB", "r1_UB"]
    assert lines[1:] == want and len(want) > 1000
    # reducer on the per-read table: equal to feeding the expected rows through the golden-pinned report()
    from nimble_amd.report import report
    (tmp_path / "exp.tsv").write_text("\n".join([lines[0]] + want) + "\n")
    assert cli(["report", "-i", str(out), "-o", str(tmp_path / "counts.tsv")]) == 0
    report(str(tmp_path / "exp.tsv"), str(tmp_path / "counts_exp.tsv"))
    got = open(tmp_path / "counts.tsv").read()
    assert got == open(tmp_path / "counts_exp.tsv").read() and len(got.splitlines()) > 5
    assert all(len(l.split("\t")) == 3 for l in got.splitlines())
    # fused per-cell counting (no TSV round trip) == report on the TSV, same threshold
    report(str(out), str(tmp_path / "counts_01.tsv"), threshold=0.1, engine="python")
    assert open(tmp_path / "fused.tsv").read() == open(tmp_path / "counts_01.tsv").read()


def test_bam_input_single_end_and_pairs(built, tmp_path):
    """`align --input x.bam`: cellranger-style single-end records with CB/UB tags (some reverse-strand, some without
    tags) and a name-collated paired uBAM as the reference's fastq-to-bam writes it."""
    from oracle import oracle
    from tests.helpers import write_bam
    lib_path, names, seqs, strs, lineage = make_library(tmp_path, intersect_level=1)
    rng = np.random.default_rng(8)
    comp = str.maketrans("ACGTN", "TGCAN")
    n = 2500
    codes = synth.synth_reads(seqs, n, 98, seed=12)
    reads = [synth.codes_to_str(c) for c in codes]
    recs, cbs, ubs = [], [], []
    for i, s in enumerate(reads):
        cb = "" if i % 17 == 0 else "".join("ACGT"[c] for c in rng.integers(0, 4, 16)) + "-1"
        ub = "".join("ACGT"[c] for c in rng.integers(0, 4, 12))
        rev = bool(rng.random() < 0.5)
        tags = {"NH": 1, "UB": ub}
        if cb:
            tags["CB"] = cb
        recs.append({"name": f"r{i}", "flag": 0x10 if rev else 0, "seq": s.translate(comp)[::-1] if rev else s, "tags": tags})
        if i % 50 == 0:
            recs.append({"name": f"r{i}", "flag": 0x100, "seq": "ACGTACGT", "tags": {"CB": "zz", "UB": "zz"}})
        cbs.append(cb); ubs.append(ub)
    bam = tmp_path / "possorted.bam"
    write_bam(bam, recs)
    out = tmp_path / "bam_out.tsv.gz"
    assert cli(["align", "--reference", str(lib_path), "--output", str(out), "--input", str(bam), "--report-output", str(tmp_path / "bam_fused.tsv")]) == 0
    lib, ora = expected_bulk(lib_path, reads)
    want = []
    for i in range(n):
        if ora["reason"][i] == 0:
            feats = lib.feature_list(engine.bits_to_members(ora["set_bits"][i]))
            want.append("\t".join([",".join(feats), "1", cbs[i], ubs[i]] + [str(x) for x in ora["cov"][i]]))
    lines = gzip.open(out, "rt").read().split("\n")
    assert lines[0].split("\t")[:4] == ["nimble_features", "nimble_score", "r1_CB", "r1_UB"] and lines[-1] == ""
    assert lines[1:-1] == want and len(want) > 1000
    # report drops the rows without a cell barcode (null r1_CB, __main__.py:244) and counts the rest
    assert cli(["report", "-i", str(out), "-o", str(tmp_path / "bam_counts.tsv")]) == 0    # .gz input, as pandas infers it
    from nimble_amd.report import report
    (tmp_path / "exp.tsv").write_text("\n".join([lines[0]] + want) + "\n")
    report(str(tmp_path / "exp.tsv"), str(tmp_path / "exp_counts.tsv"), engine="python")
    got = open(tmp_path / "bam_counts.tsv").read()
    assert got == open(tmp_path / "exp_counts.tsv").read() and len(got.splitlines()) > 100
    assert not any(l.endswith("\t") for l in got.splitlines())
    assert open(tmp_path / "bam_fused.tsv").read() == got          # fused path, incl. the rows without CB dropped
    # name-collated pairs (flags 77 / 141)
    r1 = [synth.codes_to_str(c) for c in synth.synth_reads(seqs, 800, 90, seed=21)]
    r2 = [synth.codes_to_str(c) for c in synth.synth_reads(seqs, 800, 75, seed=22)]
    recs = []
    for i in range(800):
        t = {"CB": f"CELL{i % 7}", "UB": f"UMI{i % 90:03d}"}
        recs += [{"name": f"p{i}", "flag": 77, "seq": r1[i], "tags": t}, {"name": f"p{i}", "flag": 141, "seq": r2[i], "tags": t}]
    write_bam(tmp_path / "pairs.bam", recs)
    out2 = tmp_path / "pairs.tsv"
    assert cli(["align", "--reference", str(lib_path), "--output", str(out2), "--input", str(tmp_path / "pairs.bam")]) == 0
    c = lib.config
    cfg = engine.make_config(k=lib.k, score_threshold=c["score_threshold"], score_filter=c["score_filter"], score_percent=c["score_percent"],
                             intersect_level=c["intersect_level"])
    ora = oracle.OracleIndex(lib.sequences, lib.k).align_strs(cfg, r1, r2, n_threads=4)
    want2 = []
    for i in range(800):
        if ora["reason"][i] == 0:
            feats = lib.feature_list(engine.bits_to_members(ora["set_bits"][i]))
            want2.append("\t".join([",".join(feats), str(ora["mates"][i]), f"CELL{i % 7}", f"UMI{i % 90:03d}"] + [str(x) for x in ora["cov"][i]]))
    assert open(out2).read().split("\n")[1:-1] == want2 and len(want2) > 300


def test_bam_pairs_in_any_record_order_and_orphans(built, tmp_path):
    """A paired BAM that is NOT collated by name (mates far apart, as after a coordinate sort) plus reads whose mate is missing:
    the reader pairs by read name and hands orphans on with an empty partner; calls equal the oracle's on the same templates."""
    from oracle import oracle
    from tests.helpers import write_bam
    lib_path, names, seqs, strs, lineage = make_library(tmp_path, intersect_level=1)
    lib = library.Library(str(lib_path))
    rng = np.random.default_rng(31)
    n = 600
    r1 = [synth.codes_to_str(c) for c in synth.synth_reads(seqs, n, 90, seed=41)]
    r2 = [synth.codes_to_str(c) for c in synth.synth_reads(seqs, n, 75, seed=42)]
    recs = []
    for i in range(n):
        t = {"CB": f"CELL{i:04d}", "UB": f"UMI{i % 90:03d}"}
        if i % 40 != 7:
            recs.append({"name": f"p{i}", "flag": 77, "seq": r1[i], "tags": t})
        else:
            r1[i] = ""                                                  # orphan: only the second mate is in the file
        if i % 40 != 13:
            recs.append({"name": f"p{i}", "flag": 141, "seq": r2[i], "tags": t})
        else:
            r2[i] = ""
    recs = [recs[j] for j in rng.permutation(len(recs))]
    write_bam(tmp_path / "shuffled.bam", recs)
    out = tmp_path / "shuffled.tsv"
    assert cli(["align", "--reference", str(lib_path), "--output", str(out), "--input", str(tmp_path / "shuffled.bam")]) == 0
    c = lib.config
    cfg = engine.make_config(k=lib.k, score_threshold=c["score_threshold"], score_filter=c["score_filter"], score_percent=c["score_percent"],
                             intersect_level=c["intersect_level"])
    ora = oracle.OracleIndex(lib.sequences, lib.k).align_strs(cfg, r1, r2, n_threads=4)
    want = set()
    for i in range(n):
        if ora["reason"][i] == 0:
            feats = lib.feature_list(engine.bits_to_members(ora["set_bits"][i]))
            want.add("\t".join([",".join(feats), str(ora["mates"][i]), f"CELL{i:04d}", f"UMI{i % 90:03d}"] + [str(x) for x in ora["cov"][i]]))
    got = open(out).read().split("\n")[1:-1]
    assert len(got) == len(set(got)) and set(got) == want and len(want) > 300     # (row order follows the pairing order, not the file's)


def test_two_ranks_equal_one_rank(built, tmp_path):
    """`align --devices 0,0`: two processes (ranks sharing this box's one GPU, gloo for the collectives; on a multi-GPU node
    `--devices 0,1` runs the same code over RCCL), end-of-job merge of nimble_amd/merge.py: the bulk table, the per-read table (.gz)
    and the fused per-cell counts must equal the single-process run. Twice: each rank parsing its own byte range of the FASTQ files
    (the default for plain FASTQ: same rows in the same order, the gzip members differ because the batches do) and batches dealt
    round-robin (NB_RANGE_SHARDING=0, what compressed input gets: byte-identical files)."""
    import subprocess
    import sys
    lib_path, names, seqs, strs, lineage = make_library(tmp_path)
    rng = np.random.default_rng(15)
    n = 9000
    codes = synth.synth_reads(seqs, n, 100, seed=14)
    r2 = [synth.codes_to_str(c) for c in codes]
    wl = ["".join("ACGT"[c] for c in rng.integers(0, 4, 16)) for _ in range(60)]
    umis = ["".join("ACGT"[c] for c in rng.integers(0, 4, 12)) for _ in range(n // 3)]
    r1 = [wl[i] + umis[j] for i, j in zip(rng.integers(0, 20, n), rng.integers(0, len(umis), n))]
    (tmp_path / "wl.txt").write_text("\n".join(wl) + "\n")
    write_fastq(tmp_path / "R1.fastq", [f"q{i}" for i in range(n)], r1)
    write_fastq(tmp_path / "R2.fastq", [f"q{i}" for i in range(n)], r2)
    env = dict(os.environ, NB_BATCH_READS="1000", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

    def run(tag, extra, env=env):
        bulk = [sys.executable, "-m", "nimble_amd", "align", "--reference", str(lib_path), "--output", str(tmp_path / f"{tag}_bulk.tsv"),
                "--input", str(tmp_path / "R2.fastq")] + extra
        tenx = [sys.executable, "-m", "nimble_amd", "align", "--reference", str(lib_path), "--output", str(tmp_path / f"{tag}_reads.tsv.gz"),
                "--input", str(tmp_path / "R1.fastq"), str(tmp_path / "R2.fastq"), "--tenx", "--map", str(tmp_path / "wl.txt"),
                "--report-output", str(tmp_path / f"{tag}_cells.tsv")] + extra
        for cmd in (bulk, tenx):
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
            assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    run("one", [])
    run("two", ["--devices", "0,0"], dict(env, NB_RANGE_SHARDING="0"))
    run("rng", ["--devices", "0,0"])
    for name in ("bulk.tsv", "reads.tsv.gz", "cells.tsv"):
        a, b, c = (open(tmp_path / f"{tag}_{name}", "rb").read() for tag in ("one", "two", "rng"))
        assert a == b and len(a) > 500, name
        assert (gzip.decompress(a) == gzip.decompress(c)) if name.endswith(".gz") else a == c, name
    assert len(gzip.open(tmp_path / "two_reads.tsv.gz", "rt").read().splitlines()) > 5000
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".part")]

"""Host-side pieces that need no GPU: path helper, library JSON, packing, FASTQ ingest, barcodes, CLI surface."""
import gzip
import json
import os

import numpy as np
import pytest

from nimble_amd import engine, library, synth, utils
from tests import barcodes_twin as barcodes
from nimble_amd.__main__ import build_parser

HERE = os.path.dirname(os.path.abspath(__file__))


def test_append_path_string_golden():
    for c in json.load(open(os.path.join(HERE, "golden", "append_path_string.json"))):
        assert utils.append_path_string(c["path"], c["suffix"]) == c["expected"], c


def test_generate_csv_matches_reference_text(tmp_path):
    for c in json.load(open(os.path.join(HERE, "golden", "generate_csv.json"))):
        p, o = tmp_path / c["csv_name"], tmp_path / "lib.json"
        p.write_text(c["csv"])
        library.generate(str(p), None, str(o))
        assert o.read_text() == c["json"], c["name"]


def test_generate_fasta_and_collate(tmp_path):
    fa = tmp_path / "my_hla_lib.fasta"
    fa.write_text(">A*01 some description\nACGTAC\nGTAC\n>B*02\nTTTTGGGG\n")
    o = tmp_path / "lib.json"
    library.generate(str(fa), None, str(o))
    cfg, data = json.load(open(o))
    assert list(cfg)[:3] == ["score_threshold", "score_filter", "score_percent"] and cfg["data_type"] == "RNA"
    assert data["headers"] == ["reference_genome", "sequence_name", "nt_length", "sequence"]
    assert data["columns"] == [["my hla lib"] * 2, ["A*01", "B*02"], ["10", "8"], ["ACGTACGTAC", "TTTTGGGG"]]
    meta = tmp_path / "meta.csv"
    meta.write_text("name,lineage\nB*02,B\nA*01,A\n")
    library.generate(str(fa), str(meta), str(o))
    cfg, data = json.load(open(o))
    assert data["headers"][-1] == "lineage" and data["columns"][1] == ["B*02", "A*01"]
    assert data["columns"][3] == ["TTTTGGGG", "ACGTACGTAC"] and data["columns"][2] == ["8", "10"]
    lib = library.Library(str(o))
    assert lib.features == ["B*02", "A*01"] and lib.config["max_hits_to_report"] == 10
    doc = json.load(open(o)); doc[0]["group_on"] = "lineage"; json.dump(doc, open(o, "w"))
    lib = library.Library(str(o))
    assert lib.features == ["B", "A"] and lib.feature_list([1, 0, 1]) == ["A", "B"]
    doc[0]["group_on"] = "nope"; json.dump(doc, open(o, "w"))
    with pytest.raises(ValueError):
        library.Library(str(o))


def test_pack_reads_matches_numpy_twin(built):
    codes = synth.synth_reads([np.random.default_rng(0).integers(0, 4, 400, dtype=np.uint8)], 300, 150, seed=9, n_rate=0.02)
    reads = [synth.codes_to_str(c) for c in codes]
    words, lens = engine.pack_reads(reads, 150)
    assert (lens == 150).all() and (words == synth.pack_codes(codes)).all()
    w2, l2 = engine.pack_reads(["ACGTN", "", "acgt"], 70)
    assert list(l2) == [5, 0, 4] and w2.shape == (3, 3 + 2)
    assert w2[0, 0] == (0 | 1 << 2 | 2 << 4 | 3 << 6) and w2[0, 3] == 1 << 4 and w2[2, 0] == w2[0, 0] & 0xFF
    with pytest.raises(engine._lib.NimbleError):
        engine.pack_reads(["ACGT" * 10], 20)


def test_fastq_ingest_plain_and_gzip(built, tmp_path):
    from nimble_amd.align import FastqReader
    recs = [("r1", "ACGTNACGT", "IIIIIIIII"), ("r2", "TTTT", "!!!!"), ("r3", "", "")]
    text = "".join(f"@{n}\n{s}\n+\n{q}\n" for n, s, q in recs)
    p, pz = tmp_path / "a.fastq", tmp_path / "a.fastq.gz"
    p.write_text(text)
    with gzip.open(pz, "wt") as f:
        f.write(text)
    want_w, want_l = engine.pack_reads([s for _, s, _ in recs], 20)
    for path in (p, pz):
        r = FastqReader(str(path))
        b = r.next_batch(2, 20)
        assert b["n"] == 2 and (b["words"] == want_w[:2]).all() and list(b["lens"]) == [9, 4]
        b = r.next_batch(10, 20)
        assert b["n"] == 1 and b["lens"][0] == 0
        assert r.next_batch(10, 20)["n"] == 0
        r.close()
    r = FastqReader(str(p))
    b = r.next_batch(1, barcodes_only=True, cb_len=4, umi_len=3)
    assert b["cb"][0].tobytes() == b"ACGT" and b["umi"][0].tobytes() == b"NAC" and b["cbq"][0].tobytes() == b"IIII"
    r.close()
    bad = tmp_path / "bad.fastq"
    bad.write_text("@x\nACGT\nIIII\n")
    r = FastqReader(str(bad))
    with pytest.raises(engine._lib.NimbleError):
        r.next_batch(5, 20)


def _drain(reader, batch, cap_len):
    reads, lens = [], []
    while True:
        b = reader.next_batch(batch, cap_len)
        if b["n"] == 0:
            return reads, lens
        w, l = b["words"], b["lens"]
        assert w.shape == (b["n"], engine.packed_stride_words(b["max_len"])) and int(l.max()) <= b["max_len"]
        for i in range(b["n"]):
            reads.append(engine.unpack_read(w[i], int(l[i]), b["max_len"]))
        lens.extend(int(x) for x in l)


@pytest.mark.parametrize("newline,blank,trailing", [("\n", False, True), ("\r\n", False, True), ("\n", True, True), ("\n", False, False)])
def test_fastq_ingest_framing_variants(built, tmp_path, newline, blank, trailing):
    """Parallel strict-framing indexer and the serial walker (blank lines between records, CRLF, no final newline) must give
    the same records; batches are packed at the layout of their own longest read and end early before a longer one."""
    from nimble_amd.align import FastqReader
    rng = np.random.default_rng(7)
    n = 100_000
    lens = rng.integers(0, 61, n)
    lens[70_000] = 97                      # one long read deep in the file: forces an early batch end + a wider layout
    seqs = ["".join("ACGTNacgt"[c] for c in rng.integers(0, 9, L)) for L in lens]
    recs = [f"@r{i} x{newline}{s}{newline}+{newline}{'I' * len(s)}{newline}" for i, s in enumerate(seqs)]
    if blank:
        recs[1000] = newline + recs[1000]
        recs[50_000] = newline + newline + recs[50_000]
    text = "".join(recs)
    if not trailing:
        text = text[:-len(newline)]
    elif blank:
        text += newline
    p = tmp_path / "v.fastq"
    p.write_bytes(text.encode())
    for threads, batch in ((1, 1 << 20), (4, 40_000), (3, 7)):
        if batch == 7 and n > 20_000:
            continue
        r = FastqReader(str(p), threads)
        got, got_l = _drain(r, batch, 128)
        r.close()
        assert got_l == [int(x) for x in lens]
        assert got == [s.upper() for s in seqs]
    r = FastqReader(str(p), 2)
    with pytest.raises(engine._lib.NimbleError):
        _drain(r, 1 << 20, 96)             # the 97-base read exceeds the cap
    r.close()


def test_fastq_ingest_errors(built, tmp_path):
    from nimble_amd.align import FastqReader
    cases = {"no_plus": "@a\nACGT\nIIII\n@b\nAC\n+\nII\n", "truncated": "@a\nACGT\n+\nIIII\n@b\nAC\n", "no_at": "a\nACGT\n+\nIIII\n"}
    for name, text in cases.items():
        p = tmp_path / f"{name}.fastq"
        p.write_text(text)
        r = FastqReader(str(p), 2)
        with pytest.raises(engine._lib.NimbleError):
            _drain(r, 100, 50)
        r.close()
    z = tmp_path / "corrupt.fastq.gz"
    good = tmp_path / "good.fastq.gz"
    with gzip.open(good, "wt") as f:
        f.write("".join(f"@r{i}\n{'ACGT' * 20}\n+\n{'I' * 80}\n" for i in range(20000)))
    raw = bytearray(good.read_bytes())
    raw[len(raw) // 2] ^= 0xFF
    z.write_bytes(bytes(raw[: len(raw) - 40]))
    r = FastqReader(str(z), 2)
    with pytest.raises(engine._lib.NimbleError):
        _drain(r, 1 << 20, 100)
    r.close()


def test_barcode_correction_rule():
    wl = {"AAAA", "AAAC", "CAAT", "GGGG"}
    c = barcodes.BarcodeCorrector(wl)
    assert c.correct("AAAA", "IIII") == "AAAA"                       # exact
    assert c.correct("GGGT", "IIII") == "GGGG"                       # single candidate
    assert c.correct("TTTT", "IIII") is None                         # none -> dropped
    # AAAT: candidates AAAA/AAAC (position 3) and CAAT (position 0); lowest quality at the differing base wins
    assert barcodes.BarcodeCorrector(wl).correct("AAAT", "!III") == "CAAT"
    assert barcodes.BarcodeCorrector(wl).correct("AAAT", "III!") == "AAAA"   # tie between AAAA/AAAC -> smallest
    c2 = barcodes.BarcodeCorrector(wl)
    assert c2.correct("AAAT", "!III") == "CAAT" and c2.correct("AAAT", "III!") == "CAAT"   # first sighting is cached
    assert c2.stats == {"cb_perfect_match": 0, "cb_corrected": 2, "cb_no_correction": 0}


def test_cli_flag_surface():
    """Flags of the reference's align/report/generate parsers (__main__.py:389-418), same names, types and defaults."""
    p = build_parser()
    a = p.parse_args(["align", "--reference", "a.json,b.json", "--output", "o.tsv", "--input", "r1.fq", "r2.fq"])
    assert (a.num_cores, a.strand_filter, a.trim, a.tmpdir) == (1, "unstranded", "", None) and a.input == ["r1.fq", "r2.fq"]
    a = p.parse_args(["align", "--reference", "a", "--output", "o", "--input", "x", "-c", "8", "--strand_filter", "fiveprime", "--trim", "50:0.9"])
    assert a.num_cores == 8 and a.strand_filter == "fiveprime"
    r = p.parse_args(["report", "-i", "in", "-o", "out"])
    assert r.threshold == 0.05 and r.disable_thresholding is False and r.summarize is None
    r = p.parse_args(["report", "--input", "in", "--output", "out", "-t", "0.2", "--disable_thresholding", "-s", "a,b"])
    assert r.threshold == 0.2 and r.disable_thresholding and r.summarize == "a,b"
    g = p.parse_args(["generate", "--file", "x.fasta", "--opt-file", "m.csv", "--output_path", "lib.json"])
    assert g.opt_file == "m.csv"
    assert utils.parse_trim("50:0.9,40:0.5", 2) == [(50, 0.9), (40, 0.5)] and utils.parse_trim("", 3) is None
    with pytest.raises(ValueError):
        utils.parse_trim("50:0.9", 2)


def test_align_without_gpu_returns_nonzero(built, tmp_path, capsys):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nimble_amd.align import align
    fa = tmp_path / "l.fasta"; fa.write_text(">s\n" + "ACGT" * 20 + "\n")
    library.generate(str(fa), None, str(tmp_path / "l.json"))
    fq = tmp_path / "r.fastq"; fq.write_text("@r\n" + "ACGT" * 10 + "\n+\n" + "I" * 40 + "\n")
    assert align(str(tmp_path / "l.json"), str(tmp_path / "o.tsv"), [str(fq)], 1, "unstranded", "", None) == 1
    assert "no CPU alignment path" in capsys.readouterr().err
    assert align(str(tmp_path / "l.json"), str(tmp_path / "o.tsv"), ["x.bam"], 1, "unstranded", "", None) == 1


def test_unsupported_options_are_announced(built, tmp_path, capsys):
    """--trim, trim_* / filters / data_type of a library are read by the reference's aligner (nimble/__main__.py:191-192,
    types.py:20-25) and not acted on here: align() must say so on stderr instead of accepting them silently. (The warnings
    are printed before the device check, so this runs without a GPU.)"""
    import json
    from nimble_amd.align import align
    fa = tmp_path / "l.fasta"; fa.write_text(">s\n" + "ACGT" * 20 + "\n")
    library.generate(str(fa), None, str(tmp_path / "l.json"))
    fq = tmp_path / "r.fastq"; fq.write_text("@r\n" + "ACGT" * 10 + "\n+\n" + "I" * 40 + "\n")
    align(str(tmp_path / "l.json"), str(tmp_path / "o.tsv"), [str(fq)], 1, "unstranded", "", None)
    assert "not honoured" not in capsys.readouterr().err            # defaults: nothing to announce
    align(str(tmp_path / "l.json"), str(tmp_path / "o.tsv"), [str(fq)], 1, "unstranded", "40:0.8", None)
    err = capsys.readouterr().err
    assert "not honoured" in err and "--trim '40:0.8'" in err and "NOT trimmed" in err
    doc = json.load(open(tmp_path / "l.json"))
    doc[0].update(trim_target_length=30, filters=["x"], data_type="DNA")
    json.dump(doc, open(tmp_path / "l2.json", "w"))
    align(str(tmp_path / "l2.json"), str(tmp_path / "o.tsv"), [str(fq)], 1, "unstranded", "", None)
    err = capsys.readouterr().err
    assert err.count("not honoured") == 3 and "trim_target_length = 30" in err and "filters = ['x']" in err and "data_type = 'DNA'" in err


def test_native_whitelist_matches_python_rule(built, tmp_path):
    """C++ corrector (nb_whitelist_correct) == barcodes.BarcodeCorrector on random barcodes incl. cache order effects."""
    from nimble_amd.align import NativeWhitelist
    rng = np.random.default_rng(11)
    wl = sorted({"".join("ACGT"[c] for c in rng.integers(0, 4, 8)) for _ in range(3000)})
    p = tmp_path / "wl.txt.gz"
    with gzip.open(p, "wt") as f:
        f.write("\n".join(wl) + "\n\n")
    raws, quals = [], []
    for _ in range(50000):             # enough for the threaded exact-match pass of the native corrector
        c = list(wl[rng.integers(0, len(wl))])
        for _ in range(int(rng.integers(0, 3))):
            c[rng.integers(0, 8)] = "ACGTN"[rng.integers(0, 5)]
        raws.append("".join(c)); quals.append("".join(chr(33 + int(q)) for q in rng.integers(2, 41, 8)))
    raws += raws[:500]; quals += ["".join(chr(33 + int(q)) for q in rng.integers(2, 41, 8)) for _ in range(500)]   # repeats: cache decides
    py = barcodes.BarcodeCorrector(barcodes.load_whitelist(str(p)))
    want = [py.correct(r, q) for r, q in zip(raws, quals)]
    nat = NativeWhitelist(str(p), threads=3)
    cb = np.array([list(r) for r in raws], dtype="S1"); cq = np.array([list(q) for q in quals], dtype="S1")
    out, ok = nat.correct(cb, cq)
    got = [o.tobytes().decode() if k else None for o, k in zip(out, ok)]
    assert got == want and nat.stats == py.stats and 0 < sum(w is None for w in want) < len(want)


def test_class_features_native_matches_feature_list(built):
    """nb_class_features == Library.feature_list + the max_hits / multi_hits rule (docs/SPEC.md §4-5), grouped names included."""
    import ctypes as ct
    rng = np.random.default_rng(0)
    n_seqs = 130
    feats = [f"L{i % 7}" if i % 3 else f"Z{i}" for i in range(n_seqs)]
    names = sorted(set(feats))
    idx = {f: i for i, f in enumerate(names)}
    seq_feature = np.array([idx[f] for f in feats], dtype=np.int32)
    enc = [f.encode() for f in names]
    arr = (ct.c_char_p * len(enc))(*enc)
    rows = np.zeros((400, 3), dtype=np.uint64)
    for r in range(400):
        for m in rng.choice(n_seqs, int(rng.integers(1, 12)), replace=False):
            rows[r, m >> 6] |= np.uint64(1) << np.uint64(m & 63)
    for max_hits, multi in ((4, 0), (10, 2)):
        a, b, reasons = engine.class_features(rows, seq_feature, arr, len(names), max_hits, multi, max(len(e) for e in enc))
        for r in range(400):
            seen = []
            for m in engine.bits_to_members(rows[r]):
                if feats[m] not in seen:
                    seen.append(feats[m])
            want = 10 if len(seen) > max_hits else (11 if multi and len(seen) > multi else 0)
            assert a[r] == ",".join(seen) and b[r] == ",".join(sorted(seen)) and reasons[r] == want


def _gzip_members(data):
    """Split a BGZF byte string into its gzip members (BSIZE in the BC extra field)."""
    out, o = [], 0
    while o < len(data):
        bsize = int.from_bytes(data[o + 16:o + 18], "little") + 1
        out.append(data[o:o + bsize])
        o += bsize
    return out


def test_bam_reader_single_and_paired(built, tmp_path):
    from nimble_amd.align import BamReader
    from tests.helpers import write_bam
    comp = str.maketrans("ACGTN", "TGCAN")
    s1, s2, s3 = "ACGTTGCANNACGTAGGCTA" * 3, "TTGACCATGACCAGTNACGT" * 2, "GGGGACGTACGTAAAC"
    recs = [
        {"name": "a", "flag": 0, "seq": s1, "tags": {"XX": 7, "CB": "ACGTACGTACGTACGT-1", "BQ": [1, 2, 3], "UB": "AACCGGTTAACC"}},
        {"name": "sec", "flag": 0x100, "seq": "ACGT", "tags": {"CB": "X"}},                       # secondary: skipped
        {"name": "b", "flag": 0x10, "seq": s2.translate(comp)[::-1], "tags": {"UB": "TTTT"}},    # stored reverse-complemented, no CB
        {"name": "c", "flag": 0, "seq": s3},
    ]
    p = tmp_path / "se.bam"
    write_bam(p, recs, block=97)                                                                # many small BGZF members
    r = BamReader(str(p))
    b = r.next_batch(2, 100)
    assert b["n"] == 2 and not b["paired"]
    assert b["max_len"] == 60                                                                  # packed at the batch's longest read
    want, wl = engine.pack_reads([s1, s2], b["max_len"])
    assert (b["words"] == want).all() and list(b["lens"]) == list(wl)
    assert b["cb"][0].tobytes().rstrip(b"\x00") == b"ACGTACGTACGTACGT-1" and b["umi"][0].tobytes().rstrip(b"\x00") == b"AACCGGTTAACC"
    assert b["cb"][1].tobytes().rstrip(b"\x00") == b"" and b["umi"][1].tobytes().rstrip(b"\x00") == b"TTTT"
    b = r.next_batch(5, 100)
    assert b["n"] == 1 and b["max_len"] == 16 and (b["words"] == engine.pack_reads([s3], 16)[0]).all()
    assert r.next_batch(5, 100)["n"] == 0
    r.close()
    # pairs: flags 77 / 141 as written by the reference's fastq-to-bam (fastq_barcode_processor.py:185,199)
    pairs = [{"name": "q1", "flag": 77, "seq": s1, "tags": {"CB": "AAAA", "UB": "CC"}}, {"name": "q1", "flag": 141, "seq": s2, "tags": {"CB": "AAAA", "UB": "CC"}},
             {"name": "q2", "flag": 141, "seq": s3, "tags": {"CB": "GGGG", "UB": "TT"}}, {"name": "q2", "flag": 77, "seq": s2, "tags": {"CB": "GGGG", "UB": "TT"}}]
    p2 = tmp_path / "pe.bam"
    write_bam(p2, pairs)
    r = BamReader(str(p2))
    b = r.next_batch(10, 100)
    assert b["n"] == 2 and b["paired"]
    assert (b["max_len"], b["mate_max_len"]) == (60, 40)
    assert (b["words"] == engine.pack_reads([s1, s2], 60)[0]).all() and (b["mate_words"] == engine.pack_reads([s2, s3], 40)[0]).all()
    assert b["cb"][1].tobytes().rstrip(b"\x00") == b"GGGG"
    r.close()
    # mates that are not adjacent (a coordinate-sorted file): paired through the waiting map; orphans come out at the end with
    # an empty partner; two records claiming the same end of a pair are an error
    rng2 = np.random.default_rng(12)
    tmpl = [("t%03d" % i, "".join("ACGT"[c] for c in rng2.integers(0, 4, int(rng2.integers(20, 60)))),
             "".join("ACGT"[c] for c in rng2.integers(0, 4, int(rng2.integers(20, 60))))) for i in range(300)]
    recs2 = []
    for name, a, c in tmpl:
        recs2.append({"name": name, "flag": 77, "seq": a, "tags": {"CB": "CB" + name, "UB": "UA"}})
        recs2.append({"name": name, "flag": 141, "seq": c, "tags": {"CB": "CB" + name, "UB": "UA"}})
    order = rng2.permutation(len(recs2))
    shuffled = [recs2[i] for i in order] + [{"name": "orphan1", "flag": 77, "seq": "ACGTACGTAC", "tags": {"CB": "O1"}},
                                              {"name": "orphan2", "flag": 141, "seq": "TTTTGGGGCC", "tags": {"CB": "O2"}}]
    shuffled.insert(5, {"name": "supp", "flag": 0x801, "seq": "ACGT"})
    p3 = tmp_path / "shuffled.bam"
    write_bam(p3, shuffled, block=500)
    for batch in (1000, 37):
        r = BamReader(str(p3), 2)
        got = {}
        while True:
            b = r.next_batch(batch, 100)
            if b["n"] == 0:
                break
            assert b["paired"]
            for i in range(b["n"]):
                cb = b["cb"][i].tobytes().rstrip(b"\x00").decode()
                got[cb] = (engine.unpack_read(b["words"][i], int(b["lens"][i]), b["max_len"]),
                           engine.unpack_read(b["mate_words"][i], int(b["mate_lens"][i]), b["mate_max_len"]))
        r.close()
        want = {"CB" + name: (a, c) for name, a, c in tmpl}
        want["O1"] = ("ACGTACGTAC", ""); want["O2"] = ("", "TTTTGGGGCC")
        assert got == want
    p4 = tmp_path / "dup.bam"
    write_bam(p4, [pairs[0], pairs[3], dict(pairs[0])])
    r = BamReader(str(p4))
    with pytest.raises(engine._lib.NimbleError):
        while r.next_batch(10, 100)["n"]:
            pass
    r.close()
    # a larger file: several refills, worker threads, sub-batches, a plain-gzip (non-BGZF) copy, a corrupted block
    rng = np.random.default_rng(5)
    seqs = ["".join("ACGTN"[c] for c in rng.integers(0, 5, int(rng.integers(20, 91)))) for _ in range(70_000)]
    big = tmp_path / "big.bam"
    write_bam(big, [{"name": f"r{i}", "flag": 0, "seq": s, "tags": {"CB": "ACGT", "UB": "TT"}} for i, s in enumerate(seqs)], block=60000)
    import zlib
    raw = b"".join(zlib.decompress(m, 31) for m in _gzip_members(big.read_bytes()))
    plain = tmp_path / "plain.bam"
    plain.write_bytes(gzip.compress(raw, 1))
    for path, threads in ((big, 1), (big, 4), (plain, 2)):
        r = BamReader(str(path), threads)
        got = []
        while True:
            b = r.next_batch(50_000, 100)
            if b["n"] == 0:
                break
            got.extend(engine.unpack_read(b["words"][i], int(b["lens"][i]), b["max_len"]) for i in range(0, b["n"], 97))
        r.close()
        assert got == seqs[::97][:len(got)] and len(got) == len(seqs[::97]) or got == [s for j in range(0, len(seqs), 50_000) for s in seqs[j:j + 50_000][::97]]
    data = bytearray(big.read_bytes())
    data[len(data) // 2] ^= 0x55
    (tmp_path / "corrupt.bam").write_bytes(bytes(data))
    with pytest.raises(engine._lib.NimbleError):     # (the CRC of every block is checked; a small file fails at open already)
        r = BamReader(str(tmp_path / "corrupt.bam"), 2)
        while r.next_batch(50_000, 100)["n"]:
            pass
    (tmp_path / "not.bam").write_bytes(gzip.compress(b"hello world, not a bam"))
    with pytest.raises(engine._lib.NimbleError):
        BamReader(str(tmp_path / "not.bam"))


def test_fused_row_accumulator_equals_report_on_tsv(built, tmp_path):
    """merge_rows + nb_report_rows (the fused per-cell counting) == report() on the per-read TSV, incl. N-containing UMIs,
    several batches and duplicate rows."""
    from nimble_amd import fused, report as rep
    rng = np.random.default_rng(17)
    feats = ["A*01", "A*01,A*02", "B*07", "A*02,B*07,C*04", "C*04"]
    cells = ["".join("ACGT"[c] for c in rng.integers(0, 4, 16)) for _ in range(12)]
    umis = ["".join("ACGTN"[c] for c in rng.integers(0, 5, 12)) for _ in range(400)]
    n = 6000
    ci, ui, fi = rng.integers(0, 12, n), rng.integers(0, 400, n), rng.integers(0, 5, n)
    mates = rng.integers(1, 3, n)
    tsv = tmp_path / "in.tsv"
    with open(tsv, "w") as f:
        f.write("nimble_features\tnimble_score\tr1_CB\tr1_UB\n")
        for i in range(n):
            f.write(f"{feats[fi[i]]}\t{mates[i]}\t{cells[ci[i]]}\t{umis[ui[i]]}\n")
    rep.report(str(tsv), str(tmp_path / "want.tsv"), threshold=0.2, engine="python")
    acc = fused.RowAccumulator()
    cb_arr = np.array([list(cells[c]) for c in ci], dtype="S1"); umi_arr = np.array([list(umis[u]) for u in ui], dtype="S1")
    for lo in range(0, n, 2500):
        sl = slice(lo, lo + 2500)
        acc.add(acc.cb_ids_of(cb_arr[sl]), acc.umi_keys_of(umi_arr[sl]), fi[sl], mates[sl])
    merged = acc.finish(None, feats, str(tmp_path / "got.tsv"), threshold=0.2)
    assert merged < n and open(tmp_path / "got.tsv").read() == open(tmp_path / "want.tsv").read()
    assert fused.decode_umi(fused.encode_umis(umi_arr[:1])[0], 12) == umis[ui[0]]


def test_top_up_never_touches_the_batch_in_flight(built, tmp_path):
    """A reader may end a batch early (before a longer read); align() then tops the batch up with further next_batch() calls
    from the helper thread while the PREVIOUS batch is still being uploaded. Those extra calls must not land in the buffer
    set of that previous batch (they did: the turn advanced once per call) — they read into private scratch buffers."""
    from nimble_amd.align import FastqReader, _top_up
    rng = np.random.default_rng(11)
    n = 260_000
    lens = rng.integers(20, 50, n)
    lens[140_000] = 120                    # late long read: the second outer batch ends early in front of it
    seqs = ["".join("ACGT"[c] for c in rng.integers(0, 4, L)) for L in lens]
    p = tmp_path / "t.fastq"
    p.write_text("".join(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n" for i, s in enumerate(seqs)))
    r = FastqReader(str(p), 2)
    a = r.next_batch(100_000, 128)
    assert a["n"] == 100_000
    a_words, a_lens = a["words"].copy(), a["lens"].copy()
    b = r.next_batch(100_000, 128)
    assert 0 < b["n"] < 100_000            # ended early: the 120-base read needs a wider layout
    full = _top_up(r, b, 100_000)
    assert full["n"] == 100_000
    assert (a["words"] == a_words).all() and (a["lens"] == a_lens).all()      # batch A (still "on its way to the GPU") is untouched
    got = [engine.unpack_read(full["words"][i], int(full["lens"][i]), full["max_len"]) for i in range(0, 100_000, 997)]
    assert got == [seqs[100_000 + i] for i in range(0, 100_000, 997)]
    assert engine.unpack_read(full["words"][40_000], int(full["lens"][40_000]), full["max_len"]) == seqs[140_000]
    c = r.next_batch(100_000, 128)         # the stream continues where the top-up stopped
    assert c["n"] == 60_000 and engine.unpack_read(c["words"][0], int(c["lens"][0]), c["max_len"]) == seqs[200_000]
    r.close()


def test_fused_merge_rows_wide_ids():
    """merge_rows with ids beyond the packed 21 + 42-bit pair key (e.g. > 2^21 raw cell barcodes without a whitelist):
    dense ranks, and the three-key sort when even the ranks do not fit — same merged rows as a dictionary."""
    import torch
    from nimble_amd import fused
    rng = np.random.default_rng(1)
    n = 20000

    def ref(cb, umi, f, s):
        d = {}
        for key in zip(cb.tolist(), umi.tolist(), f.tolist(), s.tolist()):
            d[key[:3]] = d.get(key[:3], 0) + key[3]
        return d
    for cbmax, umimax in ((100, 1 << 24), (1 << 30, 1 << 24), (1 << 40, 1 << 50)):
        cb = torch.from_numpy(rng.integers(0, cbmax, 300)[rng.integers(0, 300, n)])
        umi = torch.from_numpy(rng.integers(0, umimax, 500)[rng.integers(0, 500, n)])
        f, s = torch.from_numpy(rng.integers(0, 5, n)), torch.from_numpy(rng.integers(1, 3, n))
        out = fused.merge_rows(cb, umi, f, s)
        assert {k[:3]: k[3] for k in zip(*[x.tolist() for x in out])} == ref(cb, umi, f, s)
    cb, umi = torch.from_numpy(rng.integers(0, 50, n)), torch.from_numpy(rng.integers(0, 60, n))
    f, s = torch.from_numpy(rng.integers(0, 5, n)), torch.from_numpy(rng.integers(1, 3, n))
    out = fused._merge_wide(cb, umi, f, s)
    assert {k[:3]: k[3] for k in zip(*[x.tolist() for x in out])} == ref(cb, umi, f, s)


def test_bench_reference_binary_hook(tmp_path, monkeypatch, built):
    """bench.py times a nimble-aligner binary ONLY when NIMBLE_ALIGNER_BIN names one, with the argv the reference
    builds (/root/reference/nimble/__main__.py:177-189). A stand-in script records its argv; nothing is ever downloaded."""
    import importlib.util
    import stat
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    _, seqs, _ = synth.synth_library(2, 3, locus_len=300, seed=5)
    strs = [synth.codes_to_str(s) for s in seqs]
    codes = synth.synth_reads(seqs, 50, 100, seed=1)
    assert bench.time_reference_binary(None, codes, strs, 4) is None
    assert bench.time_reference_binary(str(tmp_path / "missing"), codes, strs, 4) is None
    fake = tmp_path / "aligner"
    fake.write_text(f"#!/bin/sh\necho \"$@\" > {tmp_path}/argv.txt\nwc -l < \"$2\" > {tmp_path}/lines.txt\nexit 0\n")
    fake.chmod(fake.stat().st_mode | stat.S_IXUSR)
    r = bench.time_reference_binary(str(fake), codes, strs, 4)
    assert r["kind"] == "reference" and r["cores"] == 4 and r["value"] > 0
    argv = open(tmp_path / "argv.txt").read().split()
    assert argv[0] == "--input" and argv[1].endswith("reads.fastq") and argv[2:6] == ["-c", "4", "--strand_filter", "unstranded"]
    assert argv[6] == "-r" and argv[7].endswith("lib.json") and argv[8] == "-o" and argv[9].endswith("out.tsv")
    assert int(open(tmp_path / "lines.txt").read()) == 200            # 50 records x 4 lines
    fake.write_text("#!/bin/sh\nexit 3\n")
    assert bench.time_reference_binary(str(fake), codes, strs, 4) is None

"""BASELINE configs[2] at size: 10x pairs (R1 = 16-bp cell barcode + 12-bp UMI, R2 = 91-bp cDNA) vs an MHC+KIR-like library of
1 520 sequences (40 loci x 38 alleles), whitelist of 100 000 barcodes with 1 % single-base errors in the reads, 10 M pairs
(NB_C3_PAIRS to change) — through the real path: FASTQ files -> `align --tenx --map` -> per-read TSV, and the fused per-cell
counts (`--report-output`, GPU merge of (cell, UMI, feature) rows + the golden-pinned reducer) against `report` run on that TSV.

  * per-read calls: the rows of the first 20 000 pairs are rebuilt from the oracle's calls + the Python twin of the whitelist
    rule and compared line by line; the row count of the whole table equals the kernel's CALLED counter minus dropped barcodes;
  * per-cell counts: fused output == `report` on the per-read TSV, byte for byte, on ALL pairs;
  * device side: the four size-independent properties + an oracle-checked sample on the packed 91-bp batch (kernel
    nb_align_se_kernel<2, 1, false>);
  * `bench.py --config c3 --per-cell` runs and reports a consistent per_cell object.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from nimble_amd import engine, library, synth
from nimble_amd.__main__ import main as cli
from tests.helpers import check_fullsize_properties

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L2, CB, UMI = 91, 16, 12
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _c3_library(tmp_path):
    names, seqs, lineage = synth.synth_library(40, 38)
    strs = [synth.codes_to_str(s) for s in seqs]
    assert len(strs) == 1520
    csv_path = tmp_path / "mhc_kir.csv"
    with open(csv_path, "w") as f:
        f.write("name,sequence,lineage\n")
        for n, s, l in zip(names, strs, lineage):
            f.write(f"{n},{s},{l}\n")
    out = tmp_path / "mhc_kir.json"
    assert cli(["generate", "--file", str(csv_path), "--output_path", str(out)]) == 0
    return out, names, seqs, strs


def _fastq_block(first_id, tag, seq_ascii):
    """[m, width] uint8 bases -> FASTQ text (uint8) with names q<10 digits>/<tag>, constant quality 'I'."""
    m, w = seq_ascii.shape
    ids = first_id + np.arange(m, dtype=np.int64)
    name = np.empty((m, 14), dtype=np.uint8)
    name[:, 0], name[:, 1] = ord("@"), ord("q")
    for d in range(10):
        name[:, 11 - d] = ord("0") + (ids // 10 ** d) % 10
    name[:, 12], name[:, 13] = ord("/"), ord(tag)
    rec = np.empty((m, 14 + 1 + w + 3 + w + 1), dtype=np.uint8)
    rec[:, :14] = name
    rec[:, 14] = 10
    rec[:, 15:15 + w] = seq_ascii
    rec[:, 15 + w:18 + w] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 18 + w:18 + 2 * w] = ord("I")
    rec[:, 18 + 2 * w] = 10
    return rec


def test_c3_pairs_per_read_and_per_cell(built, tmp_path):
    import torch
    from oracle import oracle
    from tests import barcodes_twin
    dev = torch.device("cuda", 0)
    n = int(os.environ.get("NB_C3_PAIRS", 10_000_000))
    lib_path, names, seqs, strs = _c3_library(tmp_path)
    rng = np.random.default_rng(33)
    # whitelist: 100 000 distinct 16-mers; 10 000 of them are cells
    wl_codes = np.unique(rng.integers(0, 4, size=(101_000, CB), dtype=np.uint8), axis=0)[:100_000]
    rng.shuffle(wl_codes)
    wl = [bytes(ACGT[c]).decode() for c in wl_codes]
    (tmp_path / "wl.txt").write_text("\n".join(wl) + "\n")
    n_umis = max(1, n // 4)                                   # mean 4 reads per UMI
    umi_codes = rng.integers(0, 4, size=(n_umis, UMI), dtype=np.uint8)
    umi_cell = rng.integers(0, 10_000, size=n_umis)           # a UMI belongs to one cell
    f1, f2 = open(tmp_path / "R1.fastq", "wb"), open(tmp_path / "R2.fastq", "wb")
    chunk, first_r2, first_cb, first_umi = 1 << 20, None, None, None
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        _, codes = synth.synth_reads_packed_torch(seqs, m, L2, 3 + lo // chunk, dev, return_codes_of_first=m)
        u = rng.integers(0, n_umis, size=m)
        cb = wl_codes[umi_cell[u]].copy()
        bad = np.nonzero(rng.random(m) < 0.01)[0]             # 1 % of the reads carry a single-base barcode error
        pos = rng.integers(0, CB, size=bad.size)
        cb[bad, pos] = (cb[bad, pos] + rng.integers(1, 4, size=bad.size, dtype=np.uint8)) & 3
        r1 = np.concatenate([ACGT[cb], ACGT[umi_codes[u]]], axis=1)
        r2 = synth.BASES[codes]
        _fastq_block(lo, "1", r1).tofile(f1)
        _fastq_block(lo, "2", r2).tofile(f2)
        if lo == 0:
            k = min(20_000, m)
            first_r2 = [bytes(r).decode() for r in r2[:k]]
            first_cb = [bytes(r).decode() for r in ACGT[cb[:k]]]
            first_umi = [bytes(r).decode() for r in ACGT[umi_codes[u[:k]]]]
    f1.close(); f2.close()
    out, fused = tmp_path / "per_read.tsv", tmp_path / "fused.tsv"
    assert cli(["align", "--reference", str(lib_path), "--output", str(out), "--input", str(tmp_path / "R1.fastq"), str(tmp_path / "R2.fastq"),
                "--tenx", "--map", str(tmp_path / "wl.txt"), "-c", "16", "--report-output", str(fused)]) == 0
    # ---- per-read rows of the first 20 000 pairs vs oracle + whitelist twin ----
    lib = library.Library(str(lib_path))
    cfg = engine.make_config(k=lib.k)
    ora = oracle.OracleIndex(lib.sequences, lib.k).align_strs(cfg, first_r2, n_threads=min(16, os.cpu_count() or 1))
    corr = barcodes_twin.BarcodeCorrector(set(wl))
    want = []
    for i in range(len(first_r2)):
        c = corr.correct(first_cb[i], "I" * CB)
        if ora["reason"][i] == 0 and c is not None:
            feats = lib.feature_list(engine.bits_to_members(ora["set_bits"][i]))
            if len(feats) <= 10:
                want.append("\t".join([",".join(feats), "1", c, first_umi[i]] + [str(x) for x in ora["cov"][i]]))
    n_rows = 0
    with open(out) as f:
        header = f.readline().rstrip("\n").split("\t")
        assert header[:4] == ["nimble_features", "nimble_score", "r1_CB", "r1_UB"]
        head = []
        for line in f:
            if len(head) < len(want):
                head.append(line.rstrip("\n"))
            n_rows += 1
    assert head == want and len(want) > 10_000
    assert 0.55 * n < n_rows < 0.85 * n
    # ---- per-cell counts: fused (no TSV round trip) == report on the per-read TSV, all pairs ----
    assert cli(["report", "-i", str(out), "-o", str(tmp_path / "counts.tsv")]) == 0
    got, exp = open(fused).read(), open(tmp_path / "counts.tsv").read()
    assert got == exp and len(got.splitlines()) > min(n // 20, 100_000)
    assert all(len(l.split("\t")) == 3 for l in got.splitlines()[:1000])
    cells = {l.split("\t")[2] for l in got.splitlines()}
    # (a read whose barcode error lands on ANOTHER whitelist entry is a perfect match to that entry: a few cells beyond the 10 000)
    assert cells <= set(wl) and 9_000 < len(cells) < 10_000 + n // 1000


def test_c3_device_side_properties(built):
    """The packed 91-bp batch at 10 M reads vs the 1 520-sequence library: kernel nb_align_se_kernel<2, 1, false>."""
    import torch
    dev = torch.device("cuda", 0)
    n = int(os.environ.get("NB_C3_PAIRS", 10_000_000))
    _, seqs, _ = synth.synth_library(40, 38)
    strs = [synth.codes_to_str(s) for s in seqs]
    index = engine.Index(strs, 20)
    assert engine._lib.lib().nb_kernel_name(L2, 20, len(strs), 0) == b"nb_align_se_kernel<2, 1, false, true>"
    d_reads, first = synth.synth_reads_packed_torch(seqs, n, L2, 3, dev, return_codes_of_first=20000)
    res = check_fullsize_properties(index, strs, engine.make_config(), d_reads, first, L2, max_classes=1 << 21)
    assert 0.7 < res["called_frac"] < 0.9


def test_bench_c3_per_cell(built):
    """`bench.py --config c3 --per-cell` (reduced read count): one JSON line whose per_cell object is consistent."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c3", "--per-cell", "--reads", "4000000", "--steps", "2", "--warmup", "1",
                        "--cpu-sample", "50000", "--parity-sample", "5000"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    pc = j["per_cell"]
    assert j["oracle_self_consistency_ok"] is True and pc is not None
    assert pc["reads"] == 4_000_000 and 0.5 * pc["reads"] < pc["called_rows"] <= pc["reads"]
    assert 0 < pc["output_rows"] <= pc["merged_rows"] <= pc["called_rows"] and pc["cells"] == 10_000
    assert j["config"]["read_len"] == 91 and j["config"]["n_seqs"] == 1520 and j["roofline"]["kernel"] in ("nb_align_se_kernel<2, 1, false, true>", "nb_lane_kernel<2, 0>", "nb_anchor_kernel<2, 1, true, false, false>")

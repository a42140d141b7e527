"""GPU parity: HIP path (through the C-ABI) vs the CPU oracle, bit-exact on per-read calls and on class counts.

Integer/bit work: tolerance is zero."""
import numpy as np
import pytest

from nimble_amd import engine, synth
from tests.helpers import assert_same, library_strings, run_gpu

pytestmark = pytest.mark.gpu


def _oracle(strs, k):
    from oracle import oracle
    return oracle.OracleIndex(strs, k)


def _both(strs, k, cfg, codes, max_classes=1 << 16):
    index = engine.Index(strs, k)
    words = synth.pack_codes(codes)
    n, L = codes.shape
    gpu = run_gpu(index, cfg, words, None, L, max_classes)
    blob, off = synth.reads_to_ascii(codes)
    ora = _oracle(strs, k).align(cfg, blob, off, n_threads=8)
    return gpu, ora


def test_c1_50seq_1k_reads(built):
    """BASELINE config 1: 1k synthetic 150-bp single-end reads vs a 50-sequence HLA-like library."""
    _, seqs, strs, _ = library_strings(6, 9)
    seqs, strs = seqs[:50], strs[:50]
    codes = synth.synth_reads(seqs, 1000, 150, seed=1)
    gpu, ora = _both(strs, 20, engine.make_config(), codes)
    assert (ora["reason"] == 0).sum() > 500
    assert_same(gpu, ora, "c1")


@pytest.mark.parametrize("kw", [
    dict(discard_multiple_matches=True),
    dict(strand_filter="fiveprime"),
    dict(strand_filter="threeprime"),
    dict(strand_filter="none", score_threshold=60, score_filter=100),
    dict(score_percent=0.9),
    dict(score_percent=0.0, score_threshold=0, score_filter=0),
])
def test_config_variants(built, kw):
    _, seqs, strs, _ = library_strings(4, 8, seed=7)
    codes = synth.synth_reads(seqs, 3000, 150, seed=11, err=0.02)
    gpu, ora = _both(strs, 20, engine.make_config(**kw), codes)
    assert_same(gpu, ora, str(kw))


@pytest.mark.parametrize("k,L", [(20, 91), (31, 150), (16, 64), (20, 250), (25, 28), (20, 20), (12, 339 - 8)])
def test_k_and_length(built, k, L):
    _, seqs, strs, _ = library_strings(3, 6, locus_len=700, seed=3)
    codes = synth.synth_reads(seqs, 1500, L, seed=5, err=0.015, n_rate=0.003)
    gpu, ora = _both(strs, k, engine.make_config(k=k), codes)
    assert_same(gpu, ora, f"k={k} L={L}")


def test_1k_seq_library(built):
    """1 000-sequence library (16 words per colour row): the BASELINE config 2 library at a size the oracle finishes fast."""
    _, seqs, strs, _ = library_strings(20, 50)
    codes = synth.synth_reads(seqs, 20000, 150, seed=2)
    gpu, ora = _both(strs, 20, engine.make_config(), codes)
    assert (ora["reason"] == 0).sum() > 10000
    assert_same(gpu, ora, "1k-seq")


def test_wide_library_multiword_rows(built):
    """More than 4096 sequences: colour rows span several words per lane (TW = 4 kernel)."""
    _, seqs, strs, _ = library_strings(60, 80, locus_len=240, seed=9)
    assert len(strs) == 4800
    codes = synth.synth_reads(seqs, 4000, 100, seed=4)
    gpu, ora = _both(strs, 20, engine.make_config(), codes)
    assert_same(gpu, ora, "wide")


def test_library_with_both_strands(built):
    """A library that holds sequences AND reverse complements of (other) sequences: k-mers present on both strands, reads whose two
    orientations are both viable (ties go to F, the better coverage wins otherwise). In the lane-per-read path these are exactly the
    reads handed to the wave-per-read kernel (k-mer on both strands / both strands viable): the redo list must reproduce the oracle."""
    _, seqs, strs, _ = library_strings(3, 6, locus_len=500, seed=31)
    comp = str.maketrans("ACGT", "TGCA")
    strs = strs + [s.translate(comp)[::-1] for s in strs[:9]] + [strs[10][:250] + strs[11].translate(comp)[::-1][250:]]
    seqs2 = [np.frombuffer(s.encode(), dtype=np.uint8) for s in strs]
    lut = np.zeros(256, dtype=np.uint8); lut[ord("C")] = 1; lut[ord("G")] = 2; lut[ord("T")] = 3
    codes = synth.synth_reads([lut[x] for x in seqs2], 6000, 150, seed=19, err=0.015)
    for kw in (dict(), dict(strand_filter="fiveprime"), dict(discard_multiple_matches=True)):
        gpu, ora = _both(strs, 20, engine.make_config(**kw), codes)
        assert (ora["reason"] == 0).sum() > 3000
        assert_same(gpu, ora, f"both strands {kw}")


def test_ragged_and_edge_reads(built):
    """Variable lengths (incl. shorter than k and empty), all-N reads, exact reference substrings, reverse complements."""
    _, seqs, strs, _ = library_strings(3, 5, locus_len=400, seed=21)
    rng = np.random.default_rng(3)
    reads = ["", "ACGT", "N" * 150, strs[0][:150], strs[1][100:250], strs[2][:19], strs[2][:20], strs[3][5:26]]
    comp = str.maketrans("ACGT", "TGCA")
    reads += [strs[4][50:200].translate(comp)[::-1], strs[5][0:150][:70] + "N" + strs[5][71:150]]
    for _ in range(500):
        a = rng.integers(0, len(strs)); L = int(rng.integers(0, 180)); o = int(rng.integers(0, 400 - L + 1))
        s = list(strs[a][o:o + L])
        for j in range(len(s)):
            r = rng.random()
            if r < 0.02: s[j] = "ACGT"[rng.integers(0, 4)]
            elif r < 0.025: s[j] = "N"
        s = "".join(s)
        if rng.random() < 0.5: s = s.translate(comp)[::-1]
        reads.append(s)
    max_len = 180
    words, lens = engine.pack_reads(reads, max_len)
    index = engine.Index(strs, 20)
    cfg = engine.make_config()
    gpu = run_gpu(index, cfg, words, lens, max_len)
    ora = _oracle(strs, 20).align_strs(cfg, reads, n_threads=4)
    assert_same(gpu, ora, "ragged")


def test_determinism_and_batches(built):
    """Same batch twice -> identical per-read output; two half batches accumulate to the same class counts."""
    import torch
    _, seqs, strs, _ = library_strings(5, 10, seed=13)
    codes = synth.synth_reads(seqs, 8000, 150, seed=17)
    index = engine.Index(strs, 20)
    cfg = engine.make_config()
    words = synth.pack_codes(codes)
    a = run_gpu(index, cfg, words, None, 150)
    b = run_gpu(index, cfg, words, None, 150)
    for key in ("score", "reason", "orientation", "cov", "set_bits"):
        assert (a[key] == b[key]).all()
    ctx = engine.Context(0, 1 << 16); ctx.upload(index)
    dev = torch.device("cuda", 0)
    for half in (words[:4000], words[4000:]):
        ctx.align_batch(cfg, torch.from_numpy(half.view(np.int64)).to(dev), None, 150, n=half.shape[0])
    ctx.sync()
    rows, counts = ctx.classes()
    got = {r.tobytes(): int(c) for r, c in zip(rows, counts)}
    want = {r.tobytes(): int(c) for r, c in zip(a["rows"], a["counts"])}
    assert got == want


def test_class_table_overflow_is_loud(built):
    """More distinct result sets than max_classes: no silent loss — reading the classes raises NB_ERR_OVERFLOW."""
    import torch
    _, seqs, strs, _ = library_strings(20, 50)
    codes = synth.synth_reads(seqs, 60000, 150, seed=23)
    index = engine.Index(strs, 20)
    ctx = engine.Context(0, max_classes=64)
    ctx.upload(index)
    d = torch.from_numpy(synth.pack_codes(codes).view(np.int64)).cuda()
    ctx.align_batch(engine.make_config(), d, None, 150, n=codes.shape[0])
    ctx.sync()
    with pytest.raises(engine._lib.NimbleError) as e:
        ctx.classes()
    assert e.value.code == -4
    ctx.reset(); ctx.sync()
    rows, counts = ctx.classes()
    assert rows.shape[0] == 0


def test_many_classes_under_contention(built):
    """Tens of thousands of distinct sets interned concurrently by thousands of waves into a tight table (load ~0.5):
    ids must be unique per set and counts exact."""
    _, seqs, strs, _ = library_strings(20, 50)
    codes = synth.synth_reads(seqs, 400000, 150, seed=29, err=0.03)
    index = engine.Index(strs, 20)
    words = synth.pack_codes(codes)
    gpu = run_gpu(index, engine.make_config(), words, None, 150, max_classes=1 << 15)
    keys = [r.tobytes() for r in gpu["rows"]]
    assert len(set(keys)) == len(keys) > 12000                       # every set interned exactly once
    called = gpu["class"] != engine.NO_CLASS
    assert int(gpu["counts"].sum()) == int(called.sum()) == int(gpu["reasons"][0])
    assert (np.bincount(gpu["class"][called], minlength=len(keys)) == gpu["counts"]).all()
    assert (gpu["rows"] != 0).any(axis=1).all()

"""GPU parity of the general kernel (paired templates, tolerated mismatches) against the oracle. Bit-exact."""
import numpy as np
import pytest

from nimble_amd import engine, synth
from tests.helpers import assert_same, library_strings, run_gpu

pytestmark = pytest.mark.gpu


def synth_pairs(seqs, n, l1, l2, seed, frag=300, err=0.01):
    """Fragment of length `frag` from a random allele/strand; r1 = first l1 bases, r2 = reverse complement of the last l2."""
    rng = np.random.Generator(np.random.PCG64(seed))
    r1 = np.zeros((n, l1), dtype=np.uint8); r2 = np.zeros((n, l2), dtype=np.uint8)
    for i in range(n):
        s = seqs[rng.integers(0, len(seqs))]
        o = rng.integers(0, len(s) - frag + 1)
        f = s[o:o + frag].copy()
        if rng.random() < 0.5:
            f = (3 - f)[::-1]
        a, b = f[:l1].copy(), (3 - f[frag - l2:])[::-1].copy()
        kind = rng.random()
        if kind < 0.08:
            a = rng.integers(0, 4, l1, dtype=np.uint8)            # r1 junk
        elif kind < 0.16:
            b = rng.integers(0, 4, l2, dtype=np.uint8)            # r2 junk
        elif kind < 0.22:
            t = seqs[rng.integers(0, len(seqs))]; b = t[:l2].copy()  # mate from elsewhere / wrong strand
        for x in (a, b):
            m = rng.random(x.shape[0]) < err
            x[m] = (x[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
            x[rng.random(x.shape[0]) < 0.002] = 4
        r1[i], r2[i] = a, b
    return r1, r2


def _both(strs, cfg, c1, c2=None):
    from oracle import oracle
    index = engine.Index(strs, cfg.k)
    w1 = synth.pack_codes(c1)
    kw = {}
    if c2 is not None:
        kw = dict(mate_words=synth.pack_codes(c2), mate_max_len=c2.shape[1])
    gpu = run_gpu(index, cfg, w1, None, c1.shape[1], **kw)
    b1, o1 = synth.reads_to_ascii(c1)
    b2, o2 = synth.reads_to_ascii(c2) if c2 is not None else (None, None)
    ora = oracle.OracleIndex(strs, cfg.k).align(cfg, b1, o1, b2, o2, n_threads=8)
    return gpu, ora


@pytest.mark.parametrize("kw", [
    dict(), dict(intersect_level=1), dict(intersect_level=2), dict(require_valid_pair=True),
    dict(strand_filter="fiveprime"), dict(strand_filter="threeprime", intersect_level=1),
    dict(discard_multiple_matches=True, intersect_level=2), dict(score_filter=120, score_percent=0.8),
])
def test_pairs(built, kw):
    _, seqs, strs, _ = library_strings(4, 8, locus_len=600, seed=31)
    c1, c2 = synth_pairs(seqs, 3000, 100, 91, seed=5)
    gpu, ora = _both(strs, engine.make_config(**kw), c1, c2)
    assert (ora["mates"] == 2).sum() > 500
    assert_same(gpu, ora, f"pairs {kw}")


@pytest.mark.parametrize("M", [1, 2, 5])
def test_tolerated_mismatches_single_end(built, M):
    _, seqs, strs, _ = library_strings(3, 4, locus_len=800, seed=8, )
    codes = synth.synth_reads(seqs, 3000, 150, seed=M, err=0.02, n_rate=0.004)
    cfg = engine.make_config(num_mismatches=M)
    gpu, ora = _both(strs, cfg, codes)
    from oracle import oracle
    base = oracle.OracleIndex(strs, 20).align(engine.make_config(), *synth.reads_to_ascii(codes), n_threads=8)
    assert (ora["cov"] > base["cov"]).sum() > 200          # the tolerance actually changes outcomes on this input
    assert_same(gpu, ora, f"M={M}")


def test_tolerated_mismatches_pairs_wide(built):
    _, seqs, strs, _ = library_strings(70, 70, locus_len=320, seed=12)      # 4900 sequences: multi-word rows
    c1, c2 = synth_pairs(seqs, 1500, 120, 100, seed=2, frag=300, err=0.015)
    gpu, ora = _both(strs, engine.make_config(num_mismatches=2, intersect_level=1), c1, c2)
    assert_same(gpu, ora, "pairs M=2 wide")


def test_general_kernel_ragged_pairs(built):
    """Variable-length mates, including mates shorter than k and empty ones."""
    from oracle import oracle
    _, seqs, strs, _ = library_strings(3, 5, locus_len=500, seed=4)
    rng = np.random.default_rng(6)
    comp = str.maketrans("ACGT", "TGCA")
    r1s, r2s = [], []
    for _ in range(600):
        a = rng.integers(0, len(strs)); l1, l2 = int(rng.integers(0, 160)), int(rng.integers(0, 130))
        o = int(rng.integers(0, 500 - 300)); frag = strs[a][o:o + 300]
        if rng.random() < 0.5: frag = frag.translate(comp)[::-1]
        r1s.append(frag[:l1]); r2s.append(frag[300 - l2:].translate(comp)[::-1] if l2 else "")
    w1, n1 = engine.pack_reads(r1s, 160); w2, n2 = engine.pack_reads(r2s, 130)
    cfg = engine.make_config(num_mismatches=1)
    gpu = run_gpu(engine.Index(strs, 20), cfg, w1, n1, 160, mate_words=w2, mate_lens=n2, mate_max_len=130)
    ora = oracle.OracleIndex(strs, 20).align_strs(cfg, r1s, r2s, n_threads=4)
    assert_same(gpu, ora, "ragged pairs")


def test_dictionary_agreement_ops_on_gpu(built):
    """The torch ops the multi-rank reduce runs on the GPU (unique / scatter_reduce / index_add on int64) behave as on CPU."""
    import torch
    from nimble_amd import distributed
    rng = np.random.default_rng(5)
    base = rng.integers(-2**62, 2**62, size=(5000, 16))
    R = np.concatenate([base, base[:1700], base[300:900]])
    g_cpu, i_cpu = distributed._agree(torch.from_numpy(R))
    g_gpu, i_gpu = distributed._agree(torch.from_numpy(R).cuda())
    assert g_gpu.is_cuda and bool((g_gpu[i_gpu].cpu() == torch.from_numpy(R)).all())
    assert {tuple(r) for r in g_gpu.cpu().tolist()} == {tuple(r) for r in g_cpu.tolist()} and g_gpu.shape[0] == 5000
    dense = torch.zeros(g_gpu.shape[0], dtype=torch.int64, device="cuda")
    dense.index_add_(0, i_gpu[:5000], torch.ones(5000, dtype=torch.int64, device="cuda"))
    assert int(dense.sum()) == 5000 and int(dense.max()) == 1


@pytest.mark.parametrize("l1,l2,kw", [(150, 140, dict(intersect_level=1)), (150, 60, dict()), (100, 150, dict(intersect_level=2, require_valid_pair=True)),
                                      (211, 250, dict(intersect_level=1))])
def test_pairs_exact_mode_front_ends(built, l1, l2, kw):
    """Exact-mode paired path: three-per-lane front end when the longer mate has 129..192 positions, one-per-lane otherwise."""
    _, seqs, strs, _ = library_strings(4, 8, locus_len=700, seed=41)
    c1, c2 = synth_pairs(seqs, 2500, l1, l2, seed=9, frag=320 if max(l1, l2) < 300 else 400)
    gpu, ora = _both(strs, engine.make_config(**kw), c1, c2)
    assert_same(gpu, ora, f"pairs exact {l1}/{l2} {kw}")


def test_library_beyond_16k_sequences(built):
    """21 000 sequences (300 loci x 70 alleles): class rows of 329 words (600 in the engine's internal numbering, where every locus starts at
    a word boundary) — beyond the 256 words four-words-per-lane kernels hold; the wave-per-read kernels run with sixteen words per lane
    (`nb_align_se_kernel<., 16, ., .>`, `nb_align_general_kernel<., 16, ., .>`). Single-end through the anchor path (its redo list on the
    sixteen-word kernel), with tolerated mismatches, and paired on the general kernel alone; every read against the oracle."""
    from oracle import oracle
    from tests.helpers import library_strings
    _, seqs, strs, _ = library_strings(300, 70, locus_len=300, seed=5150)
    index = engine.Index(strs, 20)
    assert index.stats()["words_internal"] > 256
    ox = oracle.OracleIndex(strs, 20)
    codes = synth.synth_reads(seqs, 30000, 150, seed=8, err=0.01, n_rate=0.002)
    blob, off = synth.reads_to_ascii(codes)
    for cfg in (engine.make_config(), engine.make_config(num_mismatches=2, discard_multiple_matches=True)):
        gpu = run_gpu(index, cfg, synth.pack_codes(codes), None, 150, max_classes=1 << 18)
        assert any("16" in n.split("<")[1].split(",")[1] for n in gpu["kernels"] if n.startswith(("nb_align_se_kernel", "nb_align_general_kernel"))), gpu["kernels"]
        assert_same(gpu, ox.align(cfg, blob, off, n_threads=16), f"21 000 sequences, {cfg.num_mismatches} mismatches")
    import os
    os.environ["NB_ANCHOR"] = "0"; os.environ["NB_LANE"] = "0"
    try:
        c1, c2 = synth_pairs(seqs, 6000, 120, 91, seed=4)
        cfg = engine.make_config(intersect_level=1)
        gpu = run_gpu(index, cfg, synth.pack_codes(c1), None, 120, max_classes=1 << 18, mate_words=synth.pack_codes(c2), mate_max_len=91)
        assert gpu["kernels"][0].startswith("nb_align_general_kernel") and ", 16," in gpu["kernels"][0], gpu["kernels"]
        b1, o1 = synth.reads_to_ascii(c1); b2, o2 = synth.reads_to_ascii(c2)
        assert_same(gpu, ox.align(cfg, b1, o1, b2, o2, n_threads=16), "21 000 sequences, pairs on the general kernel")
    finally:
        os.environ.pop("NB_ANCHOR", None); os.environ.pop("NB_LANE", None)

"""BASELINE config 2 at FULL size (10 M x 150 bp vs 1 000 sequences) through size-independent properties — the oracle
cannot run 10 M reads in test time, so instead of comparing every read the test checks what must hold at any size:
  * conservation: per-class counts sum to the number of CALLED reads, the reason histogram sums to n;
  * strand symmetry: the reverse-complemented batch gives the same {set -> count} multiset, same scores, orientation flipped
    for every read whose two strands do not tie;
  * idempotence: the same batch twice gives identical per-read words;
  * linearity: counts(first half) + counts(second half) == counts(whole);
  * an oracle-checked sample (first 20 000 reads) bit for bit.
"""
import os

import numpy as np
import pytest

from nimble_amd import engine, synth
from tests.helpers import check_fullsize_properties

pytestmark = pytest.mark.gpu
N, L = 10_000_000, 150


def test_c2_full_size_properties(built):
    import torch
    dev = torch.device("cuda", 0)
    _, seqs, _ = synth.synth_library(20, 50)
    strs = [synth.codes_to_str(s) for s in seqs]
    index = engine.Index(strs, 20)
    assert engine._lib.lib().nb_kernel_name(L, 20, len(strs), 0) == b"nb_align_se_kernel<3, 1, true, true>"
    d_reads, first = synth.synth_reads_packed_torch(seqs, N, L, 2, dev, return_codes_of_first=20000)
    res = check_fullsize_properties(index, strs, engine.make_config(), d_reads, first, L, max_classes=1 << 21)
    assert 0.6 < res["called_frac"] < 0.8


def test_c2_tolerated_mismatches_at_size(built):
    """4 M x 150 bp vs 1 000 sequences with num_mismatches = 2 (anchor kernels passing over bases, their redo list on the general kernel):
    conservation, an oracle-checked sample, idempotence and split linearity (no strand symmetry: SPEC 3's walk is directional once it may
    pass over a base)."""
    import torch
    dev = torch.device("cuda", 0)
    _, seqs, _ = synth.synth_library(20, 50)
    strs = [synth.codes_to_str(s) for s in seqs]
    index = engine.Index(strs, 20)
    d_reads, first = synth.synth_reads_packed_torch(seqs, 4_000_000, L, 21, dev, return_codes_of_first=20000)
    res = check_fullsize_properties(index, strs, engine.make_config(num_mismatches=2), d_reads, first, L, max_classes=1 << 21, strand_symmetry=False)
    assert 0.6 < res["called_frac"] < 0.85


def test_c5_10k_seq_library_full_size(built):
    """BASELINE configs[4]: 10 000-sequence library (100 loci x 100 alleles; colour
    rows are 157 words: the TW = 4 kernel nb_align_se_kernel<3, 4, true>, classes wider than the two inline words of a colour
    descriptor go through `packed`) x 100 M reads (NB_C5_READS to change): an oracle-checked sample of 20 000 reads bit for
    bit, and the four size-independent properties on the whole batch."""
    import torch
    dev = torch.device("cuda", 0)
    n = int(os.environ.get("NB_C5_READS", 100_000_000))
    _, seqs, _ = synth.synth_library(100, 100)
    strs = [synth.codes_to_str(s) for s in seqs]
    assert len(strs) == 10_000
    index = engine.Index(strs, 20)
    st = index.stats()
    assert st["words_per_row"] == 157 and st["n_kmers"] > 2_000_000
    assert engine._lib.lib().nb_kernel_name(L, 20, len(strs), 0) == b"nb_align_se_kernel<3, 4, true, false>"
    d_reads, first = synth.synth_reads_packed_torch(seqs, n, L, 5, dev, return_codes_of_first=20000)
    res = check_fullsize_properties(index, strs, engine.make_config(), d_reads, first, L, max_classes=1 << 23)
    assert 0.5 < res["called_frac"] < 0.7            # (100 alleles per locus: more reads end with an empty intersection than on c2)
    # the sample must have exercised result sets wider than two u64 words (=> every class of those reads was wider than the
    # two inline words of its descriptor: the `packed` gather of intersect_descs)
    rows = res["sample_sets"]
    nz = rows != 0
    span = np.where(nz.any(axis=1), nz.shape[1] - np.argmax(nz[:, ::-1], axis=1) - np.argmax(nz, axis=1), 0)
    assert int((span >= 3).sum()) > 50, f"only {(span >= 3).sum()} sampled result sets span three or more words"

"""Host index build (product, sort-based, LSB-first canonical table) vs the oracle's independent hash-map build."""
import numpy as np
import pytest

from nimble_amd import engine, synth
from tests.helpers import library_strings


@pytest.mark.parametrize("k,n_loci,alleles,locus_len", [(20, 4, 6, 500), (31, 3, 5, 300), (5, 2, 3, 120), (16, 6, 11, 200)])
def test_index_matches_oracle(built, k, n_loci, alleles, locus_len):
    from oracle import oracle
    _, _, strs, _ = library_strings(n_loci, alleles, locus_len, seed=k)
    strs[0] = strs[0][:50] + "N" + strs[0][51:]          # an N breaks k-mers and extensions
    strs[1] = strs[1].lower()                            # case-insensitive
    strs.append("ACGT")                                  # shorter than k (for k > 4)
    strs.append("ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT")  # periodic: palindromes / self-overlaps for even k
    ix, ox = engine.Index(strs, k), oracle.OracleIndex(strs, k)
    st = ix.stats()
    assert st["n_kmers"] == ox.num_kmers and st["n_colours"] == ox.num_colours
    rng = np.random.default_rng(k)
    probes = []
    for s in strs:
        probes += [s[i:i + k] for i in range(0, max(0, len(s) - k + 1), 3)]
    comp = str.maketrans("ACGTacgt", "TGCAtgca")
    probes += [p.translate(comp)[::-1] for p in probes[::5]]          # reverse complements: mostly absent (stranded)
    probes += ["".join("ACGT"[c] for c in rng.integers(0, 4, k)) for _ in range(500)]
    for p in probes:
        if len(p) == k:
            assert ix.query(p) == ox.query(p), p


def test_colour_ids_ascend_by_size(built):
    """The kernels seed the intersection with min(colour id): ids must be ordered by class size."""
    _, _, strs, _ = library_strings(3, 7, 300, seed=5)
    ix = engine.Index(strs, 20)
    import ctypes as ct
    # walk all k-mers, collect (payload colour -> size) through query(); sizes must be monotone in id order
    sizes = {}
    for s in strs:
        for i in range(len(s) - 19):
            q = ix.query(s[i:i + 20])
            sizes[tuple(q["members"])] = len(q["members"])
    assert len(sizes) == ix.stats()["n_colours"]

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


@pytest.mark.parametrize("attempts", [None, "0"])
def test_allelic_variants_at_k23_and_compact_placement_failure(built, monkeypatch, attempts):
    """Found by scripts/fuzz_gpu.py (seed 1315): 3 loci x 50 alleles of 600 bases at k = 23. The compact table stays at its minimum
    size there, a bucket index is almost all 'high bits of the k-mer', and allelic variants of one k-mer used to share home AND alternate
    bucket: no placement, and nb_index_build failed. Now the alternate bucket mixes every bit of the remainder, and a placement that
    fails anyway (forced here with NB_COMPACT_ATTEMPTS=0) falls back to the wide table instead of an error."""
    from oracle import oracle
    if attempts is not None:
        monkeypatch.setenv("NB_COMPACT_ATTEMPTS", attempts)
    _, _, strs, _ = library_strings(3, 50, locus_len=600, seed=1315)
    ix, ox = engine.Index(strs, 23), oracle.OracleIndex(strs, 23)
    st = ix.stats()
    assert st["compact"] == (1 if attempts is None else 0) and st["n_kmers"] == ox.num_kmers and st["anchor_bytes"] > 0
    rng = np.random.default_rng(3)
    comp = str.maketrans("ACGT", "TGCA")
    probes = [s[i:i + 23] for s in strs[::7] for i in range(0, len(s) - 22, 5)]
    probes += [p.translate(comp)[::-1] for p in probes[::3]] + ["".join("ACGT"[c] for c in rng.integers(0, 4, 23)) for _ in range(300)]
    for p in probes:
        assert ix.query(p) == ox.query(p), p


def test_mid_size_library_gets_the_large_first_level(built):
    """2 000 sequences (626 k k-mers): an LDS-sized first level would have groups of 7.3 k-mers, which one-byte pilots cannot place at any
    load — the builder then takes one pilot per 3 k-mers (first level in global memory, as for c5) instead of giving up on the perfect hash
    and with it on the anchor kernels (found by scripts/fuzz_gpu.py: such libraries ran on the wave-per-read kernels only)."""
    from oracle import oracle
    _, _, strs, _ = library_strings(40, 50, locus_len=1098, seed=3128)
    ix = engine.Index(strs, 20)
    st = ix.stats()
    assert st["ph_table_bytes"] > 0 and st["ph_pilot_bytes"] > 150_000 and st["anchor_bytes"] > 0, st
    ox = oracle.OracleIndex(strs, 20)
    rng = np.random.default_rng(2)
    comp = str.maketrans("ACGT", "TGCA")
    probes = [s[i:i + 20] for s in strs[::41] for i in range(0, len(s) - 19, 7)]
    probes += [p.translate(comp)[::-1] for p in probes[::3]] + ["".join("ACGT"[c] for c in rng.integers(0, 4, 20)) for _ in range(300)]
    for p in probes:
        assert ix.query(p) == ox.query(p), p                   # (nb_index_query cross-checks the perfect-hash table on every call)


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


def test_perfect_hash_table_agrees_with_compact_table(built):
    """The BASELINE configs[1] library gets both tables: the compact cuckoo table (wave-per-read kernels) and the perfect-hash table
    of the lane-per-read kernels. nb_index_query answers from the first and cross-checks the second on every call (presence on the
    strand, extension masks, class word): stored k-mers, their reverse complements and random (absent) k-mers must all agree, and
    presence must equal the oracle's."""
    from oracle import oracle
    _, seqs, strs, _ = library_strings(20, 50)
    ix, ox = engine.Index(strs, 20), oracle.OracleIndex(strs, 20)
    st = ix.stats()
    assert st["compact"] == 1 and st["ph_table_bytes"] > 4_000_000 and 60_000 < st["ph_pilot_bytes"] <= 96 * 1024
    assert st["words_internal"] == 20 and st["words_per_row"] == 16          # 20 loci x 50 alleles: one word per locus internally
    rng = np.random.default_rng(1)
    comp = str.maketrans("ACGT", "TGCA")
    probes = []
    for s in strs[::97]:
        probes += [s[i:i + 20] for i in range(0, len(s) - 19, 7)]
    probes += [p.translate(comp)[::-1] for p in probes[::3]]
    probes += ["".join("ACGT"[c] for c in rng.integers(0, 4, 20)) for _ in range(20000)]
    present = 0
    for p in probes:
        a, b = ix.query(p), ox.query(p)
        assert (a is None) == (b is None), p
        present += a is not None
        if a is not None:
            assert a["members"] == b["members"] and a["lext"] == b["lext"] and a["rext"] == b["rext"]
    assert 1500 < present < len(probes) - 20000 + 50


def test_index_with_reverse_complement_sequences(built):
    """A library that contains sequences and reverse complements of sequences: most canonical k-mers are present on BOTH strands (two
    entries per bucket pair in the compact table: it fills to about half the usual load; `both` entries in the perfect-hash table)."""
    from oracle import oracle
    _, _, strs, _ = library_strings(3, 6, locus_len=500, seed=31)
    comp = str.maketrans("ACGT", "TGCA")
    strs = strs + [s.translate(comp)[::-1] for s in strs[:9]] + [strs[10][:250] + strs[11].translate(comp)[::-1][250:]]
    ix, ox = engine.Index(strs, 20), oracle.OracleIndex(strs, 20)
    st = ix.stats()
    assert st["n_kmers"] == ox.num_kmers and st["n_canonical"] < 0.7 * st["n_kmers"] and st["compact"] == 1
    rng = np.random.default_rng(4)
    probes = [s[i:i + 20] for s in strs[::3] for i in range(0, len(s) - 19, 5)]
    probes += [p.translate(comp)[::-1] for p in probes[::2]] + ["".join("ACGT"[c] for c in rng.integers(0, 4, 20)) for _ in range(3000)]
    for p in probes:
        assert ix.query(p) == ox.query(p), p


@pytest.mark.parametrize("with_rc,alleles,wide_table,merged", [(False, 6, False, False), (True, 6, False, False), (False, 70, False, False),
                                                               (False, 6, True, False), (False, 70, False, True)])
def test_anchor_structures(built, with_rc, alleles, wide_table, merged, monkeypatch):
    """Anchor structures of the anchor-extend kernel (csrc/nb_internal.h, AEntry): for library k-mers and their neighbours
      * the anchor is an occurrence of the k-mer in the concatenated library text (which one is the builder's choice),
      * room = how far the text can be followed from there: next k-mers valid (no N, same sequence), not present on both strands, < 255,
      * the sparse-table AND over [tpos, tpos + run] equals the AND of the classes nb_index_query reports for those text k-mers,
    and k-mers that are absent, or present on both strands, have no anchor. Also with loci of 70 alleles (classes of two words, first level of
    the table sized for global memory), next to the wide k-mer table, and with two 70-allele loci merged into one component of three
    words by a shared block (anchors in it are marked: no class words, the wave-per-read kernel maps such reads)."""
    if wide_table:
        monkeypatch.setenv("NB_TABLE", "wide")
    _, _, strs, _ = library_strings(9 if merged else 3, alleles, locus_len=420 if alleles < 20 else (100 if merged else 150), seed=17)
    if merged:
        strs[5] = strs[5][:40] + strs[80][30:60] + strs[5][70:]
    strs[2] = strs[2][:100] + "N" + strs[2][101:]
    comp = str.maketrans("ACGTN", "TGCAN")
    if with_rc:
        strs = strs + [strs[4].translate(comp)[::-1], strs[7][:200]]
    K = 20
    ix = engine.Index(strs, K)
    st = ix.stats()
    assert st["anchor_bytes"] > 0 and st["compact"] == (0 if wide_table else 1)
    if alleles > 64:
        assert st["words_internal"] == (17 if merged else 6) and st["ph_pilot_bytes"] > st["n_canonical"] / 3.5      # two words per locus; one pilot per ~3 k-mers
    text = "".join(s.upper() for s in strs)
    starts = np.cumsum([0] + [len(s) for s in strs])
    seq_of = np.repeat(np.arange(len(strs)), [len(s) for s in strs])

    def kmer_at(t):
        return text[t:t + K] if t + K <= len(text) and seq_of[t] == seq_of[t + K - 1] and "N" not in text[t:t + K] else None

    first, info, occ = {}, {}, {}
    for t in range(len(text)):
        km = kmer_at(t)
        if km is not None:
            first.setdefault(km, t)
            occ.setdefault(km, set()).add(t)
            if km not in info:
                info[km] = (ix.query(km), ix.query(km.translate(comp)[::-1]) is not None)
    checked = both = wide = 0
    for km, t0 in list(first.items())[::3 if alleles < 20 else (23 if merged else 7)]:
        q, is_both = info[km]
        a = ix.anchor_query(km, 0)
        if is_both:
            assert a is None
            both += 1
            continue
        assert a is not None and a["tpos"] in occ[km]
        t0 = a["tpos"]
        if merged and int(seq_of[t0]) < 140:                 # loci 0 and 1 are one component of 140 sequences = three words
            assert a["members"] is None
            wide += 1
            continue
        assert a["members"] == q["members"]
        room = 0
        while room < 254:
            nxt = kmer_at(t0 + room + 1)
            if nxt is None or info[nxt][1]:
                break
            room += 1
        assert a["room"] == room, (km, t0)
        for run in {0, 1, 2, 3, 7, 8, 31, 64, 100, 254, room}:
            want = set(q["members"])
            for i in range(1, min(run, room) + 1):
                want &= set(info[kmer_at(t0 + i)][0]["members"])
            assert ix.anchor_query(km, run)["members"] == sorted(want), (km, run)
        checked += 1
    assert checked > (300 if merged else 600) and (both > 100) == with_rc and (wide > 50) == merged
    rng = np.random.default_rng(3)
    for _ in range(2000):
        assert ix.anchor_query("".join("ACGT"[c] for c in rng.integers(0, 4, K))) is None


@pytest.mark.parametrize("k,n_loci,alleles", [(20, 20, 50), (12, 2, 30), (23, 6, 25), (16, 4, 10), (20, 2, 4), (31, 3, 8)])
def test_presence_filter_has_no_false_negatives_and_few_false_positives(built, k, n_loci, alleles):
    """The presence filter of the scan kernel (nb_internal.h: the m-mers, m = k + 1 - S, of every k-mer of the table in both orientations;
    an LDS-sized one-bit table in front of a blocked Bloom filter): every stretch of a library sequence passes, as given and reverse-
    complemented — a false negative would turn a hit into NO_HIT on the GPU; nb_index_query checks the same for each k-mer it finds — and
    uniform random m-mers pass rarely (the kernel's exact lookups stay few)."""
    _, _, strs, _ = library_strings(n_loci, alleles, locus_len=700, seed=77)
    ix = engine.Index(strs, k)
    if ix.filter_params() is None:
        pytest.skip("library without anchor structures")
    m = ix.filter_params()[0]
    assert k + 1 - m in (1, 2, 4, 8) and m >= min(k, 12), m                  # m = k + 1 - S, S by k and by the size of the library (pf_stride)
    comp = str.maketrans("ACGT", "TGCA")
    n = 0
    for s in strs[::3]:
        for i in range(0, len(s) - k + 1, 5):
            km = s[i:i + k]
            assert ix.filter_maybe(km) is True and ix.filter_maybe(km.translate(comp)[::-1]) is True, km
            assert ix.query(km) is not None
            n += 1
        assert ix.filter_maybe(s) is True                                    # every m-mer of a whole sequence
    assert n > 300
    rng = np.random.default_rng(9)
    rand = ["".join("ACGT"[c] for c in rng.integers(0, 4, m)) for _ in range(20000)]
    lib = set()
    for s in strs:
        for t in (s, s.translate(comp)[::-1]):
            lib.update(t[i:i + m] for i in range(len(t) - m + 1))
    absent = [x for x in rand if x not in lib]
    fp = sum(ix.filter_maybe(x) for x in absent) / len(absent)
    assert len(absent) > 10000 and fp < 0.03, (len(absent), fp)
    assert ix.filter_maybe("ACGTN" * 8) is True                              # an N: nothing is known, the exact path decides


def test_presence_filter_query_contract(built):
    """nb_index_filter_maybe (include/nimble_amd.h): parameters with len = 0, NB_ERR_INVALID for a sequence shorter than m, 2 (None here) for an
    index without the filter (k > 23: no anchor structures)."""
    from nimble_amd._lib import NimbleError
    _, _, strs, _ = library_strings(2, 4, locus_len=300, seed=5)
    ix = engine.Index(strs, 20)
    m, lds = ix.filter_params()
    assert m == 13 and lds is True
    with pytest.raises(NimbleError):
        ix.filter_maybe("ACGTACGT")
    assert ix.filter_maybe(strs[0][:m]) is True and ix.filter_maybe(strs[0][:60]) is True
    wide = engine.Index(strs, 27)
    assert wide.filter_params() is None and wide.filter_maybe("A" * 27) is None

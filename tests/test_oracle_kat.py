"""Known-answer tests that pin the oracle to docs/SPEC.md on cases small enough to work out by hand.

(The Rust aligner has no golden vectors in the reference — parity unpinned, SURVEY.md §8c — so these KATs are derived
from the written spec, not from the reference.)"""
import numpy as np
import pytest

from nimble_amd import engine


def _rand(rng, n):
    return "".join("ACGT"[c] for c in rng.integers(0, 4, n))


def _other(b):
    return "ACGT"[("ACGT".index(b) + 1) % 4]


COMP = str.maketrans("ACGT", "TGCA")


@pytest.fixture(scope="module")
def lib():
    rng = np.random.default_rng(42)
    S = _rand(rng, 200)
    S1 = S[:100] + _other(S[100]) + S[101:]      # one SNP at position 100
    U = _rand(rng, 200)
    return S, [S, S1, U]


def members(row):
    return engine.bits_to_members(row)


def run(strs, reads, mates=None, **kw):
    from oracle import oracle
    cfg = engine.make_config(**kw)
    return oracle.OracleIndex(strs, cfg.k).align_strs(cfg, reads, mates)


def test_exact_and_ambiguous(built, lib):
    S, strs = lib
    r = run(strs, [S[0:150], S[0:90], S[20:170].translate(COMP)[::-1], S[0:19]])
    assert list(r["reason"]) == [0, 0, 0, 1]
    assert members(r["set_bits"][0]) == [0] and r["score"][0] == 150 and r["orientation"][0] == 0
    assert members(r["set_bits"][1]) == [0, 1] and r["score"][1] == 90          # SNP at 100 not covered: both alleles
    assert members(r["set_bits"][2]) == [0] and r["score"][2] == 150 and r["orientation"][2] == 1
    assert list(r["cov"][0]) == [150, 0, 0, 0] and list(r["cov"][2]) == [0, 150, 0, 0] and r["mates"][0] == 1


def test_error_reseed_and_mismatch_tolerance(built, lib):
    S, strs = lib
    read = S[0:75] + _other(S[75]) + S[76:150]
    r0 = run(strs, [read])
    # run 1 covers bases 0..74 (75); re-seed scans k-mer positions 75 (holds the error), 78 (hit): bases 78..149 (72)
    assert r0["score"][0] == 147 and members(r0["set_bits"][0]) == [0]
    r1 = run(strs, [read], num_mismatches=1)
    assert r1["score"][0] == 150 and members(r1["set_bits"][0]) == [0]        # inside a unitig: tolerated
    two = read[:30] + _other(read[30]) + read[31:]
    assert run(strs, [two], num_mismatches=1)["score"][0] < 150               # budget of one


def test_left_extension_only_from_a_late_first_hit(built, lib):
    S, strs = lib
    rng = np.random.default_rng(1)
    junk = _rand(rng, 31)
    junk = junk[:30] + _other(S[30])                   # base 30 differs from the reference
    late = junk + S[31:150]                            # first stride-3 hit at k-mer position 33 >= floor(150/5) = 30
    r = run(strs, [late])
    assert r["score"][0] == 2 + (150 - 33)             # positions 32 and 31 recovered leftwards, then bases 33..149
    early = junk[:20] + _other(S[20]) + S[21:150]      # first hit at 21 < 30: no left extension
    assert run(strs, [early])["score"][0] == 150 - 21


def test_filters(built, lib):
    S, strs = lib
    rng = np.random.default_rng(2)
    tail = _other(S[40]) + _rand(rng, 109)
    part = S[0:40] + tail
    r = run(strs, [part, _rand(rng, 150), "N" * 150])
    assert r["reason"][0] == 4 and r["cov"][0][0] == 40          # 40/150 < score_percent 0.5
    assert list(r["reason"][1:]) == [2, 2]                        # NO_HIT
    assert run(strs, [part], score_percent=0.2)["reason"][0] == 0
    assert run(strs, [part], score_percent=0.2, score_filter=41)["reason"][0] == 9
    assert run(strs, [part], score_percent=0.0, score_threshold=41)["reason"][0] == 3
    amb = S[0:90]
    assert run(strs, [amb], discard_multiple_matches=True)["reason"][0] == 6
    assert run(strs, [amb], strand_filter="threeprime")["reason"][0] == 8
    assert run(strs, [amb], strand_filter="fiveprime")["reason"][0] == 0
    chim = S[0:75] + strs[2][75:150]                              # two references that share nothing: empty intersection
    assert run(strs, [chim])["reason"][0] == 5


def test_pairs_and_intersect_levels(built, lib):
    S, strs = lib
    r1 = S[0:90]                                   # {0,1}
    r2 = S[60:150].translate(COMP)[::-1]           # mate on the opposite strand, covers the SNP: {0}
    for level, want in ((0, [0, 1]), (1, [0]), (2, [0])):
        r = run(strs, [r1], [r2], intersect_level=level)
        assert r["reason"][0] == 0 and members(r["set_bits"][0]) == want and r["score"][0] == 180 and r["mates"][0] == 2
        assert list(r["cov"][0]) == [90, 0, 0, 90]
    u = strs[2][10:100].translate(COMP)[::-1]      # mate from the unrelated reference: disjoint sets
    assert members(run(strs, [r1], [u], intersect_level=1)["set_bits"][0]) == [0, 1, 2]    # falls back to the union
    assert run(strs, [r1], [u], intersect_level=2)["reason"][0] == 5
    junk = "ACGT" * 25
    r = run(strs, [r1], [junk])
    assert r["reason"][0] == 0 and r["mates"][0] == 1 and r["score"][0] == 90
    assert run(strs, [r1], [junk], require_valid_pair=True)["reason"][0] == 7
    # both mates reverse-oriented template: orientation R
    r = run(strs, [r2], [r1])
    assert r["orientation"][0] == 1 and r["score"][0] == 180


@pytest.mark.parametrize("k", [5, 12, 20])
def test_cpp_oracle_equals_the_pure_python_reading_of_the_spec(built, k):
    """oracle/spec_py.py restates docs/SPEC.md §1-§4 a second time (sets and string slices, written from the text): the C++
    oracle must agree with it on every field for random polymorphic libraries, reads with errors / Ns / short lengths, pairs,
    every intersect level, strand filter, mismatch budget and filter setting; the index facts (extensions, unitig-interior
    flags, colours) too."""
    from oracle import oracle, spec_py
    rng = np.random.default_rng(100 + k)
    for trial in range(6):
        anc = [_rand(rng, int(rng.integers(60, 160))) for _ in range(3)]
        seqs = []
        for a in anc:
            for _ in range(int(rng.integers(2, 6))):
                s = list(a)
                for _ in range(int(rng.integers(0, 5))):
                    s[int(rng.integers(0, len(s)))] = "ACGTN"[int(rng.integers(0, 5))]
                seqs.append("".join(s))
        seqs.append(_rand(rng, k - 1))                       # too short to hold a k-mer
        seqs.append(seqs[0][:k + 3].lower())                 # lower case, shares k-mers with sequence 0
        spec = spec_py.SpecIndex(seqs, k)
        oix = oracle.OracleIndex(seqs, k)
        # index facts
        assert oix.num_kmers == len(spec.colour)
        for x in list(spec.colour)[:: max(1, len(spec.colour) // 150)]:
            q = oix.query(x)
            assert q is not None and q["members"] == sorted(spec.colour[x])
            assert {"ACGT"[b] for b in range(4) if q["rext"] >> b & 1} == spec.rext[x] and {"ACGT"[b] for b in range(4) if q["lext"] >> b & 1} == spec.lext[x]
            assert bool(q["r_internal"]) == spec.r_internal(x) and bool(q["l_internal"]) == spec.l_internal(x)
        # reads
        reads, mates = [], []
        for _ in range(120):
            def draw():
                kind = rng.random()
                if kind < 0.12:
                    return _rand(rng, int(rng.integers(0, 70)))
                src = seqs[int(rng.integers(0, len(seqs) - 2))]
                L = int(rng.integers(max(1, k - 3), min(len(src), 90) + 1))
                st = int(rng.integers(0, len(src) - L + 1))
                r = list(src[st:st + L])
                for _ in range(int(rng.integers(0, 4))):
                    if r:
                        r[int(rng.integers(0, len(r)))] = "ACGTN"[int(rng.integers(0, 5))]
                r = "".join(r)
                return r if rng.random() < 0.5 else spec_py.revcomp(r)
            reads.append(draw()); mates.append(draw())
        for cfgkw in (dict(), dict(num_mismatches=1), dict(num_mismatches=3, intersect_level=2), dict(intersect_level=1, require_valid_pair=True),
                      dict(discard_multiple_matches=True, score_percent=0.3), dict(strand_filter="fiveprime", score_threshold=k, score_filter=k),
                      dict(strand_filter="threeprime", score_filter=60), dict(score_threshold=40, score_percent=0.8)):
            kw = dict(k=k, score_threshold=min(20, k + 2), score_filter=min(25, k + 4))
            kw.update(cfgkw)
            cfg = engine.make_config(**kw)
            cfgd = dict(score_threshold=cfg.score_threshold, score_filter=cfg.score_filter, score_percent=cfg.score_percent,
                        num_mismatches=cfg.num_mismatches, discard_multiple_matches=bool(cfg.discard_multiple_matches),
                        intersect_level=cfg.intersect_level, require_valid_pair=bool(cfg.require_valid_pair), strand_filter=cfg.strand_filter)
            for paired in (False, True):
                got = oix.align_strs(cfg, reads, mates if paired else None)
                for i, r in enumerate(reads):
                    want = spec_py.align_template(spec, cfgd, r, mates[i] if paired else None)
                    have = dict(reason=int(got["reason"][i]), score=int(got["score"][i]), orientation=int(got["orientation"][i]), mates=int(got["mates"][i]),
                                set=members(got["set_bits"][i]), cov=[int(x) for x in got["cov"][i]])
                    assert have == want, (k, trial, cfgkw, paired, i, r, mates[i] if paired else None)
        oix.close()

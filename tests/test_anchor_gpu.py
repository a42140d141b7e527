"""The anchor-extend kernel (csrc/nb_anchor.hip) against the oracle on the inputs that stress what is new in it: the walk follows the
concatenated library TEXT from an anchor instead of looking every k-mer up, so the cases are the places where text and graph differ —
sequence ends inside the text, N in library sequences, k-mers that occur twice in one sequence, identical sequences, runs longer than
the sparse table's 255 positions, reads that change allele at every variant, k-mers on both strands next to a run — plus the shapes of
the per-lane record in LDS (every read length mod 32, both orientations, N at every offset). Bit-exact, tolerance zero."""
import numpy as np
import pytest

from nimble_amd import engine, synth
from tests.helpers import assert_same, library_strings, run_gpu

pytestmark = pytest.mark.gpu
COMP = str.maketrans("ACGTN", "TGCAN")


def rc(s):
    return s.translate(COMP)[::-1]


def _check(strs, k, reads, max_len, what, cfg=None, expect_anchor=True):
    from oracle import oracle
    cfg = cfg or engine.make_config(k=k)
    words, lens = engine.pack_reads(reads, max_len)
    gpu = run_gpu(engine.Index(strs, k), cfg, words, lens, max_len)
    assert any(name.startswith("nb_anchor_kernel") for name in gpu["kernels"]) == expect_anchor, gpu["kernels"]
    ora = oracle.OracleIndex(strs, k).align_strs(cfg, reads, n_threads=8)
    assert_same(gpu, ora, what)
    return ora


def test_text_boundaries_and_long_runs(built):
    """Reads that run off the end of one sequence into the next one of the text, across an N of the library, along identical sequences,
    and error-free reads of 320 positions (runs are cut at 254 steps and must resume exactly)."""
    rng = np.random.default_rng(11)
    _, _, strs, _ = library_strings(3, 5, locus_len=640, seed=41)
    strs[3] = strs[3][:300] + "N" + strs[3][301:]
    strs[6] = strs[5]                                        # two identical sequences
    strs.append(strs[2][100:400])                            # a sequence that is a substring of another
    reads = []
    for a in range(len(strs) - 1):                           # the text holds strs[a] + strs[a + 1]: such a read follows the text, not the graph
        for cut in (1, 19, 20, 21, 60, 150, 310):
            reads.append(strs[a][-cut:] + strs[a + 1][:331 - cut])
            reads.append(rc(reads[-1]))
    for a in range(len(strs)):
        s = strs[a]
        for L in (331, 300, 256 + 19, 255 + 19, 254 + 19, 200):
            if len(s) >= L:
                o = int(rng.integers(0, len(s) - L + 1))
                reads += [s[o:o + L], rc(s[o:o + L])]
    reads += [strs[3][150:450], strs[3][282:320], strs[3][301:331], strs[3][200:301], rc(strs[3][250:350])]
    ora = _check(strs, 20, reads, 331, "text boundaries / long runs")
    assert (ora["reason"] == 0).sum() > 0.6 * len(reads) and (ora["reason"] == 5).sum() > 20


def test_repeats_inside_sequences(built):
    """k-mers that occur several times in ONE sequence (tandem repeats, a duplicated block): the anchor is the first occurrence, the read
    may follow any of them."""
    rng = np.random.default_rng(5)
    unit = "".join("ACGT"[c] for c in rng.integers(0, 4, 37))
    block = "".join("ACGT"[c] for c in rng.integers(0, 4, 90))
    def rnd(n):
        return "".join("ACGT"[c] for c in rng.integers(0, 4, n))
    strs = [rnd(80) + unit * 6 + rnd(100), rnd(60) + block + rnd(70) + block + rnd(50), rnd(50) + unit * 3 + rnd(40) + block[:60] + rnd(90),
            "ACGT" * 60, "A" * 100 + rnd(100), rnd(300)]
    reads = []
    for s in strs:
        for _ in range(60):
            L = int(rng.integers(20, 181)); o = int(rng.integers(0, max(1, len(s) - L + 1)))
            r = list(s[o:o + L])
            for j in range(len(r)):
                if rng.random() < 0.01:
                    r[j] = "ACGTN"[rng.integers(0, 5)]
            r = "".join(r)
            reads.append(r if rng.random() < 0.5 else rc(r))
    _check(strs, 20, reads, 180, "repeats")
    _check(strs, 12, reads, 180, "repeats k=12", cfg=engine.make_config(k=12))


def test_allele_switching_reads(built):
    """Chimeric reads that take every other variant from a different allele of the locus: every run ends at an allele difference, the STEP
    lookup succeeds and moves the anchor to another sequence; with errors and N on top."""
    rng = np.random.default_rng(23)
    _, seqs, strs, _ = library_strings(2, 30, locus_len=500, seed=77)
    reads = []
    for _ in range(1500):
        locus = int(rng.integers(0, 2)); L = int(rng.integers(40, 151)); o = int(rng.integers(0, 500 - L + 1))
        parts, pos = [], o
        while pos < o + L:
            a = locus * 30 + int(rng.integers(0, 30)); step = int(rng.integers(5, 60))
            parts.append(strs[a][pos:min(pos + step, o + L)]); pos += step
        r = list("".join(parts))
        for j in range(len(r)):
            x = rng.random()
            if x < 0.006: r[j] = "ACGT"[rng.integers(0, 4)]
            elif x < 0.008: r[j] = "N"
        r = "".join(r)
        reads.append(r if rng.random() < 0.5 else rc(r))
    ora = _check(strs, 20, reads, 150, "allele switching")
    assert 0.3 < (ora["reason"] == 0).mean()
    for kw in (dict(discard_multiple_matches=True), dict(score_percent=0.8), dict(strand_filter="threeprime")):
        _check(strs, 20, reads[:600], 150, f"allele switching {kw}", cfg=engine.make_config(**kw))


def test_record_shapes(built):
    """Every read length from k - 1 to 3 words of bases, both orientations, an N walking through the read: the in-place reverse complement
    of the LDS record and the 32-base windows must be exact at every alignment."""
    _, _, strs, _ = library_strings(2, 4, locus_len=300, seed=3)
    reads = []
    for L in range(19, 132):
        s = strs[L % 8][(L * 7) % 100:(L * 7) % 100 + L]
        reads += [s, rc(s)]
        j = (L * 13) % L
        t = s[:j] + "N" + s[j + 1:]
        reads += [t, rc(t)]
    _check(strs, 20, reads, 131, "record shapes")
    _check(strs, 20, reads, 160, "record shapes, wider layout")


def test_both_strand_kmers_next_to_runs(built):
    """A library with a reverse-complemented block inside otherwise forward sequences: runs must stop in front of k-mers that occur on
    both strands (room), and the reads that touch them go to the wave-per-read kernel."""
    rng = np.random.default_rng(9)
    _, _, strs, _ = library_strings(2, 5, locus_len=400, seed=13)
    strs.append(strs[0][:150] + rc(strs[1][150:260]) + strs[0][260:])
    strs.append(rc(strs[7][100:300]))
    reads = []
    for _ in range(1200):
        a = int(rng.integers(0, len(strs))); L = int(rng.integers(30, 151)); o = int(rng.integers(0, max(1, len(strs[a]) - L + 1)))
        r = strs[a][o:o + L]
        reads.append(r if rng.random() < 0.5 else rc(r))
    _check(strs, 20, reads, 150, "both strands next to runs")


@pytest.mark.parametrize("table", ["default", "wide"])
def test_two_word_classes_and_global_first_level(built, table, monkeypatch):
    """Loci of 90 alleles: classes span two words (sparse table with 16-byte entries, two-word interning), the first level of the
    perfect-hash table is sized for global memory; once next to the compact table, once next to the wide table (the BASELINE configs[4]
    shape: 100-allele loci, table beyond the L2s)."""
    if table == "wide":
        monkeypatch.setenv("NB_TABLE", "wide")
    _, seqs, strs, _ = library_strings(3, 90, locus_len=260, seed=61)
    rng = np.random.default_rng(2)
    codes = synth.synth_reads(seqs, 6000, 150, seed=12, err=0.012, n_rate=0.002)
    reads = ["".join("ACGTN"[c] for c in row) for row in codes]
    for j in range(0, 600, 7):                                 # ragged lengths too
        reads[j] = reads[j][:int(rng.integers(15, 150))]
    ora = _check(strs, 20, reads, 150, f"two-word classes ({table} table)")
    assert (ora["reason"] == 0).mean() > 0.4
    both_words = 0                                             # result sets with members in BOTH class-row words of their locus (alleles 0-63 | 64-89)
    for row in ora["set_bits"][ora["reason"] == 0]:
        within = np.array(engine.bits_to_members(row)) % 90
        both_words += bool((within < 64).any() and (within >= 64).any())
    assert both_words > 100
    _check(strs, 20, reads[:1500], 150, "two-word classes, discard multiple", cfg=engine.make_config(discard_multiple_matches=True))


def test_component_wider_than_the_sparse_entries(built):
    """Two 70-allele loci that share a block form one component of three class-row words: their anchors are marked, reads that touch them
    are mapped by the wave-per-read kernel, reads of the other loci by the anchor kernel — in one batch."""
    _, seqs, strs, _ = library_strings(9, 70, locus_len=120, seed=17)
    strs[5] = strs[5][:40] + strs[80][30:60] + strs[5][70:]
    lut = np.zeros(256, dtype=np.uint8); lut[ord("C")] = 1; lut[ord("G")] = 2; lut[ord("T")] = 3
    codes = synth.synth_reads([lut[np.frombuffer(s.encode(), dtype=np.uint8)] for s in strs], 5000, 100, seed=3, err=0.01)
    reads = ["".join("ACGTN"[c] for c in row) for row in codes]
    reads += [strs[5][10:110], rc(strs[5][5:105]), strs[80][0:100]]
    ora = _check(strs, 20, reads, 100, "wide component")
    assert (ora["reason"] == 0).mean() > 0.4


@pytest.mark.parametrize("alleles,kw", [(8, dict()), (8, dict(intersect_level=1, strand_filter="fiveprime")), (8, dict(intersect_level=2, require_valid_pair=True)),
                                        (8, dict(discard_multiple_matches=True)), (80, dict()), (80, dict(intersect_level=1))])
def test_paired_templates(built, alleles, kw):
    """Paired templates in exact mode: each mate through the anchor kernels (raw alignments), SPEC §4 for the pair in nb_anchor_pair_kernel;
    junk mates, mates from another locus (a union over two components: general kernel), ragged lengths, two-word classes (80-allele loci)."""
    from oracle import oracle
    from tests.test_general_gpu import synth_pairs
    _, seqs, strs, _ = library_strings(3, alleles, locus_len=500, seed=31)
    c1, c2 = synth_pairs(seqs, 3000, 100, 91, seed=5)
    r1 = ["".join("ACGTN"[c] for c in row) for row in c1]
    r2 = ["".join("ACGTN"[c] for c in row) for row in c2]
    rng = np.random.default_rng(1)
    for j in range(0, 3000, 11):
        r1[j] = r1[j][:int(rng.integers(0, 100))]
        r2[(j * 7) % 3000] = r2[(j * 7) % 3000][:int(rng.integers(10, 91))]
    cfg = engine.make_config(**kw)
    w1, l1 = engine.pack_reads(r1, 100)
    w2, l2 = engine.pack_reads(r2, 91)
    gpu = run_gpu(engine.Index(strs, 20), cfg, w1, l1, 100, mate_words=w2, mate_lens=l2, mate_max_len=91)
    assert any(name.startswith("nb_anchor_pair_kernel") for name in gpu["kernels"]), gpu["kernels"]
    ora = oracle.OracleIndex(strs, 20).align_strs(cfg, r1, r2, n_threads=8)
    assert (ora["mates"] == 2).sum() > 300
    assert_same(gpu, ora, f"paired templates {alleles} {kw}")


@pytest.mark.parametrize("env,kernel", [(("NB_ANCHOR", "0"), "nb_lane_kernel"), (("NB_LANE", "0"), "nb_align_se_kernel")])
def test_fallback_kernels_stay_exact(built, monkeypatch, env, kernel):
    """The kernels behind the anchor path are still product code (indexes without anchor structures: k > 23, a library whose components are
    mostly wider than two words, HBM too short for the sparse table): the lane-per-read kernels (NB_ANCHOR=0) and the wave-per-read kernel
    alone (NB_LANE=0) on the 1 000-sequence library, single-end and paired."""
    from oracle import oracle
    from tests.test_general_gpu import synth_pairs
    monkeypatch.setenv(*env)
    _, seqs, strs, _ = library_strings(20, 50)
    codes = synth.synth_reads(seqs, 12000, 150, seed=21)
    cfg = engine.make_config()
    index = engine.Index(strs, 20)
    gpu = run_gpu(index, cfg, synth.pack_codes(codes), None, 150)
    assert gpu["kernels"][0].startswith(kernel), gpu["kernels"]
    blob, off = synth.reads_to_ascii(codes)
    ox = oracle.OracleIndex(strs, 20)
    assert_same(gpu, ox.align(cfg, blob, off, n_threads=8), f"single-end {env}")
    c1, c2 = synth_pairs(seqs, 3000, 100, 91, seed=8)
    gpu = run_gpu(index, cfg, synth.pack_codes(c1), None, 100, mate_words=synth.pack_codes(c2), mate_max_len=91)
    assert gpu["kernels"][0].startswith("nb_align_general_kernel"), gpu["kernels"]
    b1, o1 = synth.reads_to_ascii(c1); b2, o2 = synth.reads_to_ascii(c2)
    assert_same(gpu, ox.align(cfg, b1, o1, b2, o2, n_threads=8), f"pairs {env}")


@pytest.mark.parametrize("lanes,rounds", [(0, 0), (6, 10), (8, 4), (24, 2), (63, 0)])
def test_straggler_retry_pass(built, monkeypatch, lanes, rounds):
    """nb_anchor_kernel hands the reads still walking after NB_ANCHOR_RETRY_ROUNDS rounds, when at most NB_ANCHOR_RETRY_LANES lanes of the wave
    are busy, to a second pass over the workgroup's retry-list segment. Small batches skip that pass unless NB_ANCHOR_RETRY_MIN says otherwise:
    force it here, from 'never' (0 lanes) to 'every walking read, before its first round' (63 lanes, 0 rounds), single-end (ragged lengths,
    Ns) and paired, against the oracle."""
    from oracle import oracle
    from tests.test_general_gpu import synth_pairs
    monkeypatch.setenv("NB_ANCHOR_RETRY_MIN", "0"); monkeypatch.setenv("NB_ANCHOR_RETRY_LANES", str(lanes)); monkeypatch.setenv("NB_ANCHOR_RETRY_ROUNDS", str(rounds))
    _, seqs, strs, _ = library_strings(20, 50)
    codes = synth.synth_reads(seqs, 30000, 150, seed=33, err=0.02, n_rate=0.003)
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 151, codes.shape[0]).astype(np.uint16); lens[::3] = 150
    cfg = engine.make_config()
    index = engine.Index(strs, 20)
    gpu = run_gpu(index, cfg, synth.pack_codes(codes), lens, 150)
    assert gpu["kernels"][0].startswith("nb_anchor_kernel"), gpu["kernels"]
    assert (gpu["kernels"][1] == gpu["kernels"][0]) == (lanes > 0), gpu["kernels"]          # the second launch is reported
    reads = ["".join("ACGTN"[c] for c in row[:L]) for row, L in zip(codes, lens)]
    ox = oracle.OracleIndex(strs, 20)
    assert_same(gpu, ox.align_strs(cfg, reads, n_threads=8), f"single-end, retry lanes {lanes} rounds {rounds}")
    c1, c2 = synth_pairs(seqs, 6000, 100, 91, seed=9)
    gpu = run_gpu(index, cfg, synth.pack_codes(c1), None, 100, mate_words=synth.pack_codes(c2), mate_max_len=91)
    assert gpu["kernels"][0].startswith("nb_anchor_kernel<mates>"), gpu["kernels"]
    b1, o1 = synth.reads_to_ascii(c1); b2, o2 = synth.reads_to_ascii(c2)
    assert_same(gpu, ox.align(cfg, b1, o1, b2, o2, n_threads=8), f"pairs, retry lanes {lanes} rounds {rounds}")


@pytest.mark.parametrize("seed,k_over", [(s, None) for s in range(48)] + [(100 + s, kk) for s in range(6) for kk in (24, 27, 31)])
def test_awkward_libraries(built, seed, k_over):
    """The scenarios of tests/test_anchor_emul_cpu.py::test_awkward_libraries (and three times as many seeds) on the GPU, every read checked
    (the redo list included); every fourth scenario as paired templates (reads i and i + 80 as mates)."""
    from oracle import oracle
    from tests.helpers import weird_scenario
    k, strs, reads, max_len = weird_scenario(seed, k_over)               # k_over: k above 23 (wide table slots), same scenarios
    cfg = engine.make_config(k=k, score_threshold=k, score_filter=k + 2)
    words, lens = engine.pack_reads(reads, max_len)
    if seed % 4 == 3:
        r1, r2 = reads[:80], reads[80:]
        w1, l1 = engine.pack_reads(r1, max_len); w2, l2 = engine.pack_reads(r2, max_len)
        gpu = run_gpu(engine.Index(strs, k), cfg, w1, l1, max_len, mate_words=w2, mate_lens=l2, mate_max_len=max_len)
        assert_same(gpu, oracle.OracleIndex(strs, k).align_strs(cfg, r1, r2, n_threads=4), f"awkward library {seed}, paired (k={k}, kernels {gpu['kernels']})")
        return
    gpu = run_gpu(engine.Index(strs, k), cfg, words, lens, max_len)
    assert_same(gpu, oracle.OracleIndex(strs, k).align_strs(cfg, reads, n_threads=4), f"awkward library {seed} (k={k}, kernels {gpu['kernels']})")


@pytest.mark.parametrize("k,loci,alleles,lds_stage", [(20, 8, 20, True), (12, 3, 10, True), (14, 3, 10, True), (16, 4, 10, True), (23, 4, 10, True), (20, 60, 70, False)])
def test_scan_kernel_behind_the_presence_filter(built, k, loci, alleles, lds_stage):
    """nb_anchor_scan_kernel settles the reads none of whose three probes hits through the presence filter (one m-mer per tile of S k-mer
    positions: LDS stage, blocked Bloom filter, exact lookup for what passes; strides S = 1, 2, 4, 8 by k; the last library is too large
    for the LDS stage). Foreign reads must come out NO_HIT, and a read that hides ONE library k-mer — at every offset, in either
    orientation, next to an N, at the very end of a ragged read — must be found and handed to the wave-per-read kernel, whose result equals
    the oracle's like everything else."""
    rng = np.random.default_rng(1000 + k + loci)
    _, _, strs, _ = library_strings(loci, alleles, locus_len=500, seed=300 + k)
    ix = engine.Index(strs, k)
    assert ix.filter_params() is not None and ix.filter_params()[1] == lds_stage, ix.filter_params()
    L = 150
    rnd = lambda n: "".join("ACGT"[c] for c in rng.integers(0, 4, n))        # noqa: E731
    reads = [rnd(L) for _ in range(3000)]                                     # foreign
    hidden = []
    for off in range(0, L - k + 1):                                           # one library k-mer at every offset, both orientations
        s = strs[int(rng.integers(0, len(strs)))]
        o = int(rng.integers(0, len(s) - k + 1))
        km = s[o:o + k]
        for kk in (km, rc(km)):
            r = rnd(L)
            hidden.append(r[:off] + kk + r[off + k:])
    reads += hidden
    reads += [r[:70] + "N" + r[71:] for r in hidden[::7]]                     # an N somewhere (inside the k-mer for some)
    reads += [r[:int(rng.integers(k, L + 1))] for r in hidden[::5]]           # ragged: the planted k-mer may be cut off
    reads += [rnd(int(rng.integers(0, k))) for _ in range(20)] + ["N" * L, "A" * L, "ACGT" * 37]
    # library reads whose probe positions (first, middle, last k-mer) are all hit by an error: they reach the scan kernel too
    for _ in range(300):
        s = strs[int(rng.integers(0, len(strs)))]
        o = int(rng.integers(0, len(s) - L + 1))
        r = list(s[o:o + L])
        last = L - k
        for p in (k // 2, last // 2 + k // 2, last + k // 2):
            r[p] = "ACGT"[("ACGT".index(r[p]) + 1) % 4] if r[p] in "ACGT" else "A"
        reads.append("".join(r))
    cfg = engine.make_config(k=k, score_threshold=k, score_filter=k)
    ora = _check(strs, k, reads, L, f"scan kernel, k={k}", cfg=cfg)
    no_hit = int((ora["reason"] == 2).sum())
    assert no_hit >= 2500 and int((ora["reason"] != 2).sum()) > 200, (no_hit, len(reads))


def test_scan_kernel_several_batches_per_wave(built):
    """600 000 reads, 95 % of them foreign to the library: every workgroup's segment of the scan list holds several wave-batches per wave, so
    the scan kernel's prefetch of the next batch (list entry, length, record) runs; a few hundred of the foreign reads hide one library
    k-mer. Every read against the oracle."""
    rng = np.random.default_rng(77)
    _, seqs, strs, _ = library_strings(8, 20, locus_len=500, seed=320)
    n, L, k = 600_000, 150, 20
    codes = synth.synth_reads(seqs, n, L, seed=12, frac_random=0.95, err=0.01, n_rate=0.001)
    lut = {c: i for i, c in enumerate("ACGT")}
    for i in rng.integers(0, n, 400):                                       # plant a library k-mer at a random offset
        s = strs[int(rng.integers(0, len(strs)))]
        o = int(rng.integers(0, len(s) - k + 1)); off = int(rng.integers(0, L - k + 1))
        codes[i, off:off + k] = [lut[c] for c in s[o:o + k]]
    lens = rng.integers(0, L + 1, n).astype(np.uint16); lens[::2] = L
    cfg = engine.make_config(k=k)
    gpu = run_gpu(engine.Index(strs, k), cfg, synth.pack_codes(codes), lens, L, max_classes=1 << 18)
    assert any(name.startswith("nb_anchor_scan_kernel") for name in gpu["kernels"]), gpu["kernels"]
    from oracle import oracle
    reads = ["".join("ACGTN"[c] for c in row[:ln]) for row, ln in zip(codes, lens)]
    ora = oracle.OracleIndex(strs, k).align_strs(cfg, reads, n_threads=16)
    assert_same(gpu, ora, "scan kernel, several batches per wave")
    assert int((ora["reason"] == 2).sum()) > 400_000


@pytest.mark.parametrize("M,kw", [(1, dict()), (2, dict(discard_multiple_matches=True)), (3, dict(strand_filter="fiveprime", score_percent=0.8)), (31, dict())])
def test_tolerated_mismatches_on_the_anchor_path(built, M, kw, monkeypatch):
    """num_mismatches > 0 (SPEC 3: a base that differs is passed over where the graph has one way on, M times per read) on the anchor kernels:
    while the read follows the text the walk passes over the base if the text k-mer before it is r_internal and goes on from the CORRECTED
    k-mer (the next STEP lookups see the text's base there); the read's own k-mers over that base stay unsettled for the other orientation;
    left extensions and runs that end for lack of room go to the mismatch-tolerant wave-per-read kernel. Single-end (ragged, Ns, a straggler
    pass that resumes walks with their budget and corrections) and paired, every read against the oracle; NB_ANCHOR_MM=0 (the general kernel
    alone) must say the same."""
    from oracle import oracle
    from tests.test_general_gpu import synth_pairs
    monkeypatch.setenv("NB_ANCHOR_RETRY_MIN", "0"); monkeypatch.setenv("NB_ANCHOR_RETRY_ROUNDS", "4")
    _, seqs, strs, _ = library_strings(12, 30)
    codes = synth.synth_reads(seqs, 40000, 150, seed=50 + M, err=0.015, n_rate=0.003)
    rng = np.random.default_rng(M)
    lens = rng.integers(0, 151, codes.shape[0]).astype(np.uint16); lens[::3] = 150
    cfg = engine.make_config(num_mismatches=M, **kw)
    index = engine.Index(strs, 20)
    gpu = run_gpu(index, cfg, synth.pack_codes(codes), lens, 150)
    assert gpu["kernels"][0].startswith("nb_anchor_kernel") and any(n.startswith("nb_align_general_kernel") for n in gpu["kernels"]), gpu["kernels"]
    reads = ["".join("ACGTN"[c] for c in row[:L]) for row, L in zip(codes, lens)]
    ox = oracle.OracleIndex(strs, 20)
    ora = ox.align_strs(cfg, reads, n_threads=8)
    assert_same(gpu, ora, f"single-end, num_mismatches {M}")
    exact = ox.align_strs(engine.make_config(**kw), reads, n_threads=8)
    assert int((np.asarray(ora["score"]) != np.asarray(exact["score"])).sum()) > 5000            # (the budget matters for these reads)
    c1, c2 = synth_pairs(seqs, 8000, 100, 91, seed=9 + M)
    gpu = run_gpu(index, cfg, synth.pack_codes(c1), None, 100, mate_words=synth.pack_codes(c2), mate_max_len=91)
    assert gpu["kernels"][0].startswith("nb_anchor_kernel<mates>"), gpu["kernels"]
    b1, o1 = synth.reads_to_ascii(c1); b2, o2 = synth.reads_to_ascii(c2)
    assert_same(gpu, ox.align(cfg, b1, o1, b2, o2, n_threads=8), f"pairs, num_mismatches {M}")
    monkeypatch.setenv("NB_ANCHOR_MM", "0")
    gpu = run_gpu(index, cfg, synth.pack_codes(codes[:8000]), lens[:8000], 150)
    assert gpu["kernels"][0].startswith("nb_align_general_kernel"), gpu["kernels"]
    assert_same(gpu, ox.align_strs(cfg, reads[:8000], n_threads=8), f"general kernel alone, num_mismatches {M}")


def test_tolerated_mismatches_short_kmers_other_orientation(built):
    """Found by scripts/fuzz_gpu.py (seed 9016): k = 12, 40-bp reads, num_mismatches = 2. After a base was passed over, the next STEP lookup
    uses the CORRECTED k-mer; its position was then marked as settled for the other orientation's seed scan although the READ's own k-mer
    there had never been looked up — with 12-mers that k-mer is in the table on the other strand often enough, and the other orientation's
    coverage came out 0 instead of 17. Every read against the oracle (all four coverages)."""
    from oracle import oracle
    _, seqs, strs, _ = library_strings(1, 70, locus_len=600, seed=9016)
    codes = synth.synth_reads(seqs, 200_000, 40, seed=9016, frac_random=0.6, err=0.004, n_rate=0.01)
    cfg = engine.make_config(k=12, num_mismatches=2, score_threshold=20, score_filter=25)
    gpu = run_gpu(engine.Index(strs, 12), cfg, synth.pack_codes(codes), None, 40, max_classes=1 << 20)
    assert gpu["kernels"][0].startswith("nb_anchor_kernel"), gpu["kernels"]
    blob, off = synth.reads_to_ascii(codes)
    assert_same(gpu, oracle.OracleIndex(strs, 12).align(cfg, blob, off, n_threads=16), "k = 12, 40 bp, num_mismatches 2")

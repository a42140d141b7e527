"""Differential fuzz on the GPU box: random libraries, read shapes and settings, every read and every class count against the oracle.
    python scripts/fuzz_gpu.py [seconds] [first_seed]
Batches are large enough (200 k reads) for every pass of the anchor path to run: first pass, stragglers' retry list, scan kernel, redo list.
Test infrastructure (uses oracle/ and tests/helpers.py); prints one line per scenario and stops at the first difference."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nimble_amd import engine, synth  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.helpers import assert_same, library_strings, run_gpu  # noqa: E402
from tests.test_general_gpu import synth_pairs  # noqa: E402



def scenario(seed):
    """One random scenario, GPU against the oracle (AssertionError on the first difference); returns its one-line description."""
    rng = np.random.default_rng(seed)
    loci, alleles = int(rng.choice([1, 3, 8, 20, 40])), int(rng.choice([2, 10, 50, 70]))
    L = int(rng.choice([40, 64, 91, 150, 151, 250, 339]))
    locus_len = int(rng.choice([max(L + 20, 300), 600, 1098]))
    k = int(rng.choice([20, 20, 20, 16, 23, 12, 22, 27, 31, 9, 13, 14, 15, 19, 26]))          # (presence-filter strides 1, 2, 4, 8 all occur)
    if rng.random() < 0.08:
        loci, alleles = int(rng.choice([60, 100])), 70                    # classes of two words, thousands of sequences
    _, seqs, strs, _ = library_strings(loci, alleles, locus_len=locus_len, seed=seed)
    err, n_rate, frac_random = float(rng.choice([0.0, 0.004, 0.01, 0.03])), float(rng.choice([0.0, 0.001, 0.01])), float(rng.choice([0.02, 0.1, 0.6]))
    kw = [dict(), dict(discard_multiple_matches=True), dict(strand_filter="fiveprime"), dict(score_percent=0.8, score_threshold=30), dict(strand_filter="threeprime", score_filter=40)][int(rng.integers(0, 5))]
    force_mm = bool(os.environ.get("NB_FUZZ_MM"))                             # NB_FUZZ_MM=1: every scenario tolerates mismatches, k <= 23 (the anchor kernels' path)
    if force_mm and k > 23:
        k = int(rng.choice([12, 16, 20, 23]))
    if force_mm or rng.random() < 0.15:
        kw = dict(kw, num_mismatches=int(rng.integers(1, 5 if force_mm else 3)))   # tolerated mismatches: anchor kernels (k <= 23) or the general kernel
    if "score_threshold" not in kw:
        kw = dict(kw, score_threshold=max(k, 20), score_filter=max(k + 5, kw.get("score_filter", 25)))
    cfg = engine.make_config(k=k, **kw)
    index = engine.Index(strs, k)
    ox = oracle.OracleIndex(strs, k)
    paired = rng.random() < 0.3
    L = min(L, 320 + k - 1, 250 if paired else 339)                        # NB_MAX_POSITIONS; fragments of synth_pairs are 300 bases
    t0 = time.time()
    if paired:
        n = 60_000
        L2 = int(rng.choice([L, 91])) if L >= 91 else L
        c1, c2 = synth_pairs(seqs, n, L, L2, seed=seed)
        pkw = [dict(), dict(intersect_level=1), dict(intersect_level=2, require_valid_pair=True)][int(rng.integers(0, 3))]
        cfg = engine.make_config(k=k, **kw, **pkw)
        gpu = run_gpu(index, cfg, synth.pack_codes(c1), None, L, max_classes=1 << 20, mate_words=synth.pack_codes(c2), mate_max_len=L2)
        b1, o1 = synth.reads_to_ascii(c1); b2, o2 = synth.reads_to_ascii(c2)
        ora = ox.align(cfg, b1, o1, b2, o2, n_threads=16)
        what = f"seed {seed}: pairs {n} x {L}+{L2} bp, {loci}x{alleles} seqs of {locus_len}, k={k}, {kw} {pkw}"
    else:
        n = 200_000
        codes = synth.synth_reads(seqs, n, L, seed=seed, frac_random=frac_random, err=err, n_rate=n_rate)
        lens = None
        if rng.random() < 0.5:
            lens = rng.integers(0, L + 1, n).astype(np.uint16); lens[::2] = L
        gpu = run_gpu(index, cfg, synth.pack_codes(codes), lens, L, max_classes=1 << 20)
        if lens is None:
            blob, off = synth.reads_to_ascii(codes)
            ora = ox.align(cfg, blob, off, n_threads=16)
        else:
            ora = ox.align_strs(cfg, ["".join("ACGTN"[c] for c in row[:ln]) for row, ln in zip(codes, lens)], n_threads=16)
        what = f"seed {seed}: {n} x {L} bp{' ragged' if lens is not None else ''}, {loci}x{alleles} seqs of {locus_len}, k={k}, err {err} N {n_rate} foreign {frac_random}, {kw}"
    assert_same(gpu, ora, what)
    return (f"ok {what}; called {float((ora['reason'] == 0).mean()):.3f}; kernels {[x.split('<')[0] for x in gpu['kernels']]}; {time.time() - t0:.1f} s")



if __name__ == "__main__":
    os.environ.setdefault("NB_ANCHOR_RETRY_MIN", "20000")          # (pairs come in batches of 60 k: run the stragglers' pass there too)
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t_end = time.time() + budget
    done = 0
    while time.time() < t_end:
        print(scenario(seed), flush=True)
        done += 1
        seed += 1
    print(f"{done} scenarios identical to the oracle")

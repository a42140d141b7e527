"""The walk logic of the anchor-extend kernel on the CPU, under AddressSanitizer + UBSan: tests/native/anchor_emul.cpp compiles the DEVICE
code of nimble_amd/csrc/nb_anchor.hip for the host with one lane per wave (tests/native/hipshim) and runs it on the real index structures
(nb_index.cpp). Every read the kernels finish themselves must equal the oracle on reason, score, orientation, mates, the four coverage
scores and the result set; the reads they put on the redo list (on the GPU: the wave-per-read kernel's) must be few. The GPU parity tests
(tests/test_anchor_gpu.py) cover the same inputs with 64 lanes per wave; this one needs no GPU and catches indexing errors as sanitizer
reports instead of GPU faults."""
import os
import struct
import subprocess

import numpy as np
import pytest

from nimble_amd import engine, synth
from tests.helpers import library_strings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMP = str.maketrans("ACGTN", "TGCAN")


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("emul") / "anchor_emul")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "tests", "native", "hipshim"), "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "nimble_amd", "csrc"),
           "-pthread", "-o", exe, os.path.join(ROOT, "tests", "native", "anchor_emul.cpp"), os.path.join(ROOT, "nimble_amd", "csrc", "nb_index.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def _run(emul, tmp_path, strs, k, reads, max_len, cfg, max_redo_frac, thorough=True):
    from oracle import oracle
    words, lens = engine.pack_reads(reads, max_len)
    inp, out = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("<iiii", k, len(strs), len(reads), max_len))
        for s in strs:
            f.write(struct.pack("<i", len(s)) + s.encode())
        f.write(np.ascontiguousarray(words, dtype=np.uint64).tobytes())
        f.write(np.ascontiguousarray(lens, dtype=np.uint16).tobytes())
        f.write(struct.pack("<iiiid", cfg.score_threshold, cfg.score_filter, cfg.discard_multiple_matches, cfg.strand_filter, cfg.score_percent))
    ora = oracle.OracleIndex(strs, k).align_strs(cfg, reads, n_threads=4)
    # one pass of nb_anchor_kernel, and with every other read handed to its second pass (the stragglers' retry list, with the state of the
    # walk) before its first round, after two rounds and after five
    for retry in ((0, 2, 5, None) if thorough else (2, None)):
        got, log = _emulate(emul, inp, out, retry)
        _compare(got, ora, max_redo_frac, log)
        retried = int(log.split("retried ")[1].split(",")[0])
        if retry in (0, None):                                    # (after two or five rounds a scenario of short reads may have nobody left walking)
            assert (retried > 0) == (retry is not None and len(reads) > 50), log
    # tolerated mismatches (num_mismatches 1 and 3): the walk passes over a base while it follows the text and goes on from the corrected
    # k-mer; what the kernels finish themselves must equal the oracle run with the same budget (first pass, and resumed after two rounds)
    for mm in ((1, 3) if thorough else (2,)):
        cfg_m = engine.make_config(**{**_cfg_kwargs(cfg), "num_mismatches": mm})
        ora_m = oracle.OracleIndex(strs, k).align_strs(cfg_m, reads, n_threads=4)
        for retry in ((None, 2) if thorough else (2,)):
            got_m, log_m = _emulate(emul, inp, out, retry, mm=mm)
            _compare(got_m, ora_m, min(1.0, max_redo_frac + 0.15), log_m)
    return got, ora, log


def _cfg_kwargs(cfg):
    return dict(k=cfg.k, score_threshold=cfg.score_threshold, score_filter=cfg.score_filter, score_percent=cfg.score_percent,
                discard_multiple_matches=bool(cfg.discard_multiple_matches), strand_filter=int(cfg.strand_filter))


def _emulate(emul, inp, out, retry, mm=0):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("NB_EMUL_RETRY", None)
    env.pop("NB_EMUL_MM", None)
    if mm:
        env["NB_EMUL_MM"] = str(mm)
    if retry is not None:
        env["NB_EMUL_RETRY"] = str(retry)
    r = subprocess.run([emul, inp, out], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    raw = open(out, "rb").read()
    n, w = struct.unpack_from("<ii", raw, 0)
    rec = np.dtype([("status", "<i4"), ("reason", "<i4"), ("score", "<i4"), ("orientation", "<i4"), ("mates", "<i4"), ("cov", "<u8"), ("set", "<u8", (w,))])
    return np.frombuffer(raw, dtype=rec, offset=8, count=n), r.stdout


def _compare(got, ora, max_redo_frac, log):
    n, w = got.shape[0], got["set"].shape[1] if got["set"].ndim > 1 else 1
    fin = got["status"] == 0
    assert (~fin).mean() <= max_redo_frac, (log, float((~fin).mean()))
    for key in ("reason", "score", "orientation", "mates"):
        bad = np.nonzero(fin & (got[key] != np.asarray(ora[key]).astype(np.int64)))[0]
        assert bad.size == 0, f"{key} differs at {bad[:8]}: emulation {got[key][bad[:8]]} oracle {np.asarray(ora[key])[bad[:8]]}"
    cov = engine.decode_cov(got["cov"])
    bad = np.nonzero(fin & (cov != ora["cov"]).any(axis=1))[0]
    assert bad.size == 0, f"coverage differs at {bad[:8]}"
    bad = np.nonzero(fin & (got["set"].reshape(n, w) != ora["set_bits"]).any(axis=1))[0]
    assert bad.size == 0, f"result sets differ at {bad[:8]}"


def _reads_from(strs, n, L, seed, err=0.012, n_rate=0.002, ragged=False):
    lut = np.zeros(256, dtype=np.uint8); lut[ord("C")] = 1; lut[ord("G")] = 2; lut[ord("T")] = 3
    codes = synth.synth_reads([lut[np.frombuffer(s.upper().replace("N", "A").encode(), dtype=np.uint8)] for s in strs], n, L, seed=seed, err=err, n_rate=n_rate)
    reads = ["".join("ACGTN"[c] for c in row) for row in codes]
    if ragged:
        rng = np.random.default_rng(seed)
        for j in range(0, n, 5):
            reads[j] = reads[j][:int(rng.integers(0, L))]
    return reads


def test_one_word_classes(built, emul, tmp_path):
    _, _, strs, _ = library_strings(4, 8, locus_len=400, seed=7)
    strs[3] = strs[3][:200] + "N" + strs[3][201:]
    reads = _reads_from(strs, 1500, 150, seed=3, ragged=True)
    reads += [strs[0][-90:] + strs[1][:60], strs[2][:150].translate(COMP)[::-1], "N" * 150, "", "ACGT"]
    got, ora, log = _run(emul, tmp_path, strs, 20, reads, 150, engine.make_config(), 0.08)
    assert (got["status"] == 0).sum() > 1300 and (ora["reason"] == 0).sum() > 700, log
    _run(emul, tmp_path, strs, 20, reads[:500], 150, engine.make_config(discard_multiple_matches=True, strand_filter="fiveprime", score_percent=0.8), 0.08)


def test_two_word_classes_long_reads_and_small_k(built, emul, tmp_path):
    _, _, strs, _ = library_strings(2, 80, locus_len=400, seed=23)
    reads = _reads_from(strs, 700, 331, seed=5, err=0.004)
    reads += [strs[5][:331], strs[90][40:371].translate(COMP)[::-1]]                       # error-free: runs are cut at 254 steps and resumed
    # (12-mers of the other strand turn up by chance in most 331-bp reads: those reads are the wave-per-read kernel's; the rest is checked)
    got, ora, log = _run(emul, tmp_path, strs, 12, reads, 331, engine.make_config(k=12), 0.90)
    assert (ora["reason"] == 0).mean() > 0.5 and (got["status"] == 0).sum() > 100, log
    reads = _reads_from(strs, 600, 150, seed=6)
    got, ora, log = _run(emul, tmp_path, strs, 20, reads, 150, engine.make_config(), 0.10)
    assert (got["status"] == 0).sum() > 500, log


def test_both_strands_and_repeats(built, emul, tmp_path):
    rng = np.random.default_rng(4)
    _, _, strs, _ = library_strings(2, 5, locus_len=300, seed=13)
    unit = "".join("ACGT"[c] for c in rng.integers(0, 4, 31))
    strs.append(strs[0][:100] + strs[1][100:200].translate(COMP)[::-1] + strs[0][200:])   # a reverse-complemented block: k-mers on both strands
    strs.append("".join("ACGT"[c] for c in rng.integers(0, 4, 70)) + unit * 5 + "".join("ACGT"[c] for c in rng.integers(0, 4, 90)))
    reads = _reads_from(strs, 900, 120, seed=8, ragged=True)
    _run(emul, tmp_path, strs, 20, reads, 120, engine.make_config(), 0.60)                 # (many reads touch both-strand k-mers: redo list)


@pytest.mark.parametrize("seed,k_over", [(s, None) for s in range(12)] + [(100, 24), (101, 27), (102, 31), (103, 31)])
def test_awkward_libraries(built, emul, tmp_path, seed, k_over):
    """Homopolymers, tandem repeats, palindromes, duplicated / nested / reverse-complemented sequences, Ns, k from 5 to 23 (and 24, 27, 31:
    16-byte table slots), reads that hop between sequences (tests.helpers.weird_scenario): whatever the kernels finish themselves equals the
    oracle, under the sanitizers."""
    from tests.helpers import weird_scenario
    k, strs, reads, max_len = weird_scenario(seed, k_over)
    try:
        _run(emul, tmp_path, strs, k, reads, max_len, engine.make_config(k=k, score_threshold=k, score_filter=k + 2), 1.0, thorough=False)
    except AssertionError as e:
        if "no anchor structures" in str(e):
            pytest.skip("library without anchor structures (most of it on both strands / wide components)")
        raise

"""Shared helpers for the parity tests: run the HIP path through the C-ABI and the oracle on the same inputs."""
import os

import numpy as np

from nimble_amd import engine, synth


def run_gpu(index, cfg, words, lens, max_len, max_classes=1 << 16, device=0, mate_words=None, mate_lens=None, mate_max_len=0):
    import torch
    ctx = engine.Context(device, max_classes=max_classes)
    ctx.upload(index)
    dev = torch.device("cuda", device)
    n = words.shape[0]
    d_reads = torch.from_numpy(words.view(np.int64)).to(dev)
    d_lens = torch.from_numpy(lens.view(np.int16)).to(dev) if lens is not None else None
    out_class = torch.empty(n, dtype=torch.int32, device=dev)
    out_info = torch.empty(n, dtype=torch.int32, device=dev)
    out_cov = torch.empty(n, dtype=torch.int64, device=dev)
    d_mates = torch.from_numpy(mate_words.view(np.int64)).to(dev) if mate_words is not None else None
    d_mlens = torch.from_numpy(mate_lens.view(np.int16)).to(dev) if mate_lens is not None else None
    ctx.align_batch(cfg, d_reads, d_lens, max_len, out_class, out_info, out_cov, mates=d_mates, mate_lens=d_mlens,
                    mate_max_len=mate_max_len, n=n)
    ctx.sync()
    rows, counts = ctx.classes()
    res = {"class": out_class.cpu().numpy().view(np.uint32), "rows": rows, "counts": counts, "reasons": ctx.reasons(), "kernels": ctx.last_kernels()}
    res.update(engine.decode_info(out_info.cpu().numpy().view(np.uint32)))
    res["cov"] = engine.decode_cov(out_cov.cpu().numpy().view(np.uint64))
    W = index.words
    sets = np.zeros((n, W), dtype=np.uint64)
    called = res["class"] != engine.NO_CLASS
    sets[called] = rows[res["class"][called]]
    res["set_bits"] = sets
    ctx.close()
    return res


def assert_same(gpu, ora, what=""):
    for key in ("reason", "score", "orientation", "mates"):
        bad = np.nonzero(np.asarray(gpu[key]).astype(np.int64) != np.asarray(ora[key]).astype(np.int64))[0]
        assert bad.size == 0, f"{what}: {key} differs at {bad[:10]} gpu={np.asarray(gpu[key])[bad[:10]]} oracle={np.asarray(ora[key])[bad[:10]]}"
    bad = np.nonzero((gpu["cov"] != ora["cov"]).any(axis=1))[0]
    assert bad.size == 0, f"{what}: coverage differs at {bad[:10]} gpu={gpu['cov'][bad[:5]]} oracle={ora['cov'][bad[:5]]}"
    bad = np.nonzero((gpu["set_bits"] != ora["set_bits"]).any(axis=1))[0]
    assert bad.size == 0, f"{what}: result sets differ at {bad[:10]}"
    # counts: multiset {set -> count} from the oracle's per-read sets vs the device class table
    called = ora["reason"] == 0
    want = {}
    for row in ora["set_bits"][called]:
        want[row.tobytes()] = want.get(row.tobytes(), 0) + 1
    got = {r.tobytes(): int(c) for r, c in zip(gpu["rows"], gpu["counts"]) if c}
    assert got == want, f"{what}: class counts differ ({len(got)} vs {len(want)} classes)"
    hist = np.bincount(ora["reason"], minlength=len(gpu["reasons"]))
    assert (hist == gpu["reasons"]).all(), f"{what}: reason histogram {gpu['reasons']} vs {hist}"


def library_strings(n_loci, alleles, locus_len=1098, seed=synth.LIB_SEED):
    names, seqs, lineage = synth.synth_library(n_loci, alleles, locus_len, seed)
    return names, seqs, [synth.codes_to_str(s) for s in seqs], lineage


# ---- minimal BAM writer (SAM/BAM spec v1.6 §4.2) for tests: BGZF blocks with the BC extra field + EOF block ----
def _bgzf_block(data):
    import struct
    import zlib
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = comp.compress(data) + comp.flush()
    bsize = len(cdata) + 25
    head = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
    return head + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def write_bam(path, records, refs=(("chr1", 1000),), block=3000):
    """records: dicts with name, flag, seq, optional qual, tags {tag: str | int}. Unaligned coordinates."""
    import struct
    text = b"@HD\tVN:1.6\tSO:unsorted\n"
    raw = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs))
    for name, ln in refs:
        nb = name.encode() + b"\x00"
        raw += struct.pack("<i", len(nb)) + nb + struct.pack("<i", ln)
    code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    for r in records:
        name = r["name"].encode() + b"\x00"
        seq = r["seq"].upper()
        packed = bytearray((len(seq) + 1) // 2)
        for i, ch in enumerate(seq):
            packed[i >> 1] |= code.get(ch, 15) << (0 if i & 1 else 4)
        qual = bytes(r.get("qual", [30] * len(seq)))
        tags = b""
        for t, v in r.get("tags", {}).items():
            if isinstance(v, int):
                tags += t.encode() + b"i" + struct.pack("<i", v)
            elif isinstance(v, (list, tuple)):
                tags += t.encode() + b"BS" + struct.pack("<i", len(v)) + b"".join(struct.pack("<H", x) for x in v)
            else:
                tags += t.encode() + b"Z" + v.encode() + b"\x00"
        body = struct.pack("<iiBBHHHIiii", -1, -1, len(name), 0, 4680, 0, r["flag"], len(seq), -1, -1, 0) + name + bytes(packed) + qual + tags
        raw += struct.pack("<i", len(body)) + body
    with open(path, "wb") as f:
        for i in range(0, len(raw), block):
            f.write(_bgzf_block(raw[i:i + block]))
        f.write(_bgzf_block(b""))


# ---- size-independent properties for batches too large for the oracle (tests/test_fullsize_properties_gpu.py, tests/test_c3_gpu.py) ----
def _revcomp_packed(d_reads, L):
    """Reverse-complement packed reads on the GPU with torch ops (independent of the kernels under test)."""
    import torch
    n = d_reads.shape[0]
    nwb, nwn = (L + 31) // 32, (L + 63) // 64
    dev = d_reads.device
    sh2 = (torch.arange(32, dtype=torch.int64, device=dev) * 2)[None, None, :]
    sh1 = torch.arange(64, dtype=torch.int64, device=dev)[None, None, :]
    out = torch.empty_like(d_reads)
    for lo in range(0, n, 1 << 20):
        w = d_reads[lo:lo + (1 << 20)]
        m = w.shape[0]
        bases = ((w[:, :nwb, None] >> sh2) & 3).reshape(m, nwb * 32)[:, :L]
        isn = ((w[:, nwb:, None] >> sh1) & 1).reshape(m, nwn * 64)[:, :L]
        rb = torch.where(isn.flip(1) == 1, torch.zeros_like(bases), 3 - bases.flip(1))
        pad = torch.zeros((m, nwb * 32), dtype=torch.int64, device=dev); pad[:, :L] = rb
        out[lo:lo + m, :nwb] = (pad.view(m, nwb, 32) << sh2).sum(dim=2)
        npad = torch.zeros((m, nwn * 64), dtype=torch.int64, device=dev); npad[:, :L] = isn.flip(1)
        out[lo:lo + m, nwb:] = (npad.view(m, nwn, 64) << sh1).sum(dim=2)
    return out


def _run(index, cfg, d_reads, L, max_classes):
    import torch
    n = d_reads.shape[0]
    ctx = engine.Context(0, max_classes=max_classes)
    ctx.upload(index)
    oc = torch.empty(n, dtype=torch.int32, device=d_reads.device)
    oi = torch.empty(n, dtype=torch.int32, device=d_reads.device)
    ctx.align_batch(cfg, d_reads, None, L, oc, oi, None, n=n)
    ctx.sync()
    rows, counts = ctx.classes()
    res = dict(cls=oc, info=oi, rows=rows, counts=counts, reasons=ctx.reasons())
    ctx.close()
    return res


def _multiset(rows, counts):
    return {r.tobytes(): int(c) for r, c in zip(rows, counts) if c}




def check_fullsize_properties(index, strs, cfg, d_reads, first_codes, L, max_classes=1 << 21, strand_symmetry=True):
    """conservation, an oracle-checked sample (first_codes = codes of the first reads of d_reads), idempotence, linearity over a
    split, strand symmetry. Returns dict(called_frac, sample_sets)."""
    from oracle import oracle
    n = d_reads.shape[0]
    a = _run(index, cfg, d_reads, L, max_classes)
    info = a["info"].cpu().numpy().view(np.uint32)
    reason = (info >> 16) & 0xFF
    # conservation
    assert int(a["reasons"].sum()) == n and int(a["counts"].sum()) == int((reason == 0).sum()) == int(a["reasons"][0])
    assert (np.bincount(reason, minlength=12) == a["reasons"]).all()
    # oracle-checked sample
    ns = first_codes.shape[0]
    blob, off = synth.reads_to_ascii(first_codes)
    ora = oracle.OracleIndex(strs, index.k).align(cfg, blob, off, n_threads=min(16, os.cpu_count() or 1))
    cls = a["cls"][:ns].cpu().numpy().view(np.uint32)
    sets = np.zeros((ns, index.words), dtype=np.uint64)
    called = cls != engine.NO_CLASS
    sets[called] = a["rows"][cls[called]]
    assert (sets == ora["set_bits"]).all() and ((info[:ns] & 0xFFFF) == ora["score"]).all() and (reason[:ns] == ora["reason"]).all()
    assert (((info[:ns] >> 24) & 1) == ora["orientation"]).all()
    # idempotence
    b = _run(index, cfg, d_reads, L, max_classes)
    assert bool((b["info"] == a["info"]).all()) and _multiset(b["rows"], b["counts"]) == _multiset(a["rows"], a["counts"])
    del b
    # linearity over a split
    h1, h2 = _run(index, cfg, d_reads[: n // 2], L, max_classes), _run(index, cfg, d_reads[n // 2:], L, max_classes)
    merged = _multiset(h1["rows"], h1["counts"])
    for k, v in _multiset(h2["rows"], h2["counts"]).items():
        merged[k] = merged.get(k, 0) + v
    assert merged == _multiset(a["rows"], a["counts"])
    del h1, h2
    if not strand_symmetry:                               # (tolerated mismatches: SPEC 3's walk is directional, the two strands may differ)
        return {"called_frac": float(a["reasons"][0]) / n, "sample_sets": sets}
    # strand symmetry
    rc = _run(index, cfg, _revcomp_packed(d_reads, L), L, max_classes)
    assert _multiset(rc["rows"], rc["counts"]) == _multiset(a["rows"], a["counts"])
    info_rc = rc["info"].cpu().numpy().view(np.uint32)
    assert ((info_rc & 0xFFFF) == (info & 0xFFFF)).all() and (((info_rc >> 16) & 0xFF) == reason).all()
    flipped = ((info_rc >> 24) & 1) != ((info >> 24) & 1)
    assert flipped[reason == 0].mean() > 0.999          # ties between strands (both orientations equal) are the only exceptions
    return {"called_frac": float(a["reasons"][0]) / n, "sample_sets": sets}


def weird_scenario(seed, k=None):
    """A small library + reads built to be awkward for a kernel that follows the library text: homopolymers, short tandem repeats, palindromes
    (k-mers equal to their reverse complement), duplicated and nested sequences, Ns, sequences barely longer than k, reads that hop between
    sequences, reads made of a repeat unit. Returns (k, strs, reads, max_len)."""
    rng = np.random.default_rng(seed)
    k0 = int(rng.choice([5, 8, 11, 15, 20, 23]))
    k = k0 if k is None else int(k)                                        # (k > 23: 16-byte table slots, keys with more than 32 bits above the slot hash)
    def rnd(n):
        return "".join("ACGT"[c] for c in rng.integers(0, 4, n))
    comp = str.maketrans("ACGTN", "TGCAN")
    base = rnd(int(rng.integers(3 * k, 160)))
    pal = rnd(k // 2 + 3); pal = pal + pal.translate(comp)[::-1]
    unit = rnd(int(rng.integers(1, 7)))
    strs = [base, base[:len(base) // 2] + rnd(3) + base[len(base) // 2:], rnd(30) + pal + rnd(30), rnd(10) + unit * (2 * k) + rnd(10),
            "A" * (k + int(rng.integers(0, 30))) + rnd(40), rnd(k + 1), base, rnd(50) + "N" + rnd(50) + "NN" + rnd(k), base[5:5 + 2 * k],
            rnd(60).translate(comp)[::-1]]
    if rng.random() < 0.5:
        strs.append(strs[2].translate(comp)[::-1])                         # a reverse-complemented copy: k-mers on both strands
    max_len = int(rng.choice([k + 3, 40, 77, 131]))
    max_len = max(max_len, k + 1)
    reads = []
    for _ in range(160):
        kind = rng.random()
        if kind < 0.55:
            s = strs[rng.integers(0, len(strs))]; L = int(rng.integers(0, min(len(s), max_len) + 1)); o = int(rng.integers(0, len(s) - L + 1)); r = s[o:o + L]
        elif kind < 0.75:
            a, b = strs[rng.integers(0, len(strs))], strs[rng.integers(0, len(strs))]; c = int(rng.integers(0, max_len)); r = (a[-c:] if c else "") + b
        elif kind < 0.85:
            r = unit * max_len
        else:
            r = rnd(int(rng.integers(0, max_len + 1)))
        r = list(r[:max_len])
        for j in range(len(r)):
            x = rng.random()
            if x < 0.02: r[j] = "ACGT"[rng.integers(0, 4)]
            elif x < 0.03: r[j] = "N"
        r = "".join(r)
        reads.append(r if rng.random() < 0.5 else r.translate(comp)[::-1])
    return k, strs, reads, max_len

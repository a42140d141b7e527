"""BASELINE config 2 at FULL size (10 M x 150 bp vs 1 000 sequences) through size-independent properties — the oracle
cannot run 10 M reads in test time, so instead of comparing every read the test checks what must hold at any size:
  * conservation: per-class counts sum to the number of CALLED reads, the reason histogram sums to n;
  * strand symmetry: the reverse-complemented batch gives the same {set -> count} multiset, same scores, orientation flipped
    for every read whose two strands do not tie;
  * idempotence: the same batch twice gives identical per-read words;
  * linearity: counts(first half) + counts(second half) == counts(whole);
  * an oracle-checked sample (first 20 000 reads) bit for bit.
"""
import numpy as np
import pytest

from nimble_amd import engine, synth

pytestmark = pytest.mark.gpu
N, L = 10_000_000, 150


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


def _run(index, cfg, d_reads, want_rows=True):
    import torch
    n = d_reads.shape[0]
    ctx = engine.Context(0, max_classes=1 << 21)
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


def test_c2_full_size_properties(built):
    import torch
    from oracle import oracle
    dev = torch.device("cuda", 0)
    _, seqs, _ = synth.synth_library(20, 50)
    strs = [synth.codes_to_str(s) for s in seqs]
    index = engine.Index(strs, 20)
    cfg = engine.make_config()
    d_reads, first = synth.synth_reads_packed_torch(seqs, N, L, 2, dev, return_codes_of_first=20000)
    a = _run(index, cfg, d_reads)
    info = a["info"].cpu().numpy().view(np.uint32)
    reason = (info >> 16) & 0xFF
    # conservation
    assert int(a["reasons"].sum()) == N and int(a["counts"].sum()) == int((reason == 0).sum()) == int(a["reasons"][0])
    assert (np.bincount(reason, minlength=12) == a["reasons"]).all()
    assert 0.6 < a["reasons"][0] / N < 0.8
    # oracle-checked sample
    blob, off = synth.reads_to_ascii(first)
    ora = oracle.OracleIndex(strs, 20).align(cfg, blob, off, n_threads=8)
    cls = a["cls"][:20000].cpu().numpy().view(np.uint32)
    sets = np.zeros((20000, index.words), dtype=np.uint64)
    called = cls != engine.NO_CLASS
    sets[called] = a["rows"][cls[called]]
    assert (sets == ora["set_bits"]).all() and ((info[:20000] & 0xFFFF) == ora["score"]).all() and (reason[:20000] == ora["reason"]).all()
    # idempotence
    b = _run(index, cfg, d_reads)
    assert bool((b["info"] == a["info"]).all()) and _multiset(b["rows"], b["counts"]) == _multiset(a["rows"], a["counts"])
    # linearity over a split
    h1, h2 = _run(index, cfg, d_reads[: N // 2]), _run(index, cfg, d_reads[N // 2:])
    merged = _multiset(h1["rows"], h1["counts"])
    for k, v in _multiset(h2["rows"], h2["counts"]).items():
        merged[k] = merged.get(k, 0) + v
    assert merged == _multiset(a["rows"], a["counts"])
    del b, h1, h2
    # strand symmetry
    rc = _run(index, cfg, _revcomp_packed(d_reads, L))
    assert _multiset(rc["rows"], rc["counts"]) == _multiset(a["rows"], a["counts"])
    info_rc = rc["info"].cpu().numpy().view(np.uint32)
    assert ((info_rc & 0xFFFF) == (info & 0xFFFF)).all() and (((info_rc >> 16) & 0xFF) == reason).all()
    flipped = ((info_rc >> 24) & 1) != ((info >> 24) & 1)
    assert flipped[reason == 0].mean() > 0.999          # ties between strands (both orientations equal) are the only exceptions

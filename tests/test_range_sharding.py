"""Several ranks, plain FASTQ input: every rank parses only its own byte range (nimble_amd.align.plan_ranges over
nb_file_count_newlines / nb_file_nth_newline / nb_fastq_open_range). CPU only: gloo, world_size 2 and 3.
  * the ranges of the ranks tile each file, start at record boundaries and hold the SAME record numbers in R1 and R2;
  * the records a rank reads from its range are exactly that slice of a whole-file read;
  * gzip input, files that are not 4 lines per record and R1 / R2 of different length make every rank fall back (plan_ranges ->
    None: batches are dealt round-robin instead); blank lines that slip through the line arithmetic are caught by the record count."""
import ctypes as ct
import gzip
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from nimble_amd import _lib, engine


def _fastq_text(n, seed, newline="\n", trailing=True, max_len=60):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len + 1, n)
    seqs = ["".join("ACGTN"[c] for c in rng.integers(0, 5, L)) for L in lens]
    # quality lines that start with '@' and '+' on purpose: framing must come from the line count, not from those characters
    recs = [f"@r{i} {'x' * int(rng.integers(0, 30))}{newline}{s}{newline}+{newline}{('@+' * len(s))[:len(s)]}{newline}" for i, s in enumerate(seqs)]
    text = "".join(recs)
    return (text if trailing else text[:-len(newline)]), seqs


def _drain(reader, batch=4096, cap=128):
    out = []
    while True:
        b = reader.next_batch(batch, cap)
        if b["n"] == 0:
            return out
        for i in range(b["n"]):
            out.append(engine.unpack_read(b["words"][i], int(b["lens"][i]), b["max_len"]))


def test_newline_helpers_and_range_reader(built, tmp_path):
    from nimble_amd.align import FastqReader
    L = _lib.lib()
    text, seqs = _fastq_text(5000, 3)
    p = tmp_path / "a.fastq"
    p.write_bytes(text.encode())
    raw, path = text.encode(), str(p).encode()
    for lo, hi in ((0, len(raw)), (17, 4001), (4001, 4001), (len(raw) - 1, len(raw)), (0, 0)):
        cnt, last = ct.c_int64(), ct.c_int32()
        _lib.check(L.nb_file_count_newlines(path, lo, hi, ct.byref(cnt), ct.byref(last)))
        assert cnt.value == raw[lo:hi].count(b"\n")
        assert last.value == (raw[hi - 1] if hi > lo else -1)
    newlines = [i for i, c in enumerate(raw) if c == 10]
    for begin, n in ((0, 1), (0, 400), (1234, 1), (1234, 777), (0, len(newlines))):
        off = ct.c_int64()
        _lib.check(L.nb_file_nth_newline(path, begin, n, ct.byref(off)))
        want = [i for i in newlines if i >= begin][n - 1] + 1
        assert off.value == want
    off = ct.c_int64()
    _lib.check(L.nb_file_nth_newline(path, 0, len(newlines) + 1, ct.byref(off)))
    assert off.value == -1                                                             # beyond the last newline: -1, not a guess
    # a range reader returns exactly the records of its byte range
    starts = [0] + [newlines[i] + 1 for i in range(3, len(newlines), 4)]
    for a, b in ((0, 5000), (0, 1), (1200, 3700), (4999, 5000), (2500, 2500)):
        r = FastqReader(str(p), 3, byte_range=(starts[a], starts[b]))
        assert _drain(r, batch=700) == seqs[a:b]
        r.close()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _plan_worker(rank, world, port, paths, q):
    from nimble_amd.align import FastqReader, check_range_complete, plan_ranges
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = plan_ranges(paths, world, rank)
        ranges, got = plan["ranges"] if plan else None, None
        if ranges is not None:
            got = []
            for path, rg in zip(paths, ranges):
                r = FastqReader(path, 2, byte_range=rg)
                got.append(_drain(r))
                r.close()
                check_range_complete(plan, len(got[-1]))
        q.put((rank, ranges, got, None))
    except Exception as e:                              # noqa: BLE001 - handed to the parent, which asserts on it
        q.put((rank, None, None, repr(e)))
    dist.barrier()
    dist.destroy_process_group()


def _plan(paths, world):
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, paths, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,newline,trailing", [(2, "\n", True), (3, "\n", False), (2, "\r\n", True)])
def test_ranges_tile_the_files_and_pair_up(built, tmp_path, world, newline, trailing):
    n = 20_001
    t1, s1 = _fastq_text(n, 11, newline, trailing, max_len=28)       # R1: short, R2: long -> very different byte offsets per record
    t2, s2 = _fastq_text(n, 12, newline, trailing, max_len=120)
    p1, p2 = tmp_path / "r1.fastq", tmp_path / "r2.fastq"
    p1.write_bytes(t1.encode()); p2.write_bytes(t2.encode())
    res = _plan([str(p1), str(p2)], world)
    assert all(err is None for _, _, _, err in res)
    for f, (text, seqs) in enumerate(((t1, s1), (t2, s2))):
        pos, k = 0, 0
        for rank, ranges, got, _ in res:
            b, e = ranges[f]
            assert b == pos and e >= b
            pos = e
            assert got[f] == seqs[k:k + len(got[f])]
            k += len(got[f])
        assert pos == len(text.encode()) and k == n
    for rank, ranges, got, _ in res:                                 # same record numbers in both files, shares within one record of n/world
        assert len(got[0]) == len(got[1]) and abs(len(got[0]) - n / world) <= 1


def test_fallbacks(built, tmp_path):
    text, _ = _fastq_text(4000, 5)
    plain, gz, blank, odd, empty = (tmp_path / x for x in ("a.fastq", "a.fastq.gz", "blank.fastq", "odd.fastq", "empty.fastq"))
    plain.write_bytes(text.encode())
    with gzip.open(gz, "wb") as f:
        f.write(text.encode())
    lines = text.split("\n")
    blank.write_bytes("\n".join(lines[:400] + ["", "", "", ""] + lines[400:]).encode())    # 4 blank lines: line count still a multiple of 4
    odd.write_bytes((text + "@x\nACGT\n").encode())
    empty.write_bytes(b"")
    short, _ = _fastq_text(3999, 6)
    (tmp_path / "short.fastq").write_bytes(short.encode())
    for paths, ok in (([str(gz)], False), ([str(plain), str(gz)], False), ([str(odd)], False), ([str(plain), str(tmp_path / "short.fastq")], False),
                      ([str(plain)], True), ([str(empty)], True)):
        res = _plan(paths, 2)
        assert all(err is None for _, _, _, err in res), (paths, res)
        assert all((ranges is not None) == ok for _, ranges, _, _ in res), paths
    # four blank lines in one block keep the line count a multiple of 4 and every range start on a record: the plan stands, and the
    # record count of the range that holds the block gives it away
    res = _plan([str(blank)], 2)
    assert sum(err is not None and "blank lines" in err for _, _, _, err in res) == 1

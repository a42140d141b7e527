"""Several ranks in `align` (nimble_amd/merge.py + the rank-aware parts of nimble_amd/align.py), driven on the CPU with gloo,
world_size 2: two ranks hold the class tables / per-read rows / merged (cell, UMI, feature) rows that two GPUs would have produced
for alternating batches of one input (here: the oracle's calls stand in for the kernels), run the end-of-job merge, and the files
rank 0 writes must be BYTE-IDENTICAL to those of a single rank that saw every batch:
  * bulk counts       digest-first dictionary agreement + one all-reduce  -> legacy bulk TSV
  * per-read table    part files concatenated in global batch order       -> per-read TSV (plain and .gz)
  * per-cell counts   rows exchanged by hash(cell barcode) mod world, reduced per rank, merged -> `report`-shaped TSV
"""
import ctypes as ct
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from nimble_amd import engine, fused, library, merge, synth
from nimble_amd.align import LibraryRun, _finish_fused, _write_per_read

BATCH = 700


class TableCtx:
    """engine.Context's class-table interface over host arrays: interns result sets in first-seen order, like the device table."""

    def __init__(self, words):
        self.words, self.index, self.rows, self.counts = words, {}, [], []
        self.reason_hist = np.zeros(len(engine.REASONS), dtype=np.int64)

    def add_batch(self, set_bits, reason):
        cls = np.full(len(reason), engine.NO_CLASS, dtype=np.uint32)
        for i in np.nonzero(reason == 0)[0]:
            key = set_bits[i].tobytes()
            cid = self.index.get(key)
            if cid is None:
                cid = self.index[key] = len(self.rows)
                self.rows.append(set_bits[i].copy()); self.counts.append(0)
            self.counts[cid] += 1
            cls[i] = cid
        self.reason_hist += np.bincount(reason, minlength=len(engine.REASONS))
        return cls

    def n_classes(self):
        return len(self.rows)

    def class_rows(self, first, count):
        return np.array(self.rows[first:first + count], dtype=np.uint64).reshape(count, self.words)

    def classes(self):
        return self.class_rows(0, len(self.rows)), np.array(self.counts, dtype=np.int64)

    def reasons(self):
        return self.reason_hist


def _scenario(tmp):
    """Library + oracle calls for 5 000 reads with cell / UMI tags (deterministic: every process rebuilds the same)."""
    from oracle import oracle
    names, seqs, lineage = synth.synth_library(4, 8, locus_len=500, seed=41)
    strs = [synth.codes_to_str(s) for s in seqs]
    lib_path = os.path.join(tmp, "lib.json")
    if not os.path.exists(lib_path):
        fa = os.path.join(tmp, "lib.fasta")
        with open(fa, "w") as f:
            f.write("".join(f">{n}\n{s}\n" for n, s in zip(names, strs)))
        library.generate(fa, None, lib_path)
    lib = library.Library(lib_path)
    codes = synth.synth_reads(seqs, 5000, 120, seed=6, err=0.02)
    blob, off = synth.reads_to_ascii(codes)
    ora = oracle.OracleIndex(strs, 20).align(engine.make_config(), blob, off, n_threads=2)
    rng = np.random.default_rng(9)
    cells = np.array(["".join("ACGT"[c] for c in rng.integers(0, 4, 16)) for _ in range(37)])
    umis = np.array(["".join("ACGT"[c] for c in rng.integers(0, 4, 12)) for _ in range(900)])
    cb = cells[rng.integers(0, len(cells), 5000)].astype("S16").view("S1").reshape(5000, 16)
    umi = umis[rng.integers(0, len(umis), 5000)].astype("S12").view("S1").reshape(5000, 12)
    return lib, ora, cb, umi


def _run_rank(tmp, world, rank, tag, gz):
    """What align() does after the kernels, for the batches of this rank: per-read rows, fused rows, bulk counts; then the merges."""
    lib, ora, cb, umi = _scenario(tmp)
    ext = ".tsv.gz" if gz else ".tsv"
    words = (len(lib.sequences) + 63) // 64
    # per-read run (barcoded shape) and bulk run (legacy shape) over the same calls
    pr = LibraryRun(lib, os.path.join(tmp, f"{tag}_per_read{ext}"), "unstranded", 0, True, world, rank, ctx=TableCtx(words))
    pr.rows, pr.report_path = fused.RowAccumulator(), os.path.join(tmp, f"{tag}_fused.tsv")
    bulk = LibraryRun(lib, os.path.join(tmp, f"{tag}_bulk.tsv"), "unstranded", 0, False, world, rank, ctx=TableCtx(words))
    n = len(ora["reason"])
    for b, lo in enumerate(range(0, n, BATCH)):
        if b % world != rank:
            continue
        sl = slice(lo, min(lo + BATCH, n))
        reason, sets = ora["reason"][sl], ora["set_bits"][sl]
        bulk.ctx.add_batch(sets, reason)
        cls = pr.ctx.add_batch(sets, reason)
        info = (ora["score"][sl].astype(np.uint32) & 0xFFFF) | (reason.astype(np.uint32) << 16) | (ora["orientation"][sl].astype(np.uint32) << 24) | (ora["mates"][sl].astype(np.uint32) << 25)
        cov = ora["cov"][sl].astype(np.uint64)
        cov_raw = cov[:, 0] | (cov[:, 1] << np.uint64(16)) | (cov[:, 2] << np.uint64(32)) | (cov[:, 3] << np.uint64(48))
        _write_per_read(pr, pr.snapshot(), cls, info, cov_raw, cb[sl], None, umi[sl], b)
    pr.finish(None)
    _finish_fused(pr, world, rank, 0.05, False)
    bulk.finish(None)


def _worker(rank, world, port, tmp, gz):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _run_rank(tmp, world, rank, "two", gz)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("gz", [False, True])
def test_two_ranks_write_the_single_rank_files(built, tmp_path, gz):
    tmp = str(tmp_path)
    _run_rank(tmp, 1, 0, "one", gz)                       # single rank: every batch
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tmp, gz)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    ext = ".tsv.gz" if gz else ".tsv"
    for name in (f"per_read{ext}", "fused.tsv", "bulk.tsv"):
        a, b = open(os.path.join(tmp, f"one_{name}"), "rb").read(), open(os.path.join(tmp, f"two_{name}"), "rb").read()
        assert a == b and len(a) > 1000, name
    assert not [f for f in os.listdir(tmp) if f.endswith(".part")]          # part files are removed
    import gzip
    text = (gzip.open if gz else open)(os.path.join(tmp, f"two_per_read{ext}"), "rt").read().splitlines()
    assert text[0].startswith("nimble_features\tnimble_score\tr1_CB\tr1_UB") and len(text) > 3000


@pytest.mark.parametrize("world,coarse", [(2, False), (3, False), (2, True)])
def test_reduce_by_digest_matches_exact_merge(built, world, coarse):
    """reduce_class_counts_hashed against the dictionary merge of round 1 (distributed.merge_dictionaries) on overlapping tables,
    incl. wide rows (157 words) where only 16 bytes per class are all-gathered; two and three ranks (rows owned by every rank); `coarse`:
    the first 64 bits of the digests made to tie on purpose (4 bits of them kept), which must send the agreement down its two-column path."""
    from nimble_amd import distributed
    rng = np.random.default_rng(5)
    W = 157
    pool = np.zeros((400, W), dtype=np.uint64)
    for i in range(400):
        w = rng.integers(0, W - 2)
        pool[i, w:w + 3] = rng.integers(1, 2**63, 3, dtype=np.uint64)
    id_sets = [rng.permutation(400)[:260 - 10 * r] for r in range(world)]
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_reduce_worker, args=(r, world, port, q, pool, id_sets[r], coarse)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows0, counts0 = res[0]
    for r in range(1, world):
        assert res[r][0] is None and (res[r][1] == counts0).all()          # rows on the root only, counts everywhere
    want = {}
    for ids in id_sets:
        for j, i in enumerate(ids):
            want[pool[i].tobytes()] = want.get(pool[i].tobytes(), 0) + int(j + 1)
    assert {r.tobytes(): int(c) for r, c in zip(rows0, counts0)} == want
    assert len(want) == distributed.merge_dictionaries([pool[ids] for ids in id_sets]).shape[0]


def _reduce_worker(rank, world, port, q, pool, ids, coarse=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if coarse:
        real = merge.row_digests_t

        def tied(rows_t):
            d = real(rows_t).clone()
            d[:, 0] &= 15
            return d
        merge.row_digests_t = tied
    rows, counts = merge.reduce_class_counts_hashed(pool[ids], np.arange(1, len(ids) + 1, dtype=np.int64))
    q.put((rank, (rows, counts)))
    dist.barrier()
    dist.destroy_process_group()


def test_tensor_digest_ignores_padding_and_distinguishes_rows():
    """row_digests_t (the digest the multi-rank reduce uses, tensor ops): distinct rows get distinct 128-bit digests, zero words beyond the
    last non-zero one do not change them (ranks may hold rows of the same sets in tables of different width), and the two halves differ."""
    import torch
    rng = np.random.default_rng(2)
    rows = rng.integers(0, 2**63, size=(3000, 12), dtype=np.uint64)
    rows[rng.random((3000, 12)) < 0.8] = 0
    rows[5] = rows[6]                                        # one genuine duplicate
    t = torch.from_numpy(rows.view(np.int64))
    d = merge.row_digests_t(t)
    assert len({tuple(x) for x in d.tolist()}) == len({r.tobytes() for r in rows})
    wide = torch.cat([t, torch.zeros((3000, 9), dtype=torch.int64)], dim=1)
    assert bool((merge.row_digests_t(wide) == d).all())
    swapped = t.clone(); swapped[:, [0, 1]] = swapped[:, [1, 0]]
    moved = (t[:, 0] != t[:, 1])
    assert bool((merge.row_digests_t(swapped)[moved] != d[moved]).any(dim=1).all())      # the word index is part of the hash
    assert bool((d[:, 0] != d[:, 1]).any())


def test_digest_ignores_padding_and_distinguishes_rows():
    rng = np.random.default_rng(0)
    rows = rng.integers(0, 2**63, size=(2000, 16), dtype=np.uint64)
    rows[rng.random((2000, 16)) < 0.8] = 0
    d = merge.row_digests(rows)
    assert len({tuple(x) for x in d.tolist()}) == len({r.tobytes() for r in rows})
    assert (merge.row_digests(np.concatenate([rows, np.zeros((2000, 7), np.uint64)], axis=1)) == d).all()
    assert merge.cell_partition(["ACGT", "ACGA"], 2).tolist() == merge.cell_partition(["ACGT", "ACGA"], 2).tolist()

"""The end-of-job reduce with a group of ONE rank taken through its collectives (NB_MERGE_REHEARSE=1: what bench.py's single-rank rehearsal of
the several-rank path runs on RCCL, here on gloo), with the stage clock on: same classes and counts as went in, rows in digest order."""
import os
import socket

import numpy as np
import pytest


def test_one_rank_through_the_collectives(monkeypatch, capfd):
    import torch.distributed as dist
    from nimble_amd import merge
    if dist.is_initialized():
        pytest.skip("a process group is already up in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1"); monkeypatch.setenv("MASTER_PORT", str(port))
    monkeypatch.setenv("NB_MERGE_REHEARSE", "1"); monkeypatch.setenv("NB_MERGE_TIMING", "1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        rng = np.random.default_rng(5)
        rows = rng.integers(0, 1 << 62, size=(500, 3), dtype=np.int64).view(np.uint64)
        rows[::7, 2] = 0                                                   # (trailing zero words do not change a digest)
        counts = rng.integers(1, 1000, size=500).astype(np.int64)
        g_rows, g_counts = merge.reduce_class_counts_hashed(rows, counts)
        assert g_rows.shape == rows.shape and g_counts.shape == counts.shape
        want = {tuple(r): int(c) for r, c in zip(rows.tolist(), counts.tolist())}
        got = {tuple(r): int(c) for r, c in zip(g_rows.tolist(), g_counts.tolist())}
        assert got == want
        assert "[merge] ms:" in capfd.readouterr().err
        monkeypatch.delenv("NB_MERGE_REHEARSE")
        same_rows, same_counts = merge.reduce_class_counts_hashed(rows, counts)   # one rank, plain: identity
        assert (same_rows == rows).all() and (same_counts == counts).all()
    finally:
        dist.destroy_process_group()


def test_to_host_of_a_cpu_tensor_is_a_view():
    import torch
    from nimble_amd import merge
    t = torch.arange(12, dtype=torch.int64).reshape(3, 4)
    a = merge._to_host(t)
    assert a.shape == (3, 4) and (a == t.numpy()).all()

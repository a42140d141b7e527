"""N > 1 path on CPU: world_size-2 gloo run of the end-of-job dictionary agreement + count all-reduce."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nimble_amd import distributed


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rows(ids, W=3):
    """Deterministic distinct rows keyed by small ints."""
    out = np.zeros((len(ids), W), dtype=np.uint64)
    for j, i in enumerate(ids):
        out[j] = [i * 7 + 1, (i * 1315423911) % (1 << 40), 1 << (i % 64)]
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == 0:
        ids, counts = [5, 1, 9, 3], [10, 20, 30, 40]
    else:
        ids, counts = [3, 7, 5], [1, 2, 3]          # overlaps 3 and 5, private 7; different local order
    rows, cnt = distributed.reduce_class_counts(_rows(ids), np.array(counts, dtype=np.int64), device=torch.device("cpu"))
    q.put((rank, rows.tolist(), cnt.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, rows0, c0), (_, rows1, c1) = res
    assert rows0 == rows1 and c0 == c1                                    # every rank holds the same global result
    want = {1: 20, 3: 41, 5: 13, 7: 2, 9: 30}
    got = {tuple(r): c for r, c in zip(rows0, c0)}
    assert got == {tuple(int(x) for x in _rows([i])[0]): v for i, v in want.items()}


def test_agree_hash_path_equals_exact_path():
    rng = np.random.default_rng(3)
    base = rng.integers(-2**62, 2**62, size=(300, 5))
    R = torch.from_numpy(np.concatenate([base, base[:120], base[50:70]]))
    g1, i1 = distributed._agree(R, use_hash=True)
    g2, i2 = distributed._agree(R, use_hash=False)
    assert g1.shape == g2.shape == (300, 5) and bool((g1[i1] == R).all()) and bool((g2[i2] == R).all())
    assert {tuple(r) for r in g1.tolist()} == {tuple(r) for r in g2.tolist()}


def test_single_rank_identity():
    rows, cnt = distributed.reduce_class_counts(_rows([4, 2]), np.array([3, 9]))
    assert rows.tolist() == _rows([4, 2]).tolist() and cnt.tolist() == [3, 9]


def test_unique_rows_matches_numpy():
    rng = np.random.default_rng(0)
    r = rng.integers(0, 4, size=(500, 4)).astype(np.uint64)
    u, inv = distributed.unique_rows(r)
    assert (u == np.unique(r, axis=0)).all() and (u[inv] == r).all()

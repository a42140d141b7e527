"""Multi-GPU reduce of the per-class count vector. One process per GPU, torch.distributed.

Reads shard across ranks with the index replicated; there is no data-path collective while aligning. Class ids are
assigned per rank in first-seen order, so the end-of-job reduce first agrees on the dictionary (an all-gather of each
rank's class bitset rows, a few KB..MB), then sums ONE dense count vector indexed by the agreed ids with a single
all-reduce (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests). Mirrors what the Rust binary's worker
threads do when they merge their per-thread {feature set -> count} maps at the end of `align`
(process boundary at /root/reference/nimble/__main__.py:195-198)."""
import numpy as np

from .merge import _to_host       # device -> host through pinned blocks (see there: freed pageable arrays stall the GPU)


def unique_rows(rows):
    """(unique rows in lexicographic order (word 0 most significant), inverse index) of a [n, W] uint64 array.
    np.lexsort over the W columns: ~100x faster than np.unique(axis=0) on wide rows."""
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    n, W = rows.shape
    if n == 0:
        return rows, np.zeros(0, dtype=np.int64)
    order = np.lexsort(tuple(rows[:, w] for w in range(W - 1, -1, -1))) if W else np.arange(n)
    srt = rows[order]
    new = np.ones(n, dtype=bool)
    if W:
        new[1:] = (srt[1:] != srt[:-1]).any(axis=1)
    else:
        new[1:] = False
    gid = np.cumsum(new) - 1
    inverse = np.empty(n, dtype=np.int64)
    inverse[order] = gid
    return srt[new], inverse


def merge_dictionaries(rows_list):
    """rows_list: list of [n_i, W] uint64 arrays. Returns the agreed dictionary: unique rows in lexicographic order."""
    W = rows_list[0].shape[1] if rows_list else 0
    allrows = np.concatenate([r.reshape(-1, W) for r in rows_list], axis=0) if rows_list else np.zeros((0, W), np.uint64)
    return unique_rows(allrows)[0]


def _row_hash(t):
    """[n, W] int64 tensor -> [n] int64 content hash (wrap-around multiply/xor-shift per word; any device)."""
    import torch
    h = torch.full((t.shape[0],), 0x243F6A8885A308D3, dtype=torch.int64, device=t.device)
    for w in range(t.shape[1]):
        h = (h ^ t[:, w]) * -7046029254386353131        # 0x9E3779B97F4A7C15 as int64
        h = h ^ (h >> 29)
    return h


def _agree(R, use_hash=True):
    """R [N, W] int64 tensor of every rank's rows, concatenated. Returns (glob [G, W], inv [N]) with glob[inv] == R.
    Fast path: group by 64-bit content hash on the tensor's device, then VERIFY content; any collision -> exact lexsort."""
    import torch
    N = R.shape[0]
    if N == 0:
        return R, torch.zeros(0, dtype=torch.int64, device=R.device)
    if use_hash:
        H = _row_hash(R)
        _, inv = torch.unique(H, return_inverse=True)
        G = int(inv.max().item()) + 1
        first = torch.full((G,), N, dtype=torch.int64, device=R.device)
        first.scatter_reduce_(0, inv, torch.arange(N, device=R.device), reduce="amin")
        glob = R[first]
        if bool((glob[inv] == R).all()):
            return glob, inv
    g, i = unique_rows(_to_host(R).view(np.uint64))
    return torch.from_numpy(g.view(np.int64)).to(R.device), torch.from_numpy(i).to(R.device)


def reduce_class_counts(rows, counts, device=None, group=None):
    """rows [n, W] uint64 / counts [n] int64 of THIS rank -> (global_rows [G, W], global_counts [G]) on every rank.

    With no initialised process group (single GPU) this is the identity (device id order); with several ranks the
    agreed dictionary is the same on every rank (grouped by content hash, content verified)."""
    import torch
    import torch.distributed as dist
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    W = rows.shape[1]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return rows, counts  # one rank: the device class table already holds each set exactly once
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    # 1. dictionary agreement: sizes, then padded rows (all-gather)
    n_local = torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(x.item()) for x in sizes]
    nmax = max(max(sizes), 1)
    pad = torch.zeros((nmax, max(W, 1)), dtype=torch.int64, device=dev)
    if rows.shape[0]:
        pad[:rows.shape[0], :W] = torch.from_numpy(rows.view(np.int64)).to(dev)
    gathered = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(gathered, pad, group=group)
    R = torch.cat([g[:n, :W] for g, n in zip(gathered, sizes)], dim=0)
    glob, inv = _agree(R)
    # 2. ONE all-reduce of the dense per-class count vector (this rank's rows are segment `rank` of the concatenation)
    lo = sum(sizes[:rank])
    dense = torch.zeros(max(glob.shape[0], 1), dtype=torch.int64, device=dev)
    if sizes[rank]:
        dense.index_add_(0, inv[lo:lo + sizes[rank]], torch.from_numpy(counts).to(dev))
    dist.all_reduce(dense, op=dist.ReduceOp.SUM, group=group)
    return _to_host(glob).view(np.uint64).reshape(-1, W), _to_host(dense)[:glob.shape[0]]

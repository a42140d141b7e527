"""End-of-job merge layer for several ranks (one process per GPU, torch.distributed; RCCL over xGMI with backend "nccl",
gloo in the CPU tests). Reads shard across ranks while aligning (index replicated, no data-path collective); this module is
everything that happens afterwards, for the three output shapes of `align` (docs/SPEC.md §5):

  bulk counts      class ids are first-seen per rank, so the ranks first agree on the dictionary BY DIGEST: an all-gather of
                   16 bytes per class (two independent 64-bit content hashes), then ONE all-reduce of the dense count vector
                   indexed by the agreed order; bitset rows travel to the root only, and each distinct row exactly once (from
                   the lowest rank that holds it). Per rank that is 16 B x classes instead of W x 8 B x classes for the
                   gathered rows of round 1 (157 words for a 10 000-sequence library).
  per-read table   every rank writes the rows of ITS batches; the root concatenates the parts in global batch order, which
                   reproduces the single-rank file byte for byte (gzip members concatenate to a valid gzip file).
  per-cell counts  merged (cell, UMI, feature, score) rows are exchanged by hash(cell barcode) mod world (all reads
                   of a UMI land on one rank, as the reference's UR-sorted BAM guarantees, __main__.py:344-347), each rank runs
                   the golden-pinned reducer on its cells, and the root merges the per-rank outputs — each sorted by (cell,
                   feature), cells disjoint — into the order `report` writes (__main__.py:289-293).

Mirrors what the Rust binary's worker threads do when they merge their per-thread maps at the end of `align` (process boundary at
/root/reference/nimble/__main__.py:195-198); the reference itself has no multi-process mode."""
import heapq
import os

import numpy as np

_M1, _M2 = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xC2B2AE3D27D4EB4F)
_S1, _S2 = np.uint64(0x243F6A8885A308D3), np.uint64(0x13198A2E03707344)


def _mix(h):
    h = h ^ (h >> np.uint64(31))
    h = h * np.uint64(0x7FB5D329728EA185)
    h = h ^ (h >> np.uint64(27))
    h = h * np.uint64(0x81DADEF4BC2DD44D)
    return h ^ (h >> np.uint64(33))


def row_digests(rows):
    """[n, W] uint64 bitset rows -> [n, 2] uint64: two independent 64-bit content hashes (position-dependent multiply / xor-shift
    chains with different constants). Equality of the 128-bit digest is taken as equality of the rows (collision odds ~ n^2 / 2^129);
    rows_equal_check() in the tests and the root's verification of every row it receives guard the assumption."""
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    n, W = rows.shape
    h1 = np.full(n, _S1, dtype=np.uint64)
    h2 = np.full(n, _S2, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for w in range(W):
            col = rows[:, w]
            nz = col != 0                                 # zero words do not advance the chains: digests do not depend on trailing zeros
            k = np.uint64(w + 1)
            h1 = np.where(nz, _mix((h1 ^ col) * _M1 + k), h1)
            h2 = np.where(nz, _mix((h2 + col) * _M2 ^ (k * _M1)), h2)
    return np.stack([h1, h2], axis=1)


def _world(group=None):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _all_gather_var(t, group, dev):
    """all-gather of [n_r, C] int64 tensors with different n_r: sizes first, then padded blocks. Returns the list of per-rank tensors."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    nmax = max(max(sizes), 1)
    pad = torch.zeros((nmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
    pad[:t.shape[0]] = t.to(dev)
    out = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return [o[:s] for o, s in zip(out, sizes)]


def _lsr(h, n):
    """logical right shift of an int64 tensor (torch's >> is arithmetic)"""
    return (h >> n) & ((1 << (64 - n)) - 1)


def row_digests_t(rows_t):
    """[n, W] int64 tensor (bit patterns of the bitset rows) -> [n, 2] int64 on the same device: two independent 64-bit content hashes,
    each the wrap-around SUM over the non-zero words of a mixed (word index, value): order-free, independent of trailing zero words,
    one pass of elementwise kernels whatever W is (the chained digests of row_digests need W dependent steps)."""
    import torch
    n, W = rows_t.shape
    k = torch.arange(1, W + 1, dtype=torch.int64, device=rows_t.device)
    nz = (rows_t != 0).to(torch.int64)

    def mix(h, c1, c2):
        h = h ^ _lsr(h, 31); h = h * c1
        h = h ^ _lsr(h, 27); h = h * c2
        return h ^ _lsr(h, 33)
    x1 = mix((rows_t ^ (k * -7046029254386353131)[None, :]) * -4417276706812531889, 0x7FB5D329728EA185, -9089946957438231219)
    x2 = mix((rows_t + (k * 0x243F6A8885A308D3)[None, :]) * 0x13198A2E03707345, -49064778989728563, -4265267296055464877)
    return torch.stack([(x1 * nz).sum(dim=1), (x2 * nz).sum(dim=1)], dim=1)


class _StageClock:
    def __init__(self, dev):
        import time
        self.dev, self.t, self.parts, self.time = dev, time.perf_counter(), [], time

    def __call__(self, label, last=False):
        import sys
        import torch
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)
        now = self.time.perf_counter()
        self.parts.append(f"{label} {(now - self.t) * 1e3:.2f}")
        self.t = now
        if last:
            print("[merge] ms: " + ", ".join(self.parts), file=sys.stderr)


def _to_host(t):
    """Device tensor -> numpy array through a PINNED host buffer of torch's caching host allocator. A plain `.cpu()` lands in pageable memory,
    which the HIP runtime registers with the GPU for the length of the copy when it is larger than a staging buffer; when numpy frees (unmaps)
    such an array the driver's MMU notifier evicts and restores every queue of the process: the GPU then starts nothing for 25-35 ms
    (profiles/r03_dist_rehearsal.md: the 11 MB of merged rows, freed inside the job). Pinned blocks are never unmapped while the process runs."""
    import torch
    if t.device.type != "cuda":
        return t.numpy()
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t)
    return h.numpy()                     # (the array keeps the tensor, and with it the pinned block, alive)


def reduce_class_counts_hashed(rows, counts, device=None, group=None, root=0, compute_device=None):
    """rows [n, W] uint64 / counts [n] int64 of THIS rank -> (global_rows, global_counts).
    `device` is where the collective buffers live (the GPU under RCCL, the CPU under gloo), `compute_device` where the tensor ops run
    (default: the same; a gloo rehearsal on a GPU box computes on the GPU and only moves the collective buffers through the host).

    Every rank gets global_counts (dense, in the agreed order = ascending 128-bit digest); global_rows is returned on the root only
    (None elsewhere). One rank: identity (device id order). All the heavy steps run as tensor ops on `device` (the GPU under RCCL, the
    CPU in the gloo tests): digests, one sort-based unique of the gathered digests, the dense count vector; rows travel point to point,
    each distinct row once, and only from ranks other than the root."""
    import torch
    import torch.distributed as dist
    world, rank = _world(group)
    on_device = isinstance(rows, torch.Tensor)        # (rows, counts) = Context.classes_device(): no host round trip before the collectives
    if not on_device:
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.int64)
    if world == 1 and os.environ.get("NB_MERGE_REHEARSE") != "1":       # (NB_MERGE_REHEARSE=1: the collectives below with a group of one rank)
        return (_to_host(rows).view(np.uint64), _to_host(counts)) if on_device else (rows, counts)
    comm = device if device is not None else torch.device("cpu")
    dev = compute_device if compute_device is not None else comm
    n, W = rows.shape
    clock = _StageClock(dev) if os.environ.get("NB_MERGE_TIMING") == "1" else None      # stage clock on stderr (synchronises between stages)
    rows_t = rows.to(dev) if on_device else torch.from_numpy(rows.view(np.int64)).to(dev)
    dig = row_digests_t(rows_t) if n else torch.zeros((0, 2), dtype=torch.int64, device=dev)
    if clock: clock("digests")
    # 1. dictionary agreement by digest: 16 bytes per class
    parts = _all_gather_var(dig.to(comm), group, comm)
    sizes = [int(p.shape[0]) for p in parts]
    allv = torch.cat(parts, dim=0).to(dev)
    if clock: clock("all-gather of digests")
    owner_of = torch.repeat_interleave(torch.arange(world, dtype=torch.int64, device=dev), torch.tensor(sizes, dtype=torch.int64, device=dev))
    if allv.shape[0] == 0:
        return (np.zeros((0, W), dtype=np.uint64) if rank == root else None), np.zeros(0, dtype=np.int64)
    # unique over the first 64 bits (one radix sort), second 64 bits checked per group; a 64-bit tie between different rows (odds ~1e-8 for
    # a million classes) falls back to the two-column unique
    u1, inv = torch.unique(allv[:, 0], return_inverse=True)
    first2 = torch.zeros(u1.shape[0], dtype=torch.int64, device=dev).scatter_(0, inv, allv[:, 1])
    if bool((first2[inv] != allv[:, 1]).any()):
        uniq, inv = torch.unique(allv, dim=0, return_inverse=True)
    else:
        uniq = torch.stack([u1, first2], dim=1)
    G = int(uniq.shape[0])
    if clock: clock("unique")
    lo = sum(sizes[:rank])
    my_gid = inv[lo:lo + n]
    # 2. ONE all-reduce of the dense count vector
    dense = torch.zeros(G, dtype=torch.int64, device=dev)
    if n:
        dense.index_add_(0, my_gid, counts.to(dev) if on_device else torch.from_numpy(counts).to(dev))
    dense_c = dense.to(comm)
    dist.all_reduce(dense_c, op=dist.ReduceOp.SUM, group=group)
    global_counts = _to_host(dense_c)
    if clock: clock("dense vector, all-reduce, counts to the host")
    # 3. rows to the root: each distinct row once, from its owner (the lowest rank that holds the digest); the root's own rows stay put
    owner = torch.full((G,), world, dtype=torch.int64, device=dev).scatter_reduce_(0, inv, owner_of, reduce="amin", include_self=True)
    mine = owner[my_gid] == rank if n else torch.zeros(0, dtype=torch.bool, device=dev)
    sel = torch.nonzero(mine).flatten()
    sel = sel[torch.argsort(my_gid[sel])]                                   # in ascending global id: the order the root files them in
    global_rows = None
    if clock: clock("owners")
    if rank == root:
        g = torch.zeros((G, W), dtype=torch.int64, device=dev)
        g[my_gid[sel]] = rows_t[sel]
        per_owner = torch.bincount(owner, minlength=world + 1).cpu().tolist()
        for r in range(world):
            if r == root or per_owner[r] == 0:
                continue
            block = torch.empty((per_owner[r], W), dtype=torch.int64, device=comm)
            dist.recv(block, src=r, group=group)
            block, at = block.to(dev), torch.nonzero(owner == r).flatten()
            if not bool((row_digests_t(block) == uniq[at]).all()):         # every received row must hash to the digest it is filed under
                raise RuntimeError(f"class dictionary agreement: a row received from rank {r} does not match its digest")
            g[at] = block
        global_rows = _to_host(g).view(np.uint64).reshape(G, W)
    elif sel.numel():
        dist.send(rows_t[sel].contiguous().to(comm), dst=root, group=group)
    if clock: clock("rows to the root", last=True)
    return global_rows, global_counts


def _gather_var_to_root(t, group, dev, root):
    """gather of [n_r, C] int64 tensors with different n_r to `root` (list of per-rank tensors there, None elsewhere)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    nmax = max(max(sizes), 1)
    pad = torch.zeros((nmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
    pad[:t.shape[0]] = t.to(dev)
    if dist.get_backend(group) == "nccl":                                   # RCCL has no gather primitive for lists in older builds: all-gather, root keeps
        out = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(out, pad, group=group)
        return [o[:s] for o, s in zip(out, sizes)] if rank == root else None
    out = [torch.zeros_like(pad) for _ in range(world)] if rank == root else None
    dist.gather(pad, out, dst=root, group=group)
    return [o[:s] for o, s in zip(out, sizes)] if rank == root else None


# ---------------------------------------------------------------------------------------------------------------------------
# per-cell counting across ranks
def cell_partition(cb_strings, world):
    """hash(cell barcode) mod world for every string of `cb_strings` (FNV-1a over the bytes: identical on every rank and in every
    process, unlike Python's salted hash())."""
    out = np.empty(len(cb_strings), dtype=np.int64)
    for i, s in enumerate(cb_strings):
        h = 0xCBF29CE484222325
        for b in s.encode():
            h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        out[i] = h % world
    return out


def exchange_rows_by_cell(cb_strings, cb_ids, umi_keys, feat_strings, feat_ids, scores, group=None):
    """Merged rows of THIS rank -> the rows of the cells this rank owns (hash(cell barcode) mod world), from all ranks.
    Strings travel with the rows (all_gather_object of the per-destination slices: cell and feature dictionaries are rank-local).
    Returns (cb_strings, cb_ids, umi_keys, feat_strings, feat_ids, scores) with fresh local dictionaries."""
    import torch.distributed as dist
    world, rank = _world(group)
    cb_ids, umi_keys, feat_ids, scores = (np.ascontiguousarray(x, dtype=np.int64) for x in (cb_ids, umi_keys, feat_ids, scores))
    if world == 1:
        return list(cb_strings), cb_ids, umi_keys, list(feat_strings), feat_ids, scores
    dest_of_cb = cell_partition(cb_strings, world)
    dest = dest_of_cb[cb_ids] if cb_ids.size else np.zeros(0, dtype=np.int64)
    outgoing = []
    for d in range(world):
        sel = dest == d
        cbs, cb_inv = np.unique(cb_ids[sel], return_inverse=True)
        fts, ft_inv = np.unique(feat_ids[sel], return_inverse=True)
        outgoing.append({"cb": [cb_strings[i] for i in cbs], "ft": [feat_strings[i] for i in fts], "cb_ids": cb_inv.astype(np.int64),
                         "umi": umi_keys[sel], "ft_ids": ft_inv.astype(np.int64), "scores": scores[sel]})
    incoming = [None] * world
    # (all_to_all of Python objects: one all_gather_object; every rank keeps the slices addressed to it. Volume = the merged rows once.)
    everything = [None] * world
    dist.all_gather_object(everything, outgoing, group=group)
    for src in range(world):
        incoming[src] = everything[src][rank]
    cb_index, ft_index, cb_out, ft_out = {}, {}, [], []
    cols = [[], [], [], []]
    for part in incoming:
        cmap = np.array([cb_index.setdefault(s, len(cb_index)) for s in part["cb"]], dtype=np.int64)
        fmap = np.array([ft_index.setdefault(s, len(ft_index)) for s in part["ft"]], dtype=np.int64)
        if part["cb_ids"].size:
            cols[0].append(cmap[part["cb_ids"]]); cols[1].append(part["umi"]); cols[2].append(fmap[part["ft_ids"]]); cols[3].append(part["scores"])
    cb_out = [None] * len(cb_index)
    for s, i in cb_index.items():
        cb_out[i] = s
    ft_out = [None] * len(ft_index)
    for s, i in ft_index.items():
        ft_out[i] = s
    cat = [np.concatenate(c) if c else np.zeros(0, dtype=np.int64) for c in cols]
    return cb_out, cat[0], cat[1], ft_out, cat[2], cat[3]


def merge_report_outputs(part_paths, output):
    """k-way merge of per-rank `feature<TAB>count<TAB>cell_barcode` files (each sorted by (cell, feature), cells disjoint between
    files) into the order a single `report` run writes (groupby sort on (cell_barcode, feature), __main__.py:289-293)."""
    def rows(path):
        with open(path) as f:
            for line in f:
                feat, _, cell = line.rstrip("\n").split("\t")
                yield (cell, feat), line
    with open(output, "w") as out:
        for _, line in heapq.merge(*[rows(p) for p in part_paths if os.path.exists(p) and os.path.getsize(p)]):
            out.write(line)


def concat_parts_in_batch_order(index_per_rank, part_paths, output, header=b""):
    """Per-read table: rank r wrote the rows of its batches to part_paths[r]; index_per_rank[r] = [(global batch index, bytes)]
    in the order written. The parts are concatenated in global batch order (= the order a single rank would have written)."""
    segs = []
    for r, idx in enumerate(index_per_rank):
        off = 0
        for b, nbytes in idx:
            segs.append((b, r, off, nbytes))
            off += nbytes
    segs.sort()
    files = [open(p, "rb") if os.path.exists(p) else None for p in part_paths]
    with open(output, "wb") as out:
        out.write(header)
        for _, r, off, nbytes in segs:
            files[r].seek(off)
            out.write(files[r].read(nbytes))
    for f in files:
        if f:
            f.close()

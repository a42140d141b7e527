"""Fused per-cell counting: per-read calls -> `feature<TAB>count<TAB>cell_barcode` without the per-read TSV.

The reference runs two commands: `align` writes one row per called read, `report` reads that table back and reduces it
(/root/reference/nimble/__main__.py:254-302). At 10^8 reads the text table is gigabytes and its only consumer is the
reducer, so this module feeds the same reduction (nb_report_rows, the array entry point of the golden-pinned reducer)
directly: identical rows (cell, UMI, feature string) are first merged with their scores summed — on the GPU when the
per-read words still live there (a sort + unique over one 64-bit key per read) — and only the merged rows cross PCIe.
The output text is identical to `report` run on the per-read TSV (tests/test_cli_e2e_gpu.py).
"""
import numpy as np

from .report import report_rows

_UMI_LUT = np.full(256, 4, dtype=np.uint64)          # A C G T -> 0..3, anything else (N) -> 4
for _i, _c in enumerate(b"ACGT"):
    _UMI_LUT[_c] = _i
    _UMI_LUT[_c + 32] = _i


_UMI_LUT_PAD = np.full(256, 7, dtype=np.uint64)     # as _UMI_LUT, but NUL (padding) = 5 and every other byte = 7 (not encodable)
for _i, _c in enumerate(b"ACGTN"):
    _UMI_LUT_PAD[_c] = _i
    _UMI_LUT_PAD[_c + 32] = _i
_UMI_LUT_PAD[0] = 5


def encode_umis(umi):
    """[n, w] 'S1' array (w <= 21) -> uint64 codes, 3 bits per base (N kept distinct, as the raw UMI string is the key)."""
    umi = np.ascontiguousarray(umi)
    n, w = umi.shape
    if w > 21:
        raise ValueError("UMI longer than 21 bases")
    codes = _UMI_LUT[umi.view(np.uint8).reshape(n, w)]
    out = np.zeros(n, dtype=np.uint64)
    for j in range(w):
        out |= codes[:, j] << np.uint64(3 * j)
    return out


def decode_umi(code, w):
    return "".join("ACGTN"[(int(code) >> (3 * j)) & 7] for j in range(w))


def merge_rows(cb_ids, umi_codes, feat_ids, scores):
    """Sum the scores of identical (cb, umi, feature) rows. torch tensors on any device (int64). Returns the merged four
    tensors. Two sort-based uniques: (cb, umi) pairs are dense-ranked first so that the second key fits 64 bits. Ids too
    wide for the packed pair key (more than 2^21 cell ids — raw barcodes without a whitelist — or interned UMI ids beyond
    2^42) are dense-ranked per column first; if even the ranks do not fit, a three-key sort does the merge."""
    import torch
    if cb_ids.numel() == 0:
        return cb_ids, umi_codes, feat_ids, scores
    if int(feat_ids.max()) >= (1 << 24):
        raise ValueError("too many feature strings for the packed key")
    ucb = uumi = None
    if int(umi_codes.max()) >= (1 << 42) or int(cb_ids.max()) >= (1 << 21):
        ucb, cb_ids = torch.unique(cb_ids, return_inverse=True)
        uumi, umi_codes = torch.unique(umi_codes, return_inverse=True)
    if ucb is not None and (ucb.numel() >= (1 << 21) or uumi.numel() >= (1 << 42)):
        c, u, f, summed = _merge_wide(cb_ids, umi_codes, feat_ids, scores)
    else:
        pair = (cb_ids << 42) | umi_codes
        upair, inv = torch.unique(pair, return_inverse=True)
        if upair.numel() >= (1 << 39):
            raise ValueError("too many UMIs for the packed key")
        key = (inv << 24) | feat_ids
        ukey, inv2 = torch.unique(key, return_inverse=True)
        summed = torch.zeros(ukey.numel(), dtype=torch.int64, device=key.device).index_add_(0, inv2, scores)
        p = upair[ukey >> 24]
        c, u, f = p >> 42, p & ((1 << 42) - 1), ukey & ((1 << 24) - 1)
    if ucb is not None:
        c, u = ucb[c], uumi[u]
    return c, u, f, summed


def _merge_wide(cb_rank, umi_rank, feat_ids, scores):
    """merge_rows for dense ranks whose product does not fit the 42-bit pair field: lexicographic sort on three keys (stable
    sorts from the least significant key), then a segmented sum."""
    import torch
    order = torch.argsort(feat_ids, stable=True)
    order = order[torch.argsort(umi_rank[order], stable=True)]
    order = order[torch.argsort(cb_rank[order], stable=True)]
    c, u, f, s = cb_rank[order], umi_rank[order], feat_ids[order], scores[order]
    new = torch.ones(c.numel(), dtype=torch.bool, device=c.device)
    new[1:] = (c[1:] != c[:-1]) | (u[1:] != u[:-1]) | (f[1:] != f[:-1])
    seg = torch.cumsum(new.to(torch.int64), 0) - 1
    summed = torch.zeros(int(seg[-1]) + 1, dtype=torch.int64, device=c.device).index_add_(0, seg, s)
    return c[new], u[new], f[new], summed


class RowAccumulator:
    """Collects merged rows batch by batch; finish() runs the reduction. Cell barcodes are interned to ids on the host."""

    def __init__(self, device=None):
        self.device = device
        self.cb_index = {}
        self.cb_strings = []
        self.umi_index = {}
        self.parts = []

    def umi_keys_of(self, umi):
        """[n, w] 'S1' array -> int64 keys that identify the UMI STRING: 3 bits per character (A C G T N = 0..4, NUL padding = 5)
        when no UMI has more than 21 significant characters and all of them are ACGTN — a function of the content alone, so
        keys from different batches, and from different ranks, agree — else ids interned in this accumulator (rank-local)."""
        umi = np.ascontiguousarray(umi)
        n, w = umi.shape
        u8 = umi.view(np.uint8).reshape(n, w)
        sig = w
        while sig > 0 and not u8[:, sig - 1].any():      # trailing all-NUL columns (BAM tags are NUL padded to a fixed width)
            sig -= 1
        if sig <= 21:
            codes = _UMI_LUT_PAD[u8[:, :sig]]
            if not (codes == 7).any():
                out = np.zeros(n, dtype=np.uint64)
                for j in range(sig):
                    out |= codes[:, j] << np.uint64(3 * j)
                return out.astype(np.int64)
        flat = umi.view(f"S{w}").reshape(n)
        uniq, inv = np.unique(flat, return_inverse=True)
        ids = np.empty(uniq.shape[0], dtype=np.int64)
        for j, u in enumerate(uniq):
            ids[j] = self.umi_index.setdefault(u, len(self.umi_index))
        return ids[inv]

    def cb_ids_of(self, cb):
        """[n, w] 'S1' array (NUL padded allowed) -> int64 ids (order of first appearance)."""
        cb = np.ascontiguousarray(cb)
        n, w = cb.shape
        flat = cb.view(f"S{w}").reshape(n)
        uniq, inv = np.unique(flat, return_inverse=True)
        ids = np.empty(uniq.shape[0], dtype=np.int64)
        for j, u in enumerate(uniq):
            s = u.decode()
            if s not in self.cb_index:
                self.cb_index[s] = len(self.cb_strings)
                self.cb_strings.append(s)
            ids[j] = self.cb_index[s]
        return ids[inv]

    def add(self, cb_ids, umi_codes, feat_ids, scores):
        """Tensors (any device) of the CALLED reads of one batch; rows with feat_id < 0 must already be removed."""
        import torch
        t = [x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x).astype(np.int64)) for x in (cb_ids, umi_codes, feat_ids, scores)]
        if self.device is not None:
            t = [x.to(self.device) for x in t]
        merged = merge_rows(*[x.to(torch.int64) for x in t])
        self.parts.append([x.cpu().numpy() for x in merged])

    def finish(self, cb_strings, feat_strings, output, threshold=0.05, disable_thresholding=False):
        cols = [np.concatenate([p[j] for p in self.parts]) if self.parts else np.zeros(0, dtype=np.int64) for j in range(4)]
        report_rows(cols[0], cols[1], cols[2], cols[3], cb_strings if cb_strings is not None else self.cb_strings, feat_strings, output,
                    threshold, disable_thresholding)
        return cols[0].shape[0]

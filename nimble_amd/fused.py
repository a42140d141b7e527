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
    tensors. Two sort-based uniques: (cb, umi) pairs are dense-ranked first so that the second key fits 64 bits."""
    import torch
    if cb_ids.numel() == 0:
        return cb_ids, umi_codes, feat_ids, scores
    if int(umi_codes.max()) >= (1 << 42) or int(cb_ids.max()) >= (1 << 21):
        raise ValueError("cell id / UMI code out of range for the packed key")
    pair = (cb_ids << 42) | umi_codes
    upair, inv = torch.unique(pair, return_inverse=True)
    if int(feat_ids.max()) >= (1 << 24) or upair.numel() >= (1 << 39):
        raise ValueError("too many feature strings / UMIs for the packed key")
    key = (inv << 24) | feat_ids
    ukey, inv2 = torch.unique(key, return_inverse=True)
    summed = torch.zeros(ukey.numel(), dtype=torch.int64, device=key.device).index_add_(0, inv2, scores)
    p = upair[ukey >> 24]
    return p >> 42, p & ((1 << 42) - 1), ukey & ((1 << 24) - 1), summed


class RowAccumulator:
    """Collects merged rows batch by batch; finish() runs the reduction. Cell barcodes are interned to ids on the host."""

    def __init__(self, device=None):
        self.device = device
        self.cb_index = {}
        self.cb_strings = []
        self.umi_index = {}
        self.parts = []

    def umi_keys_of(self, umi):
        """[n, w] 'S1' array -> int64 keys: the 3-bit code for fixed-width UMIs of <= 21 bases (FASTQ), else interned ids
        (BAM tags: variable width, NUL padded)."""
        umi = np.ascontiguousarray(umi)
        n, w = umi.shape
        if w <= 14 and not (umi.view(np.uint8) == 0).any():
            return encode_umis(umi).astype(np.int64)
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

"""Thin Python objects over the C-ABI (include/nimble_amd.h). PyTorch is used only for device buffers and streams."""
import ctypes as ct

import numpy as np

from . import _lib
from ._lib import NbConfig, NbIndexStats, check

STRAND_FILTERS = {"unstranded": 0, "fiveprime": 1, "threeprime": 2, "none": 3}

REASONS = ["CALLED", "TOO_SHORT", "NO_HIT", "SCORE_THRESHOLD", "SCORE_PERCENT", "EMPTY_EQCLASS", "MULTIPLE_MATCH",
           "INVALID_PAIR", "STRAND", "SCORE_FILTER", "MAX_HITS", "MULTI_HITS"]
NO_CLASS = 0xFFFFFFFF
DEFAULT_K = 20


def make_config(k=DEFAULT_K, score_threshold=20, score_filter=25, score_percent=0.5, num_mismatches=0,
                discard_multiple_matches=False, intersect_level=0, require_valid_pair=False, strand_filter="unstranded"):
    """nb_config with the reference's defaults (/root/reference/nimble/types.py:12-20, __main__.py:399)."""
    if isinstance(strand_filter, str):
        if strand_filter not in STRAND_FILTERS:
            raise ValueError(f"unknown strand_filter {strand_filter!r}; expected one of {sorted(STRAND_FILTERS)}")
        strand_filter = STRAND_FILTERS[strand_filter]
    return NbConfig(int(k), int(score_threshold), int(score_filter), int(num_mismatches), int(bool(discard_multiple_matches)),
                    int(intersect_level), int(bool(require_valid_pair)), int(strand_filter), float(score_percent))


def packed_stride_words(max_len):
    return int(_lib.lib().nb_packed_stride_words(int(max_len)))


def pack_reads(reads, max_len=None):
    """Pack ASCII reads into the device layout. `reads` is a list of str/bytes. Returns (words[n, stride] u64, lens[n] u16)."""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    n = len(bs)
    if max_len is None:
        max_len = max((len(b) for b in bs), default=1) or 1
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(b) for b in bs], out=offs[1:])
    blob = b"".join(bs)
    stride = packed_stride_words(max_len)
    words = np.zeros((n, stride), dtype=np.uint64)
    lens = np.zeros(n, dtype=np.uint16)
    check(_lib.lib().nb_pack_reads(blob, offs.ctypes.data_as(_lib.c_i64p), n, int(max_len),
                                   words.ctypes.data_as(_lib.c_u64p), lens.ctypes.data_as(_lib.c_u16p)))
    return words, lens


def unpack_read(words, length, max_len):
    """Inverse of the packing for one read (row of `words` at the layout of `max_len`): 'ACGT' string, N where masked."""
    nwb = (int(max_len) + 31) // 32
    out = []
    for j in range(int(length)):
        if (int(words[nwb + (j >> 6)]) >> (j & 63)) & 1:
            out.append("N")
        else:
            out.append("ACGT"[(int(words[j >> 5]) >> ((j & 31) * 2)) & 3])
    return "".join(out)


class Index:
    """Coloured k-mer index of one reference library (host build; docs/SPEC.md §2)."""

    def __init__(self, sequences, k=DEFAULT_K):
        seqs = [s.encode() if isinstance(s, str) else bytes(s) for s in sequences]
        self.n_seqs = len(seqs)
        self.k = int(k)
        arr = (ct.c_char_p * len(seqs))(*seqs)
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        h = ct.c_void_p()
        check(_lib.lib().nb_index_build(len(seqs), arr, lens.ctypes.data_as(_lib.c_i64p), self.k, ct.byref(h)))
        self._h = h

    def stats(self):
        s = NbIndexStats()
        check(_lib.lib().nb_index_stats(self._h, ct.byref(s)))
        return {f: getattr(s, f) for f, _ in NbIndexStats._fields_}

    @property
    def words(self):
        return (self.n_seqs + 63) // 64

    def query(self, kmer):
        """Stranded k-mer lookup on the host copy: None or dict(lext, rext, r_internal, l_internal, members)."""
        le, re, ri, li = ct.c_uint8(), ct.c_uint8(), ct.c_uint8(), ct.c_uint8()
        bits = np.zeros(self.words, dtype=np.uint64)
        rc = check(_lib.lib().nb_index_query(self._h, kmer.encode() if isinstance(kmer, str) else kmer, ct.byref(le), ct.byref(re),
                                             ct.byref(ri), ct.byref(li), bits.ctypes.data_as(_lib.c_u64p)))
        if rc == 0:
            return None
        return {"lext": le.value, "rext": re.value, "r_internal": ri.value, "l_internal": li.value, "members": bits_to_members(bits)}

    def filter_maybe(self, seq):
        """Presence filter of the scan kernel for a sequence as given (at least m bases): True = every m-mer of it passes (a k-mer holding
        them may be in the library), False = some m-mer is in no k-mer of the library on either strand, None = no filter in this index."""
        b = seq.encode() if isinstance(seq, str) else seq
        rc = check(_lib.lib().nb_index_filter_maybe(self._h, b, len(b), None, None))
        return None if rc == 2 else bool(rc)

    def filter_params(self):
        """(m, lds_stage_built) of the presence filter, or None."""
        m, l = ct.c_int32(), ct.c_int32()
        rc = check(_lib.lib().nb_index_filter_maybe(self._h, b"", 0, ct.byref(m), ct.byref(l)))
        return None if rc == 2 else (m.value, bool(l.value))

    def anchor_query(self, kmer, run=0):
        """Anchor structures (csrc/nb_internal.h, AEntry) of a stranded k-mer on the host copy: None or dict(tpos, room, members = the AND
        of the classes of the text k-mers at [tpos, tpos + min(run, room)])."""
        tpos, room = ct.c_int64(), ct.c_int32()
        bits = np.zeros(self.words, dtype=np.uint64)
        rc = check(_lib.lib().nb_index_anchor_query(self._h, kmer.encode() if isinstance(kmer, str) else kmer, int(run), ct.byref(tpos), ct.byref(room),
                                                    bits.ctypes.data_as(_lib.c_u64p)))
        if rc == 0:
            return None
        return {"tpos": tpos.value, "room": room.value, "members": bits_to_members(bits) if rc == 1 else None}

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().nb_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def class_features(rows, seq_feature, feature_names_c, n_features, max_hits, multi_hits, name_bytes_max):
    """nb_class_features over a block of class rows -> (feature strings, sorted feature strings, reasons[n] i32)."""
    n, words = rows.shape
    cap = n * 16 + int(np.unpackbits(rows.view(np.uint8)).sum()) * (name_bytes_max + 1)     # every member could name its own feature
    blob, sblob = np.empty(cap, dtype=np.uint8), np.empty(cap, dtype=np.uint8)
    off, soff = np.empty(n + 1, dtype=np.int64), np.empty(n + 1, dtype=np.int64)
    reasons = np.empty(n, dtype=np.int32)
    check(_lib.lib().nb_class_features(n, words, rows.ctypes.data, len(seq_feature), seq_feature.ctypes.data, feature_names_c, n_features, int(max_hits),
                                       int(multi_hits), blob.ctypes.data, cap, off.ctypes.data, sblob.ctypes.data, cap, soff.ctypes.data,
                                       reasons.ctypes.data))
    raw, sraw = blob.tobytes(), sblob.tobytes()
    return ([raw[off[i]:off[i + 1]].decode() for i in range(n)], [sraw[soff[i]:soff[i + 1]].decode() for i in range(n)], reasons)


def bits_to_members(row):
    """u64 bitset row -> sorted list of sequence indices."""
    row = np.ascontiguousarray(row, dtype=np.uint64)
    bits = np.unpackbits(row.view(np.uint8), bitorder="little")
    return np.nonzero(bits)[0].tolist()


class Context:
    """One GPU: holds the uploaded index and the class table; launches the alignment kernels."""

    def __init__(self, device_id=0, max_classes=1 << 20):
        h = ct.c_void_p()
        check(_lib.lib().nb_ctx_create(int(device_id), int(max_classes), ct.byref(h)))
        self._h = h
        self.device_id = int(device_id)
        self.index = None

    def upload(self, index):
        check(_lib.lib().nb_index_upload(self._h, index._h))
        self.index = index

    def align_batch(self, cfg, reads, lens, max_len, out_class=None, out_info=None, out_cov=None, mates=None, mate_lens=None,
                    mate_max_len=0, stream=None, n=None):
        """reads/lens/outputs are torch CUDA tensors (or None). Asynchronous on `stream` (torch stream or None = current)."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(self.device_id)
        sh = ct.c_void_p(stream.cuda_stream)
        if n is None:
            n = reads.shape[0]

        def p(t):
            return ct.c_void_p(t.data_ptr()) if t is not None else None
        check(_lib.lib().nb_align_batch(self._h, ct.byref(cfg), p(reads), p(lens), int(max_len), p(mates), p(mate_lens),
                                        int(mate_max_len), int(n), p(out_class), p(out_info), p(out_cov), sh))

    def last_kernels(self):
        """Names of the kernels the last align_batch launched (dominant first)."""
        return (_lib.lib().nb_last_kernel_names(self._h) or b"").decode().split(";")

    def sync(self, stream=None):
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(self.device_id)
        check(_lib.lib().nb_ctx_sync(self._h, ct.c_void_p(stream.cuda_stream)))

    def classes(self):
        """(rows[n_classes, W] u64, counts[n_classes] i64) in device id order."""
        n = ct.c_int64()
        check(_lib.lib().nb_classes_count(self._h, ct.byref(n)))
        w = self.index.words
        rows = np.empty((n.value, w), dtype=np.uint64)          # (nb_classes_read fills every word)
        counts = np.empty(n.value, dtype=np.int64)
        if n.value:
            check(_lib.lib().nb_classes_read(self._h, n.value, rows.ctypes.data_as(_lib.c_u64p), counts.ctypes.data_as(_lib.c_i64p)))
        return rows, counts

    def classes_device(self, stream=None):
        """(rows [n, words_internal] int64, counts [n] int64) as torch tensors on this GPU: the class table in the engine's INTERNAL
        sequence numbering (equal on every rank that built the same index), copied device to device; rows_to_external() converts."""
        import torch
        n = self.n_classes()
        w = self.index.stats()["words_internal"]
        dev = torch.device("cuda", self.device_id)
        rows = torch.empty((n, w), dtype=torch.int64, device=dev)
        counts = torch.empty(n, dtype=torch.int64, device=dev)
        if stream is None:
            stream = torch.cuda.current_stream(self.device_id)
        check(_lib.lib().nb_classes_copy_device(self._h, n, ct.c_void_p(rows.data_ptr()), ct.c_void_p(counts.data_ptr()), ct.c_void_p(stream.cuda_stream)))
        return rows, counts

    def rows_to_external(self, rows_internal):
        """[n, words_internal] uint64 (host) in internal numbering -> [n, words] uint64 bitsets over the original sequence indices."""
        rows_internal = np.ascontiguousarray(rows_internal, dtype=np.uint64)
        out = np.empty((rows_internal.shape[0], self.index.words), dtype=np.uint64)
        check(_lib.lib().nb_rows_to_external(self._h, rows_internal.ctypes.data_as(_lib.c_u64p), rows_internal.shape[0], out.ctypes.data_as(_lib.c_u64p)))
        return out

    def n_classes(self):
        n = ct.c_int64()
        check(_lib.lib().nb_classes_count(self._h, ct.byref(n)))
        return n.value

    def class_rows(self, first, count):
        """rows[count, W] u64 of classes first .. first + count - 1."""
        rows = np.zeros((count, self.index.words), dtype=np.uint64)
        if count:
            check(_lib.lib().nb_classes_read_range(self._h, int(first), int(count), rows.ctypes.data_as(_lib.c_u64p), None))
        return rows

    def reasons(self):
        out = np.zeros(len(REASONS), dtype=np.int64)
        check(_lib.lib().nb_reasons_read(self._h, out.ctypes.data_as(_lib.c_i64p)))
        return out

    def reset(self, stream=None):
        sh = None
        if stream is not None:
            sh = ct.c_void_p(stream.cuda_stream)
        check(_lib.lib().nb_results_reset(self._h, sh))

    def counts_device(self):
        """(device pointer, capacity) of the u64 per-class count vector, for an RCCL reduce by the caller."""
        ptr, cap = ct.c_void_p(), ct.c_int64()
        check(_lib.lib().nb_counts_device_ptr(self._h, ct.byref(ptr), ct.byref(cap)))
        return ptr.value, cap.value

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().nb_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_info(info):
    """u32 info words -> dict of numpy arrays (score, reason, orientation, mates)."""
    info = np.asarray(info, dtype=np.uint32)
    return {"score": (info & 0xFFFF).astype(np.int32), "reason": ((info >> 16) & 0xFF).astype(np.uint8),
            "orientation": ((info >> 24) & 1).astype(np.uint8), "mates": ((info >> 25) & 3).astype(np.uint8)}


def decode_cov(cov):
    cov = np.asarray(cov, dtype=np.uint64)
    return np.stack([(cov >> np.uint64(16 * j)) & np.uint64(0xFFFF) for j in range(4)], axis=1).astype(np.int32)

"""`nimble align` host path: the drop-in for the subprocess call at /root/reference/nimble/__main__.py:195.

align() keeps the reference's signature and return convention (__main__.py:153-211): it returns the exit code the CLI
passes to sys.exit, 0 on success. Instead of building an argv and spawning the Rust binary, it loads each library JSON,
builds the index, streams the FASTQ input in batches through the C-ABI (nb_fastq_next_batch -> nb_align_batch) and
writes one TSV per library at append_path_string(output, "." + stem) when several libraries are given
(__main__.py:182-189). Output shapes: docs/SPEC.md §5.

Input modes
  * one FASTQ(.gz): single-end bulk -> "nimble_features<TAB>nimble_score" per distinct feature list.
  * two FASTQ(.gz) + tenx options: R1 carries CB+UMI (fastq_barcode_processor.py:131-141), R2 is aligned; one row per
    called read with r1_CB / r1_UB — what the reference obtains via `fastq-to-bam` + BAM input.
  * two FASTQ(.gz) without tenx options: a biological pair (needs the paired kernel).
  * one BAM (single-end or pairs in any record order, with CB/UB tags): read with the built-in BAM reader
    (csrc/nb_bam.cpp), one row per called template; the reference's `samtools sort -t UR -n` staging (sort_input_bam,
    __main__.py:312-370) is unnecessary because `report` groups by (cell, UMI) itself, so nothing is sorted or deleted.
"""
import ctypes as ct
import gzip
import os
import sys
import time

import numpy as np

from . import _lib, engine, fused
from .library import Library, default_config
from .utils import append_path_string, parse_trim

BATCH_READS = int(os.environ.get("NB_BATCH_READS", 1 << 20))   # templates per GPU launch
MAX_READ_LEN = 339      # NB_MAX_POSITIONS + k - 1 at k = 20
COV_COLUMNS = ["r1_forward_score", "r1_reverse_score", "r2_forward_score", "r2_reverse_score"]


class TenxOptions:
    """R1 = cell barcode + UMI carrier (defaults of the reference's fastq-to-bam: __main__.py:430-431)."""

    def __init__(self, whitelist=None, cb_length=16, umi_length=12):
        self.whitelist, self.cb_length, self.umi_length = whitelist, int(cb_length), int(umi_length)


def _host_buffer(shape, dtype, pinned):
    """numpy array for reader output; page-locked (through torch) when a GPU is there, so that the H2D copy is a plain DMA.
    Allocated once per reader and reused: first touch of fresh pages costs more than parsing into them."""
    if pinned:
        import torch
        t = torch.empty(int(np.prod(shape)) * np.dtype(dtype).itemsize, dtype=torch.uint8, pin_memory=True)
        return t.numpy().view(dtype).reshape(shape)
    return np.empty(shape, dtype=dtype)


class FastqReader:
    """nb_fastq_* (csrc/nb_ingest.cpp). Batches alternate between two buffer sets: a returned batch stays valid until the
    second next_batch() call after it (align() parses batch i+1 while batch i is on the GPU). A call with scratch=True
    (the top-up of a batch that ended early, _top_up) reads into a third, private buffer set and does not advance the
    turn, so it can never land in the buffers of the batch that is still being uploaded."""

    def __init__(self, path, threads=1, pinned=False, byte_range=None):
        """byte_range = (begin, end) of a plain FASTQ file, begin at a record boundary: the share of one rank (plan_ranges)."""
        self._h = ct.c_void_p()
        if byte_range is None:
            _lib.check(_lib.lib().nb_fastq_open(path.encode(), ct.byref(self._h)))
        else:
            _lib.check(_lib.lib().nb_fastq_open_range(path.encode(), int(byte_range[0]), int(byte_range[1]), ct.byref(self._h)))
        _lib.check(_lib.lib().nb_fastq_set_threads(self._h, int(threads)))
        self._pinned = pinned
        self._sets, self._turn = {}, 0

    def _buffers(self, kind, max_reads, shape_tail, dtype, scratch=False):
        key = (kind, "scratch" if scratch else self._turn & 1)
        buf = self._sets.get(key)
        if buf is None or buf.shape[0] < max_reads or buf.shape[1:] != tuple(shape_tail):
            buf = self._sets[key] = _host_buffer((max_reads,) + tuple(shape_tail), dtype, self._pinned)
        return buf

    def next_batch(self, max_reads, max_len=None, barcodes_only=False, cb_len=0, umi_len=0, scratch=False):
        """Returns dict(n, words, lens, max_len, cb, umi, cbq). The reads are packed at the layout of the batch's own longest
        read (`max_len` in the result); the argument `max_len` is the longest read accepted. scratch=True: private buffers,
        valid until the next scratch call only (the caller copies)."""
        n, batch_len = ct.c_int64(), ct.c_int32()
        L = _lib.lib()
        if not scratch:
            self._turn += 1
        if barcodes_only:
            cb = self._buffers("cb", max_reads, (cb_len,), "S1", scratch)
            umi = self._buffers("umi", max_reads, (umi_len,), "S1", scratch)
            cbq = self._buffers("cbq", max_reads, (cb_len,), "S1", scratch)
            _lib.check(L.nb_fastq_next_batch_tight(self._h, max_reads, 1, None, None, ct.byref(n), ct.byref(batch_len), cb.ctypes.data, cb_len,
                                                   umi.ctypes.data, umi_len, cbq.ctypes.data))
            k = n.value
            return {"n": k, "cb": cb[:k], "umi": umi[:k], "cbq": cbq[:k]}
        flat = self._buffers("words", max_reads, (engine.packed_stride_words(max_len),), np.uint64, scratch).reshape(-1)
        lens = self._buffers("lens", max_reads, (), np.uint16, scratch)
        _lib.check(L.nb_fastq_next_batch_tight(self._h, max_reads, max_len, flat.ctypes.data, lens.ctypes.data, ct.byref(n), ct.byref(batch_len),
                                               None, 0, None, 0, None))
        k, stride = n.value, engine.packed_stride_words(max(batch_len.value, 1))
        return {"n": k, "words": flat[:k * stride].reshape(k, stride), "lens": lens[:k], "max_len": max(batch_len.value, 1)}

    def close(self):
        if self._h:
            _lib.lib().nb_fastq_close(self._h)
            self._h = None


class BamReader:
    """nb_bam_* (csrc/nb_bam.cpp): templates of a BAM in file order, CB/UB tags of r1. No sort needed (see nb_bam.cpp).
    Buffers alternate between two sets, as in FastqReader."""
    TAG_WIDTH = 48

    def __init__(self, path, threads=1, pinned=False):
        self._h = ct.c_void_p()
        _lib.check(_lib.lib().nb_bam_open(path.encode(), ct.byref(self._h)))
        _lib.check(_lib.lib().nb_bam_set_threads(self._h, int(threads)))
        self._pinned = pinned
        self._sets, self._turn = {}, 0
        self._has_mates = False

    _buffers = FastqReader._buffers

    def next_batch(self, max_reads, max_len):
        """dict(n, paired, words, lens, max_len, mate_words, mate_lens, mate_max_len, cb, umi); `max_len` = longest read accepted,
        the result's max_len / mate_max_len = the layouts the batch is packed at."""
        self._turn += 1
        cap = engine.packed_stride_words(max_len)
        f1 = self._buffers("w1", max_reads, (cap,), np.uint64).reshape(-1); l1 = self._buffers("l1", max_reads, (), np.uint16)
        cb = self._buffers("cb", max_reads, (self.TAG_WIDTH,), "S1"); ub = self._buffers("ub", max_reads, (self.TAG_WIDTH,), "S1")
        n, paired, len1, len2 = ct.c_int64(), ct.c_int32(), ct.c_int32(), ct.c_int32()
        while True:
            f2 = l2 = None
            if self._has_mates:     # mate buffers only once the file has shown pairs (most 10x BAMs are single-end)
                f2 = self._buffers("w2", max_reads, (cap,), np.uint64).reshape(-1); l2 = self._buffers("l2", max_reads, (), np.uint16)
            try:
                _lib.check(_lib.lib().nb_bam_next_batch_tight(self._h, max_reads, max_len, f1.ctypes.data, l1.ctypes.data,
                                                              f2.ctypes.data if f2 is not None else None, l2.ctypes.data if l2 is not None else None,
                                                              cb.ctypes.data, ub.ctypes.data, self.TAG_WIDTH, ct.byref(n), ct.byref(len1), ct.byref(len2),
                                                              ct.byref(paired)))
                break
            except _lib.NimbleError as e:
                if self._has_mates or "no mate buffers" not in str(e):
                    raise
                self._has_mates = True   # (the gathered templates stay pending in the reader: nothing is lost)
        k, s1, s2 = n.value, engine.packed_stride_words(len1.value), engine.packed_stride_words(len2.value)
        out = {"n": k, "paired": bool(paired.value), "words": f1[:k * s1].reshape(k, s1), "lens": l1[:k], "max_len": len1.value,
               "mate_words": None, "mate_lens": None, "mate_max_len": len2.value, "cb": cb[:k], "umi": ub[:k]}
        if f2 is not None:
            out.update(mate_words=f2[:k * s2].reshape(k, s2), mate_lens=l2[:k])
        return out

    def close(self):
        if self._h:
            _lib.lib().nb_bam_close(self._h)
            self._h = None


class LibraryRun:
    """One reference library on one GPU: index, context, class -> feature-string cache, output writer."""

    def __init__(self, lib, out_path, strand_filter, device_id, per_read, world=1, rank=0, ctx=None):
        """`ctx` (tests of the multi-rank merge layer): an object with the class-table interface of engine.Context (n_classes,
        class_rows, classes, reasons) standing in for the GPU context; by default the index is built and uploaded to `device_id`."""
        self.world, self.rank = world, rank
        self.lib = lib if isinstance(lib, Library) else Library(lib)
        c = self.lib.config
        self.cfg = engine.make_config(k=self.lib.k, score_threshold=c["score_threshold"], score_filter=c["score_filter"],
                                      score_percent=c["score_percent"], num_mismatches=c["num_mismatches"],
                                      discard_multiple_matches=c["discard_multiple_matches"], intersect_level=c["intersect_level"],
                                      require_valid_pair=c["require_valid_pair"], strand_filter=strand_filter)
        self.max_hits = int(c["max_hits_to_report"])
        self.multi_hits = int(c["discard_multi_hits"])
        if ctx is None:
            self.index = engine.Index(self.lib.sequences, self.lib.k)
            self.ctx = engine.Context(device_id, max_classes=1 << 22)
            self.ctx.upload(self.index)
        else:
            self.index, self.ctx = None, ctx
        self.out_path = out_path
        self.per_read = per_read
        self._feature_names = sorted(set(self.lib.features))
        fidx = {f: i for i, f in enumerate(self._feature_names)}
        self._seq_feature = np.array([fidx[f] for f in self.lib.features], dtype=np.int32)
        enc = [f.encode() for f in self._feature_names]
        self._feature_names_c = (ct.c_char_p * len(enc))(*enc)
        self._name_bytes_max = max(len(e) for e in enc)
        self.class_feature = []     # class id -> feature string, or None when a class-level filter rejects it
        self.class_reason = []
        self.class_feat_id = []     # class id -> index into feat_strings (sorted, comma-joined features), -1 if rejected
        self.feat_index, self.feat_strings = {}, []
        self.rows = None            # fused per-cell counting (RowAccumulator), when requested
        self.bulk = {}
        self.n_rows = 0
        self.fh = None
        self._snap = None
        self._row_buf = None
        self._gz_buf = None
        self.gz_threads = 1
        self.part_index = []          # several ranks: (global batch index, bytes) of every segment this rank wrote to its part file
        if per_read:
            self.gz = out_path.endswith(".gz")          # gzip members written by nb_gzip_members (parallel), see write_text
            header = np.frombuffer(("\t".join(["nimble_features", "nimble_score", "r1_CB", "r1_UB"] + COV_COLUMNS) + "\n").encode(), dtype=np.uint8)
            if world == 1:
                self.fh = open(out_path, "wb")
                self.write_text(header)
            else:      # every rank writes the rows of ITS batches to a part file; rank 0 concatenates them in batch order (merge.py)
                self.part_path = f"{out_path}.rank{rank}.part"
                self.fh = open(self.part_path, "wb")
                self.header_bytes = self.encode_text(header)

    def write_text(self, buf, batch_index=None):
        """Append TSV text (uint8 array) to the per-read output, gzip-compressed when the output name ends in .gz."""
        data = self.encode_text(buf)
        self.fh.write(data)
        if batch_index is not None and self.world > 1:
            self.part_index.append((int(batch_index), len(data)))

    def encode_text(self, buf):
        """The bytes write_text appends for `buf`: the text itself, or its BGZF members for a .gz output."""
        if not self.gz:
            return memoryview(np.ascontiguousarray(buf)).tobytes() if self.world > 1 else memoryview(buf)
        n = int(buf.shape[0])
        cap = n + (n >> 9) + 64 * ((n >> 20) + 1) + 64
        if self._gz_buf is None or self._gz_buf.shape[0] < cap:
            self._gz_buf = np.empty(cap, dtype=np.uint8)
        out_len = ct.c_int64()
        buf = np.ascontiguousarray(buf)
        _lib.check(_lib.lib().nb_gzip_members(buf.ctypes.data, n, 4, self.gz_threads, self._gz_buf.ctypes.data, cap, ct.byref(out_len)))
        return memoryview(self._gz_buf[:out_len.value]).tobytes() if self.world > 1 else memoryview(self._gz_buf[:out_len.value])

    def refresh_classes(self):
        """Map newly interned classes to feature strings (nb_class_features: group_on naming + the feature-level filters of
        docs/SPEC.md §4). Call with the GPU idle."""
        known, n = len(self.class_feature), self.ctx.n_classes()
        if n == known:
            return
        rows = self.ctx.class_rows(known, n - known)
        feats, sorted_feats, reasons = engine.class_features(rows, self._seq_feature, self._feature_names_c, len(self._feature_names),
                                                             self.max_hits, self.multi_hits, self._name_bytes_max)
        for f, key, reason in zip(feats, sorted_feats, reasons.tolist()):
            self.class_feature.append(f if reason == 0 else None)
            self.class_reason.append(reason)
            fid = -1
            if reason == 0:                                # report sorts the features of a row (__main__.py:248)
                fid = self.feat_index.setdefault(key, len(self.feat_strings))
                if fid == len(self.feat_strings):
                    self.feat_strings.append(key)
            self.class_feat_id.append(fid)

    def snapshot(self):
        """(ctypes array of class feature strings, its length, longest string, class -> feature-string id) for the classes
        known now; rebuilt only when classes were added. Call with the GPU idle (reads the class table)."""
        self.refresh_classes()
        feats = self.class_feature
        if self._snap is None or self._snap[1] != len(feats):
            arr = (ct.c_char_p * max(1, len(feats)))(*[f.encode() if f is not None else None for f in feats])
            self._snap = (arr, len(feats), max((len(f) for f in feats if f is not None), default=0),
                          np.asarray(self.class_feat_id, dtype=np.int64))
        return self._snap

    def features_of_rows(self, rows):
        """class bitset rows -> feature string per row, or None where a feature-count filter rejects the class (nb_class_features)."""
        if rows.shape[0] == 0:
            return []
        feats, _, reasons = engine.class_features(rows, self._seq_feature, self._feature_names_c, len(self._feature_names), self.max_hits,
                                                  self.multi_hits, self._name_bytes_max)
        return [f if r == 0 else None for f, r in zip(feats, reasons.tolist())]

    def finish(self, cdev=None):
        """End of job. Several ranks (merge.py): per-read parts are concatenated by rank 0 in global batch order; bulk counts are
        reduced with the digest-first dictionary agreement + one all-reduce and written by rank 0."""
        if self.per_read:
            self.fh.close()
            if self.world > 1:
                import torch.distributed as dist
                from . import merge
                index = [None] * self.world
                dist.all_gather_object(index, self.part_index)
                if self.rank == 0:
                    merge.concat_parts_in_batch_order(index, [f"{self.out_path}.rank{r}.part" for r in range(self.world)], self.out_path, self.header_bytes)
                dist.barrier()
                os.remove(self.part_path)
            return
        if self.world > 1:
            import torch
            from . import merge
            gpu = getattr(self.ctx, "device_id", None)                      # (the gloo tests drive this layer with a host stand-in for the ctx)
            if gpu is not None and torch.cuda.is_available():               # the table stays on the device until the merged rows reach the root
                rows, counts = self.ctx.classes_device()
                rows, counts = merge.reduce_class_counts_hashed(rows, counts, device=cdev, compute_device=torch.device("cuda", gpu))
                if self.rank == 0:
                    rows = self.ctx.rows_to_external(rows)
            else:
                rows, counts = self.ctx.classes()
                rows, counts = merge.reduce_class_counts_hashed(rows, counts, device=cdev)
            if self.rank != 0:
                return
            feats = self.features_of_rows(rows)
        else:
            self.refresh_classes()
            _, counts = self.ctx.classes()
            feats = self.class_feature
        for f, cnt in zip(feats, counts):
            if f is not None and cnt:
                self.bulk[f] = self.bulk.get(f, 0) + int(cnt)
        write_bulk_table(self.out_path, self.bulk)
        self.n_rows = len(self.bulk)


def write_bulk_table(out_path, bulk):
    """The legacy bulk shape (docs/SPEC.md §5, nimble/parse.py:39-57): header, then one row per distinct feature list, sorted."""
    opener = gzip.open if out_path.endswith(".gz") else open
    with opener(out_path, "wt") as fh:
        fh.write("nimble_features\tnimble_score\n")
        for f in sorted(bulk):
            fh.write(f"{f}\t{bulk[f]}\n")


class NativeWhitelist:
    """C++ whitelist corrector (csrc/nb_host.cpp); same rule as the Python twin tests/barcodes_twin.py, batch at a time."""

    def __init__(self, path, threads=1):
        self._h = ct.c_void_p()
        _lib.check(_lib.lib().nb_whitelist_load(path.encode(), ct.byref(self._h)))
        _lib.check(_lib.lib().nb_whitelist_set_threads(self._h, int(threads)))
        self.stats = {"cb_perfect_match": 0, "cb_corrected": 0, "cb_no_correction": 0}

    def correct(self, cb, cbq):
        """cb, cbq: [n, cb_len] 'S1' arrays -> (corrected [n, cb_len] 'S1', ok [n] uint8)."""
        n, cb_len = cb.shape
        cb, cbq = np.ascontiguousarray(cb), np.ascontiguousarray(cbq)
        out, ok = np.empty_like(cb), np.zeros(n, dtype=np.uint8)
        st = (ct.c_int64 * 3)()
        _lib.check(_lib.lib().nb_whitelist_correct(self._h, cb.ctypes.data, cbq.ctypes.data, n, cb_len, out.ctypes.data, ok.ctypes.data, st))
        self.stats = {"cb_perfect_match": st[0], "cb_corrected": st[1], "cb_no_correction": st[2]}
        return out, ok

    def close(self):
        if self._h:
            _lib.lib().nb_whitelist_destroy(self._h)
            self._h = None


def _write_per_read(run, snap, cls, info_raw, cov_raw, cb, cb_ok, umi, batch_index=None):
    """Format this batch's rows in C++ (nb_format_tenx_rows) and append them to the run's TSV. `snap` = the run's class ->
    feature tables as of this batch (LibraryRun.snapshot(), taken on the main thread while the GPU was idle)."""
    feat_arr, n_feats, feat_max, cfid = snap
    n = cls.shape[0]
    cb, umi = np.ascontiguousarray(cb), np.ascontiguousarray(umi)
    cap = n * (feat_max + cb.shape[1] + umi.shape[1] + 64) + 64
    if run._row_buf is None or run._row_buf.shape[0] < cap:
        run._row_buf = np.empty(cap, dtype=np.uint8)
    buf = run._row_buf
    out_len, n_rows = ct.c_int64(), ct.c_int64()
    _lib.check(_lib.lib().nb_format_tenx_rows(n, cls.ctypes.data, info_raw.ctypes.data, cov_raw.ctypes.data, feat_arr, n_feats,
                                              cb.ctypes.data, cb.shape[1], cb_ok.ctypes.data if cb_ok is not None else None, umi.ctypes.data,
                                              umi.shape[1], buf.ctypes.data, cap, ct.byref(out_len), ct.byref(n_rows)))
    run.write_text(buf[:out_len.value], batch_index)
    run.n_rows += n_rows.value
    if run.rows is not None:       # the same rows, as arrays, for the fused per-cell counting
        ok = ((info_raw >> 16) & 0xFF) == 0
        ok &= cls != engine.NO_CLASS
        fid = np.full(n, -1, dtype=np.int64)
        fid[ok] = cfid[cls[ok]]
        ok &= fid >= 0
        if cb_ok is not None:
            ok &= cb_ok != 0
        ok &= (cb.view(np.uint8).reshape(n, -1)[:, 0] != 0) & (umi.view(np.uint8).reshape(n, -1)[:, 0] != 0)   # null CB / UB rows
        if ok.any():
            run.rows.add(run.rows.cb_ids_of(cb[ok]), run.rows.umi_keys_of(umi[ok]), fid[ok], ((info_raw[ok] >> 25) & 3).astype(np.int64))


def align(reference, output, input, num_cores=1, strand_filter="unstranded", trim="", tmpdir=None, tenx=None, device_id=0,
          batch_reads=BATCH_READS, report_output=None, report_threshold=0.05, report_disable_thresholding=False):
    """Same positional arguments and return value as the reference's align() (__main__.py:153). `tenx`, `device_id`,
    `batch_reads` and `report_output` (fused per-cell counting: also write what `report` would produce from the per-read
    table, without re-reading it; one file per library, named like the outputs) are extensions with neutral defaults."""
    try:
        import torch
        print("Aligning input data to the reference libraries")
        sys.stdout.flush()
        is_bam = os.path.splitext(input[0])[-1].lower() == ".bam"
        if is_bam and len(input) != 1:
            raise ValueError("--input takes exactly one BAM file")
        if len(input) > 2:
            raise ValueError("--input takes one or two FASTQ files")
        if strand_filter not in engine.STRAND_FILTERS:
            raise ValueError(f"--strand_filter must be one of {sorted(engine.STRAND_FILTERS)}")
        libs = reference.split(",")
        if parse_trim(trim, len(libs)) is not None:
            _warn_not_honoured(f"--trim {trim!r}: reads are NOT trimmed by this engine (the rule lives in the Rust aligner only and is "
                               "specified nowhere in the reference tree, docs/SPEC.md §5/§7); results are those of untrimmed reads")
        loaded = [Library(lib) for lib in libs]
        for lib in loaded:
            _warn_library_options(lib)
        if not torch.cuda.is_available():
            raise RuntimeError("no MI355X/HIP device visible: nimble_amd has no CPU alignment path")
        world, rank, device_id, cdev, own_group = _dist_setup(device_id)
        dev = torch.device("cuda", device_id)
        per_read = tenx is not None or is_bam      # BAM input carries CB/UB tags: one row per called template, as the reference
        if tenx is not None and is_bam:
            raise ValueError("--tenx applies to FASTQ input; a BAM already carries its CB/UB tags")
        if per_read and not is_bam and len(input) != 2:
            raise ValueError("tenx mode needs --input R1.fastq R2.fastq")
        paired = len(input) == 2 and not per_read   # (a paired BAM is detected from its flags, batch by batch)
        runs = []
        for lib, loaded_lib in zip(libs, loaded):
            suffix = "." + os.path.splitext(os.path.basename(lib))[0] if len(libs) > 1 else ""
            runs.append(LibraryRun(loaded_lib, append_path_string(output, suffix), strand_filter, device_id, per_read, world, rank))
            runs[-1].gz_threads = max(1, int(num_cores))
            if report_output:
                if not per_read:
                    raise ValueError("report_output needs barcoded input (BAM or --tenx)")
                runs[-1].rows = fused.RowAccumulator()
                runs[-1].report_path = append_path_string(report_output, suffix)
        corrector = None
        if tenx is not None and tenx.whitelist:
            corrector = NativeWhitelist(tenx.whitelist, num_cores)
        # several ranks: each rank reads its own byte range of plain FASTQ input (plan_ranges); compressed / BAM input is read in full by
        # every rank and the batches are dealt round-robin instead
        plan = plan_ranges(input, world, rank) if world > 1 and not is_bam and os.environ.get("NB_RANGE_SHARDING", "1") != "0" else None
        ranges = plan["ranges"] if plan else None
        bam = BamReader(input[0], num_cores, pinned=True) if is_bam else None
        r1 = FastqReader(input[0], num_cores, pinned=True, byte_range=ranges[0] if ranges else None) if not is_bam else None
        r2 = FastqReader(input[1], num_cores, pinned=True, byte_range=ranges[1] if ranges else None) if len(input) == 2 else None
        total, t0 = 0, time.time()
        prof = _StageClock()

        def fetch():
            """Next batch of templates from the input files, barcodes corrected (runs in a helper thread: the C++ readers
            release the GIL, so parsing batch i+1 overlaps the GPU work and the TSV writing of batch i)."""
            t = time.perf_counter()
            if is_bam:
                bb = bam.next_batch(batch_reads, MAX_READ_LEN)
                b1 = {"cb": bb["cb"], "umi": bb["umi"], "cbq": None}
                seq = {"n": bb["n"], "words": bb["words"], "lens": bb["lens"], "max_len": bb["max_len"]}
                mate = {"n": bb["n"], "words": bb["mate_words"], "lens": bb["mate_lens"], "max_len": bb["mate_max_len"]} if bb["paired"] else None
            elif per_read:
                # R1 (barcodes) and R2 (cDNA) are parsed side by side; a reader may end a batch early, the other is then topped up
                f2 = side.submit(r2.next_batch, batch_reads, MAX_READ_LEN)
                b1 = r1.next_batch(batch_reads, barcodes_only=True, cb_len=tenx.cb_length, umi_len=tenx.umi_length)
                seq, mate = f2.result(), None
                if seq["n"] < b1["n"]:
                    seq = _top_up(r2, seq, b1["n"])
                elif seq["n"] > b1["n"]:
                    raise ValueError("R1 and R2 FASTQ files have different numbers of records")
            elif paired:
                b1 = None
                f2 = side.submit(r2.next_batch, batch_reads, MAX_READ_LEN)
                seq = r1.next_batch(batch_reads, MAX_READ_LEN)
                mate = f2.result()
                if mate["n"] < seq["n"]:
                    mate = _top_up(r2, mate, seq["n"])
                elif seq["n"] < mate["n"]:
                    seq = _top_up(r1, seq, mate["n"])
            else:
                b1, mate = None, None
                seq = r1.next_batch(batch_reads, MAX_READ_LEN)
            if per_read and seq["n"]:
                b1["cb_fixed"], b1["cb_ok"] = corrector.correct(b1["cb"], b1["cbq"]) if corrector else (b1["cb"].copy(), None)
            prof.add("ingest (helper thread)", time.perf_counter() - t)
            return seq, mate, b1

        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=1)
        side = ThreadPoolExecutor(max_workers=1)
        writer = ThreadPoolExecutor(max_workers=1)
        write_jobs = []
        pending = pool.submit(fetch)
        host_out = [None, None]          # two sets of pinned result buffers: the writer thread reads one while the GPU fills the other
        turn = 0
        batch_index = -1
        while True:
            with prof("wait for input"):
                seq, mate, b1 = pending.result()
            n = seq["n"]
            if n == 0:
                break
            pending = pool.submit(fetch)
            batch_index += 1
            if ranges is None and batch_index % world != rank:   # several ranks without byte ranges: batches are dealt round-robin (every rank
                total += n                                         # walks the whole input and aligns its share)
                continue
            part_key = batch_index if ranges is None else (rank << 32) + batch_index   # order of the parts in the per-read table = input order
            with prof("H2D"):
                words, max_len = _tight(seq)
                d_reads = torch.from_numpy(words.view(np.int64)).to(dev, non_blocking=True)
                d_lens = torch.from_numpy(seq["lens"].view(np.int16)).to(dev, non_blocking=True)
                d_mates = d_mlens = None
                mmax = 0
                if mate is not None:
                    mw, mmax = _tight(mate)
                    d_mates = torch.from_numpy(mw.view(np.int64)).to(dev, non_blocking=True)
                    d_mlens = torch.from_numpy(mate["lens"].view(np.int16)).to(dev, non_blocking=True)
                out_class = torch.empty(n, dtype=torch.int32, device=dev)
                out_info = torch.empty(n, dtype=torch.int32, device=dev)
                out_cov = torch.empty(n, dtype=torch.int64, device=dev) if per_read else None
            for run in runs:
                with prof("align kernel + sync"):
                    run.ctx.align_batch(run.cfg, d_reads, d_lens, max_len, out_class, out_info, out_cov, mates=d_mates, mate_lens=d_mlens,
                                        mate_max_len=mmax, n=n)
                    run.ctx.sync()
                if per_read:
                    with prof("wait for writer"):
                        while len(write_jobs) >= 2:
                            write_jobs.pop(0).result()
                    with prof("D2H"):
                        if host_out[turn] is None or host_out[turn][0].shape[0] < n:
                            cap = max(n, batch_reads)
                            host_out[turn] = (torch.empty(cap, dtype=torch.int32, pin_memory=True), torch.empty(cap, dtype=torch.int32, pin_memory=True),
                                              torch.empty(cap, dtype=torch.int64, pin_memory=True))
                        h_class, h_info, h_cov = (t[:n] for t in host_out[turn])
                        h_class.copy_(out_class); h_info.copy_(out_info); h_cov.copy_(out_cov)
                        turn ^= 1
                    with prof("class snapshot"):
                        snap = run.snapshot()
                    write_jobs.append(writer.submit(_timed, prof, "TSV rows (writer thread)", _write_per_read, run, snap, h_class.numpy().view(np.uint32),
                                                    h_info.numpy().view(np.uint32), h_cov.numpy().view(np.uint64), b1["cb_fixed"], b1["cb_ok"],
                                                    b1["umi"].copy(), part_key))   # (reader buffers are recycled two batches on)
            total += n
        with prof("wait for writer"):
            for j in write_jobs:
                j.result()
        writer.shutdown(wait=True)
        if plan is not None:
            import torch.distributed as dist
            check_range_complete(plan, total)
            t = torch.tensor([total], dtype=torch.int64, device=cdev)
            dist.all_reduce(t)
            total = int(t[0])                 # templates of the whole job, as every rank counts them without byte ranges
        for run in runs:
            run.finish(cdev)
            reasons = run.ctx.reasons()
            n_rows = run.n_rows
            if world > 1:
                import torch.distributed as dist
                t = torch.tensor(list(reasons) + [n_rows if run.per_read else 0], dtype=torch.int64, device=cdev)
                dist.all_reduce(t)
                reasons = t[:-1].cpu().numpy()
                n_rows = int(t[-1]) if run.per_read else n_rows
            if run.rows is not None:
                nrows = _finish_fused(run, world, rank, report_threshold, report_disable_thresholding)
                if rank == 0:
                    print(f"{run.lib.stem}: per-cell counts from {nrows} merged rows -> {run.report_path}")
            if rank == 0:
                print(f"{run.lib.stem}: {total} templates, {int(reasons[0])} called, {n_rows} output rows -> {run.out_path}")
                print("  reasons: " + ", ".join(f"{engine.REASONS[i]}={int(v)}" for i, v in enumerate(reasons) if v))
        if corrector:
            print(f"cell barcodes: {corrector.stats}")
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            if own_group:
                dist.destroy_process_group()
        print(f"Aligned {total} templates in {time.time() - t0:.2f} s" + (f" (rank {rank} of {world})" if world > 1 else ""))
        if os.environ.get("NB_PROFILE"):
            print("  stages: " + prof.report())
        pool.shutdown(wait=True)
        side.shutdown(wait=True)
        for rd in (r1, r2, bam):
            if rd:
                rd.close()
        print("Retaining all input files.")
        return 0
    except Exception as e:  # exit-code convention of the process boundary (__main__.py:198,211)
        print(f"Error -- alignment failed: {type(e).__name__}: {e}", file=sys.stderr)
        return 1


def _warn_not_honoured(msg):
    """One line on stderr for every option the reference's aligner honours and this engine does not (a drop-in must not
    accept such an option silently; the exit code stays 0 because the run itself is valid for untrimmed / unfiltered input)."""
    print(f"Warning -- not honoured: {msg}", file=sys.stderr)
    sys.stderr.flush()


def plan_ranges(paths, world, rank):
    """Several ranks, plain FASTQ input: the byte range of every input file that THIS rank reads, so that each rank parses 1/world of
    the text instead of all of it. Every rank counts the newlines of one slice of each file (nb_file_count_newlines), the counts are
    all-gathered, records (4 lines) are dealt in contiguous blocks — the same record numbers in R1 and R2 — and each rank locates the byte
    offsets of its first and last line inside one slice (nb_file_nth_newline). Returns {"ranges": [(begin, end), ...], "records": the
    number of records this rank must find in each range (check_range_complete)} or None when the input cannot be entered in the middle
    (gzip / BGZF / BAM), is not four lines per record or the files disagree: the ranks then deal batches round-robin."""
    import torch.distributed as dist
    L = _lib.lib()
    mine, ok = [], True
    for path in paths:
        size = os.path.getsize(path)
        with open(path, "rb") as f:
            magic = f.read(2)
        if magic == b"\x1f\x8b" or path.lower().endswith(".bam"):
            ok = False
            break
        lo, hi = size * rank // world, size * (rank + 1) // world
        cnt, last = ct.c_int64(), ct.c_int32()
        _lib.check(L.nb_file_count_newlines(path.encode(), lo, hi, ct.byref(cnt), ct.byref(last)))
        mine.append((size, cnt.value, last.value))
    everyone = [None] * world
    dist.all_gather_object(everyone, (ok, mine))
    if not all(e[0] for e in everyone):
        return None
    ranges, n_records = [], None
    for i, path in enumerate(paths):
        size = mine[i][0]
        counts = [e[1][i][1] for e in everyone]
        last_byte = next((e[1][i][2] for e in reversed(everyone) if e[1][i][2] >= 0), -1)
        lines = sum(counts) + (1 if size > 0 and last_byte != 10 else 0)
        if lines % 4:
            ranges = None
            break
        recs = lines // 4
        if n_records is None:
            n_records = recs
        elif recs != n_records:                        # blank lines, or really different: the whole-file readers find out which
            ranges = None
            break

        def line_offset(l):
            """byte offset at which line l (0-based) starts"""
            if l <= 0:
                return 0
            if l >= lines:
                return size
            acc = 0
            for s_, c in enumerate(counts):
                if acc + c >= l:                       # the l-th newline of the file is the (l - acc)-th of slice s_
                    off = ct.c_int64()
                    _lib.check(L.nb_file_nth_newline(path.encode(), size * s_ // world, l - acc, ct.byref(off)))
                    return off.value
                acc += c
            return size
        ranges.append((line_offset(4 * (recs * rank // world)), line_offset(4 * (recs * (rank + 1) // world))))
    good = ranges is not None
    if good:                                           # strict framing at the range starts: '@' line, sequence, '+' line
        for path, (b, e) in zip(paths, ranges):
            if e > b:
                with open(path, "rb") as f:
                    f.seek(b)
                    head = f.read(1 << 16).split(b"\n")
                good = good and head[0][:1] == b"@" and (len(head) < 3 or head[2][:1] == b"+")
    verdicts = [None] * world
    dist.all_gather_object(verdicts, good)
    if not all(verdicts):
        return None
    return {"ranges": ranges, "records": n_records * (rank + 1) // world - n_records * rank // world if paths else 0}


def check_range_complete(plan, n_read):
    """Blank lines between records are legal FASTQ for the whole-file reader but shift the line arithmetic of plan_ranges: the ranks
    together then hold fewer records than lines / 4, which shows here on at least one rank. Loud, because R1 / R2 pairing across
    ranks can no longer be trusted."""
    if n_read != plan["records"]:
        raise ValueError(f"this rank's byte range held {n_read} records where the line count promised {plan['records']}: the FASTQ input "
                         "has blank lines or is not four lines per record; rerun with NB_RANGE_SHARDING=0 (batches dealt round-robin)")


def _dist_setup(device_id):
    """Several ranks (one process per GPU: `align --devices ...` or torchrun): (world, rank, device of this rank, device for the
    collectives' buffers, whether this call created the process group). The group is created BEFORE the first GPU call of this
    process. Backend: "nccl" (= RCCL over xGMI) unless NB_DIST_BACKEND says otherwise (gloo: ranks sharing one GPU, CPU tests)."""
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 1, 0, device_id, None, False
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    devices = [int(x) for x in os.environ["NB_DEVICES"].split(",")] if os.environ.get("NB_DEVICES") else None
    local = int(os.environ.get("LOCAL_RANK", rank))
    dev = devices[rank % len(devices)] if devices else local % max(1, torch.cuda.device_count())
    backend = os.environ.get("NB_DIST_BACKEND", "nccl")
    own = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        own = True
    cdev = torch.device("cuda", dev) if dist.get_backend() == "nccl" else torch.device("cpu")
    return world, rank, dev, cdev, own


def _finish_fused(run, world, rank, threshold, disable):
    """Fused per-cell counts at the end of the job. One rank: reduce the accumulated rows. Several ranks: exchange the merged rows
    by hash(cell barcode) mod world, reduce the own cells, rank 0 merges the per-rank outputs (merge.py)."""
    if world == 1:
        return run.rows.finish(None, run.feat_strings, run.report_path, threshold, disable)
    import torch.distributed as dist
    from . import merge
    from .report import report_rows
    if run.rows.umi_index:
        raise NotImplementedError("UMI tags longer than 21 characters or outside ACGTN cannot be merged across ranks (rank-local ids)")
    cols = [np.concatenate([p[j] for p in run.rows.parts]) if run.rows.parts else np.zeros(0, dtype=np.int64) for j in range(4)]
    cbs, cb_ids, umi, fts, ft_ids, scores = merge.exchange_rows_by_cell(run.rows.cb_strings, cols[0], cols[1], run.feat_strings, cols[2], cols[3])
    part = f"{run.report_path}.rank{rank}.part"
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):          # (the reducer's "Dropped n UMIs" line once per rank would be noise)
        report_rows(cb_ids, umi, ft_ids, scores, cbs, fts, part, threshold, disable)
    dist.barrier()
    if rank == 0:
        merge.merge_report_outputs([f"{run.report_path}.rank{r}.part" for r in range(world)], run.report_path)
    dist.barrier()
    os.remove(part)
    return int(cols[0].shape[0])


def _warn_library_options(lib):
    """Config fields of a library (types.py:20-25) that the Rust aligner reads and this engine does not act on."""
    c, defaults = lib.config, default_config()
    for key in ("trim_target_length", "trim_strictness"):
        if c.get(key) != defaults[key]:
            _warn_not_honoured(f"library {lib.stem}: {key} = {c.get(key)!r} (default {defaults[key]!r}): reads are NOT trimmed by this "
                               "engine (docs/SPEC.md §5/§7)")
    if c.get("filters"):
        _warn_not_honoured(f"library {lib.stem}: filters = {c.get('filters')!r}: library filters are not applied (docs/SPEC.md §5/§7)")
    if str(c.get("data_type", "RNA")) != "RNA":
        _warn_not_honoured(f"library {lib.stem}: data_type = {c.get('data_type')!r}: has no effect in this engine (docs/SPEC.md §5/§7)")


def _timed(prof, name, fn, *args):
    t = time.perf_counter()
    try:
        return fn(*args)
    finally:
        prof.add(name, time.perf_counter() - t)


class _StageClock:
    """Wall time per pipeline stage; printed at the end of align() when NB_PROFILE is set."""

    def __init__(self):
        self.t = {}

    def add(self, name, dt):
        self.t[name] = self.t.get(name, 0.0) + dt

    def __call__(self, name):
        clock = self

        class _Span:
            def __enter__(self):
                self.t0 = time.perf_counter()

            def __exit__(self, *exc):
                clock.add(name, time.perf_counter() - self.t0)
        return _Span()

    def report(self):
        return ", ".join(f"{k} {v:.2f} s" for k, v in self.t.items())


def _top_up(reader, batch, want):
    """A reader may end a batch early (longer read ahead, csrc/nb_ingest.cpp); mates must stay aligned with their partners, so
    fetch the rest and concatenate (rare: copies)."""
    parts = [{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()}]
    have = batch["n"]
    while have < want:
        more = reader.next_batch(want - have, MAX_READ_LEN, scratch=True)   # never the buffers of the batch in flight
        if more["n"] == 0:
            raise ValueError("R1 and R2 FASTQ files have different numbers of records")
        parts.append({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in more.items()})
        have += more["n"]
    max_len = max(p["max_len"] for p in parts)
    words = np.concatenate([p["words"] if p["max_len"] == max_len else _repack(p["words"], p["max_len"], max_len) for p in parts])
    return {"n": have, "words": words, "lens": np.concatenate([p["lens"] for p in parts]), "max_len": max_len}


def _tight(batch):
    """(words, max_len) of a reader batch at the layout of its own longest read (the readers deliver that; a batch packed at
    MAX_READ_LEN by the plain entry points is re-strided here)."""
    if "max_len" in batch:
        return batch["words"], batch["max_len"]
    max_len = max(int(batch["lens"].max()), 1)
    if engine.packed_stride_words(max_len) == batch["words"].shape[1]:
        return batch["words"], max_len
    return _repack(batch["words"], MAX_READ_LEN, max_len), max_len


def _repack(words, from_len, to_len):
    """Re-stride packed reads from layout(max_len=from_len) to layout(max_len=to_len); no read may be longer than to_len."""
    nwb_f, nwn_f, nwb_t, nwn_t = (from_len + 31) // 32, (from_len + 63) // 64, (to_len + 31) // 32, (to_len + 63) // 64
    out = np.zeros((words.shape[0], nwb_t + nwn_t), dtype=np.uint64)
    nb, nn = min(nwb_f, nwb_t), min(nwn_f, nwn_t)
    out[:, :nb] = words[:, :nb]
    out[:, nwb_t:nwb_t + nn] = words[:, nwb_f:nwb_f + nn]
    return out

"""ctypes binding of libnimble_amd.so (include/nimble_amd.h). No CPU fallback: a missing library is a hard error."""
import ctypes as ct
import os

from . import _build

_LIB = None

c_u64p = ct.POINTER(ct.c_uint64)
c_i64p = ct.POINTER(ct.c_int64)
c_u16p = ct.POINTER(ct.c_uint16)
c_u32p = ct.POINTER(ct.c_uint32)
c_u8p = ct.POINTER(ct.c_uint8)


class NbConfig(ct.Structure):
    """struct nb_config (include/nimble_amd.h); numeric part of nimble/types.py:10-25 of the reference."""
    _fields_ = [
        ("k", ct.c_int32),
        ("score_threshold", ct.c_int32),
        ("score_filter", ct.c_int32),
        ("num_mismatches", ct.c_int32),
        ("discard_multiple_matches", ct.c_int32),
        ("intersect_level", ct.c_int32),
        ("require_valid_pair", ct.c_int32),
        ("strand_filter", ct.c_int32),
        ("score_percent", ct.c_double),
    ]


class NbIndexStats(ct.Structure):
    _fields_ = [
        ("n_seqs", ct.c_int64), ("n_kmers", ct.c_int64), ("n_canonical", ct.c_int64), ("table_slots", ct.c_int64), ("table_bytes", ct.c_int64),
        ("n_colours", ct.c_int64), ("words_per_row", ct.c_int64), ("device_bytes", ct.c_int64), ("ph_table_bytes", ct.c_int64), ("ph_pilot_bytes", ct.c_int64), ("anchor_bytes", ct.c_int64),
        ("k", ct.c_int32), ("uploaded", ct.c_int32), ("compact", ct.c_int32), ("words_internal", ct.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/nimble_amd.h declares
SIGNATURES = {
    "nb_version": (ct.c_int32, []),
    "nb_last_error": (ct.c_char_p, []),
    "nb_ctx_create": (ct.c_int32, [ct.c_int32, ct.c_int64, ct.POINTER(ct.c_void_p)]),
    "nb_ctx_destroy": (ct.c_int32, [ct.c_void_p]),
    "nb_ctx_sync": (ct.c_int32, [ct.c_void_p, ct.c_void_p]),
    "nb_device_alloc": (ct.c_int32, [ct.c_void_p, ct.c_int64, ct.POINTER(ct.c_void_p)]),
    "nb_device_free": (ct.c_int32, [ct.c_void_p, ct.c_void_p]),
    "nb_host_alloc_pinned": (ct.c_int32, [ct.c_int64, ct.POINTER(ct.c_void_p)]),
    "nb_host_free_pinned": (ct.c_int32, [ct.c_void_p]),
    "nb_stream_create": (ct.c_int32, [ct.c_void_p, ct.POINTER(ct.c_void_p)]),
    "nb_stream_destroy": (ct.c_int32, [ct.c_void_p, ct.c_void_p]),
    "nb_copy_h2d_async": (ct.c_int32, [ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int64, ct.c_void_p]),
    "nb_copy_d2h_async": (ct.c_int32, [ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int64, ct.c_void_p]),
    "nb_index_build": (ct.c_int32, [ct.c_int32, ct.POINTER(ct.c_char_p), c_i64p, ct.c_int32, ct.POINTER(ct.c_void_p)]),
    "nb_index_destroy": (ct.c_int32, [ct.c_void_p]),
    "nb_index_stats": (ct.c_int32, [ct.c_void_p, ct.POINTER(NbIndexStats)]),
    "nb_index_query": (ct.c_int32, [ct.c_void_p, ct.c_char_p, c_u8p, c_u8p, c_u8p, c_u8p, c_u64p]),
    "nb_index_filter_maybe": (ct.c_int32, [ct.c_void_p, ct.c_char_p, ct.c_int32, ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32)]),
    "nb_classes_copy_device": (ct.c_int32, [ct.c_void_p, ct.c_int64, ct.c_void_p, ct.c_void_p, ct.c_void_p]),
    "nb_rows_to_external": (ct.c_int32, [ct.c_void_p, c_u64p, ct.c_int64, c_u64p]),
    "nb_index_anchor_query": (ct.c_int32, [ct.c_void_p, ct.c_char_p, ct.c_int32, ct.POINTER(ct.c_int64), ct.POINTER(ct.c_int32), c_u64p]),
    "nb_index_upload": (ct.c_int32, [ct.c_void_p, ct.c_void_p]),
    "nb_packed_stride_words": (ct.c_int64, [ct.c_int32]),
    "nb_pack_reads": (ct.c_int32, [ct.c_char_p, c_i64p, ct.c_int64, ct.c_int32, c_u64p, c_u16p]),
    "nb_align_batch": (ct.c_int32, [ct.c_void_p, ct.POINTER(NbConfig), ct.c_void_p, ct.c_void_p, ct.c_int32, ct.c_void_p,
                                    ct.c_void_p, ct.c_int32, ct.c_int64, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p]),
    "nb_classes_count": (ct.c_int32, [ct.c_void_p, c_i64p]),
    "nb_classes_read": (ct.c_int32, [ct.c_void_p, ct.c_int64, c_u64p, c_i64p]),
    "nb_reasons_read": (ct.c_int32, [ct.c_void_p, c_i64p]),
    "nb_results_reset": (ct.c_int32, [ct.c_void_p, ct.c_void_p]),
    "nb_counts_device_ptr": (ct.c_int32, [ct.c_void_p, ct.POINTER(ct.c_void_p), c_i64p]),
    "nb_last_kernel_names": (ct.c_char_p, [ct.c_void_p]),
    "nb_kernel_name": (ct.c_char_p, [ct.c_int32, ct.c_int32, ct.c_int32, ct.c_int32]),
    "nb_report_tsv": (ct.c_int32, [ct.c_char_p, ct.c_char_p, ct.c_double, ct.c_int32, c_i64p]),
    "nb_report_rows": (ct.c_int32, [ct.c_int64, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int64, ct.POINTER(ct.c_char_p),
                                    ct.c_int64, ct.POINTER(ct.c_char_p), ct.c_double, ct.c_int32, ct.c_char_p, c_i64p]),
    "nb_fastq_open": (ct.c_int32, [ct.c_char_p, ct.POINTER(ct.c_void_p)]),
    "nb_fastq_close": (ct.c_int32, [ct.c_void_p]),
    "nb_fastq_open_range": (ct.c_int32, [ct.c_char_p, ct.c_int64, ct.c_int64, ct.POINTER(ct.c_void_p)]),
    "nb_file_count_newlines": (ct.c_int32, [ct.c_char_p, ct.c_int64, ct.c_int64, c_i64p, ct.POINTER(ct.c_int32)]),
    "nb_file_nth_newline": (ct.c_int32, [ct.c_char_p, ct.c_int64, ct.c_int64, c_i64p]),
    "nb_fastq_set_threads": (ct.c_int32, [ct.c_void_p, ct.c_int32]),
    "nb_fastq_index_batch": (ct.c_int32, [ct.c_void_p, ct.c_int64, c_i64p, ct.POINTER(ct.c_int32)]),
    "nb_fastq_next_batch_tight": (ct.c_int32, [ct.c_void_p, ct.c_int64, ct.c_int32, ct.c_void_p, ct.c_void_p, c_i64p, ct.POINTER(ct.c_int32),
                                               ct.c_void_p, ct.c_int32, ct.c_void_p, ct.c_int32, ct.c_void_p]),
    "nb_fastq_pack_batch": (ct.c_int32, [ct.c_void_p, ct.c_int32, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int32, ct.c_void_p, ct.c_int32,
                                         ct.c_void_p]),
    "nb_bam_open": (ct.c_int32, [ct.c_char_p, ct.POINTER(ct.c_void_p)]),
    "nb_bam_close": (ct.c_int32, [ct.c_void_p]),
    "nb_bam_set_threads": (ct.c_int32, [ct.c_void_p, ct.c_int32]),
    "nb_bam_next_batch_tight": (ct.c_int32, [ct.c_void_p, ct.c_int64, ct.c_int32, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p,
                                             ct.c_void_p, ct.c_int32, c_i64p, ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32)]),
    "nb_bam_next_batch": (ct.c_int32, [ct.c_void_p, ct.c_int64, ct.c_int32, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_void_p,
                                       ct.c_void_p, ct.c_int32, c_i64p, ct.POINTER(ct.c_int32)]),
    "nb_whitelist_load": (ct.c_int32, [ct.c_char_p, ct.POINTER(ct.c_void_p)]),
    "nb_whitelist_destroy": (ct.c_int32, [ct.c_void_p]),
    "nb_whitelist_set_threads": (ct.c_int32, [ct.c_void_p, ct.c_int32]),
    "nb_classes_read_range": (ct.c_int32, [ct.c_void_p, ct.c_int64, ct.c_int64, c_u64p, c_i64p]),
    "nb_class_features": (ct.c_int32, [ct.c_int64, ct.c_int32, ct.c_void_p, ct.c_int32, ct.c_void_p, ct.POINTER(ct.c_char_p), ct.c_int32, ct.c_int32,
                                       ct.c_int32, ct.c_void_p, ct.c_int64, ct.c_void_p, ct.c_void_p, ct.c_int64, ct.c_void_p, ct.c_void_p]),
    "nb_gzip_members": (ct.c_int32, [ct.c_void_p, ct.c_int64, ct.c_int32, ct.c_int32, ct.c_void_p, ct.c_int64, c_i64p]),
    "nb_gunzip_file": (ct.c_int32, [ct.c_char_p, ct.c_void_p, ct.c_int64, c_i64p]),
    "nb_fastq_to_bam": (ct.c_int32, [ct.c_char_p, ct.c_char_p, ct.c_void_p, ct.c_char_p, ct.c_int32, ct.c_int32, ct.c_int32, c_i64p]),
    "nb_whitelist_size": (ct.c_int64, [ct.c_void_p]),
    "nb_whitelist_correct": (ct.c_int32, [ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_int64, ct.c_int32, ct.c_void_p, ct.c_void_p, c_i64p]),
    "nb_format_tenx_rows": (ct.c_int32, [ct.c_int64, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.POINTER(ct.c_char_p), ct.c_int64, ct.c_void_p,
                                         ct.c_int32, ct.c_void_p, ct.c_void_p, ct.c_int32, ct.c_void_p, ct.c_int64, c_i64p, c_i64p]),
    "nb_fastq_next_batch": (ct.c_int32, [ct.c_void_p, ct.c_int64, ct.c_int32, ct.c_void_p, ct.c_void_p, c_i64p, ct.c_void_p,
                                         ct.c_int32, ct.c_void_p, ct.c_int32, ct.c_void_p]),
}


class NimbleError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libnimble_amd error {code}: {message}")
        self.code = code


def lib():
    """Load (once) the in-tree shared library. Raises if it has not been built: there is no Python/CPU fallback."""
    global _LIB
    if _LIB is None:
        path = _build.LIB_PATH
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). nimble_amd has no CPU fallback for the alignment path.")
        try:
            # torch bundles its own libamdhip64.so.7; load it first so the dynamic loader binds this library to the
            # SAME HIP runtime (two runtimes in one process cannot both own the GPU). Without torch, /opt/rocm's is used.
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = ct.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the header and the library disagree
            fn.restype, fn.argtypes = res, args
        _LIB = handle
    return _LIB


def check(rc):
    if rc < 0:
        raise NimbleError(rc, (lib().nb_last_error() or b"").decode("utf-8", "replace"))
    return rc

"""INTEGRATION.md §B as a runnable script: drive libnimble_amd.so through plain ctypes — no PyTorch, no GPU array library.

usage: integration_b.py LIB.json READS.txt OUT.npz      (READS.txt: one read per line)
Writes per-read class / info words and the class table to OUT.npz and asserts that torch was never imported."""
import ctypes as ct
import json
import os
import sys

import numpy as np

lib_path, reads_path, out_path = sys.argv[1:4]
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = ct.CDLL(os.path.join(ROOT, "nimble_amd", "libnimble_amd.so"))


class nb_config(ct.Structure):
    _fields_ = [("k", ct.c_int32), ("score_threshold", ct.c_int32), ("score_filter", ct.c_int32), ("num_mismatches", ct.c_int32),
                ("discard_multiple_matches", ct.c_int32), ("intersect_level", ct.c_int32), ("require_valid_pair", ct.c_int32),
                ("strand_filter", ct.c_int32), ("score_percent", ct.c_double)]


L.nb_last_error.restype = ct.c_char_p
L.nb_packed_stride_words.restype = ct.c_int64


def check(rc):
    if rc < 0:
        raise RuntimeError(L.nb_last_error().decode())


cfg_json, data = json.load(open(lib_path))
seqs = [s.encode() for s in data["columns"][3]]
lens = (ct.c_int64 * len(seqs))(*map(len, seqs))
index, ctx, stream = ct.c_void_p(), ct.c_void_p(), ct.c_void_p()
check(L.nb_index_build(len(seqs), (ct.c_char_p * len(seqs))(*seqs), lens, 20, ct.byref(index)))
check(L.nb_ctx_create(0, ct.c_int64(1 << 16), ct.byref(ctx)))
check(L.nb_index_upload(ctx, index))
check(L.nb_stream_create(ctx, ct.byref(stream)))
cfg = nb_config(20, cfg_json["score_threshold"], cfg_json["score_filter"], cfg_json["num_mismatches"], int(cfg_json["discard_multiple_matches"]),
                cfg_json["intersect_level"], int(cfg_json["require_valid_pair"]), 0, cfg_json["score_percent"])

reads = [l.strip().encode() for l in open(reads_path)]
n = len(reads)
max_len = max(len(r) for r in reads)
offsets = np.zeros(n + 1, np.int64)
np.cumsum([len(r) for r in reads], out=offsets[1:])
blob = b"".join(reads)
stride = L.nb_packed_stride_words(max_len)
words, lens16 = np.zeros((n, stride), np.uint64), np.zeros(n, np.uint16)
check(L.nb_pack_reads(blob, offsets.ctypes.data_as(ct.POINTER(ct.c_int64)), ct.c_int64(n), max_len, words.ctypes.data_as(ct.POINTER(ct.c_uint64)),
                      lens16.ctypes.data_as(ct.POINTER(ct.c_uint16))))
d_reads, d_lens, d_class, d_info = ct.c_void_p(), ct.c_void_p(), ct.c_void_p(), ct.c_void_p()
for d, nbytes in ((d_reads, words.nbytes), (d_lens, lens16.nbytes), (d_class, 4 * n), (d_info, 4 * n)):
    check(L.nb_device_alloc(ctx, ct.c_int64(nbytes), ct.byref(d)))
check(L.nb_copy_h2d_async(ctx, d_reads, ct.c_void_p(words.ctypes.data), ct.c_int64(words.nbytes), stream))
check(L.nb_copy_h2d_async(ctx, d_lens, ct.c_void_p(lens16.ctypes.data), ct.c_int64(lens16.nbytes), stream))
check(L.nb_align_batch(ctx, ct.byref(cfg), d_reads, d_lens, max_len, None, None, 0, ct.c_int64(n), d_class, d_info, None, stream))
cls, info = np.empty(n, np.uint32), np.empty(n, np.uint32)
check(L.nb_copy_d2h_async(ctx, ct.c_void_p(cls.ctypes.data), d_class, ct.c_int64(4 * n), stream))
check(L.nb_copy_d2h_async(ctx, ct.c_void_p(info.ctypes.data), d_info, ct.c_int64(4 * n), stream))
check(L.nb_ctx_sync(ctx, stream))
n_classes = ct.c_int64()
check(L.nb_classes_count(ctx, ct.byref(n_classes)))
rows = np.zeros((n_classes.value, (len(seqs) + 63) // 64), np.uint64)
counts = np.zeros(n_classes.value, np.int64)
check(L.nb_classes_read(ctx, n_classes, rows.ctypes.data_as(ct.POINTER(ct.c_uint64)), counts.ctypes.data_as(ct.POINTER(ct.c_int64))))
for d in (d_reads, d_lens, d_class, d_info):
    check(L.nb_device_free(ctx, d))
check(L.nb_stream_destroy(ctx, stream))
check(L.nb_ctx_destroy(ctx))
check(L.nb_index_destroy(index))
assert "torch" not in sys.modules
np.savez(out_path, cls=cls, info=info, rows=rows, counts=counts)
print(f"integration_b ok: {n} reads, {n_classes.value} classes, torch imported: {'torch' in sys.modules}")

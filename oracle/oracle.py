"""ctypes wrapper of oracle/libnb_oracle.so (nb_oracle.cpp). TEST INFRASTRUCTURE — never imported by nimble_amd/.

Parity unpinned against the Rust nimble-aligner (see nb_oracle.cpp header and docs/SPEC.md)."""
import ctypes as ct
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libnb_oracle.so")
_LIB = None


class OracleConfig(ct.Structure):
    _fields_ = [("k", ct.c_int32), ("score_threshold", ct.c_int32), ("score_filter", ct.c_int32), ("num_mismatches", ct.c_int32),
                ("discard_multiple_matches", ct.c_int32), ("intersect_level", ct.c_int32), ("require_valid_pair", ct.c_int32),
                ("strand_filter", ct.c_int32), ("score_percent", ct.c_double)]


def build(force=False):
    src = os.path.join(HERE, "nb_oracle.cpp")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(LIB_PATH):
        subprocess.run(["make", "-C", HERE, "-B" if force else "-s", "libnb_oracle.so"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = ct.CDLL(LIB_PATH)
        L.nbo_index_build.restype = ct.c_void_p
        L.nbo_index_build.argtypes = [ct.c_int32, ct.POINTER(ct.c_char_p), ct.POINTER(ct.c_int64), ct.c_int32]
        L.nbo_index_destroy.argtypes = [ct.c_void_p]
        for f in ("nbo_index_num_kmers", "nbo_index_num_colours"):
            getattr(L, f).restype = ct.c_int64
            getattr(L, f).argtypes = [ct.c_void_p]
        L.nbo_index_words.restype = ct.c_int32
        L.nbo_index_words.argtypes = [ct.c_void_p]
        L.nbo_index_query.restype = ct.c_int32
        L.nbo_index_query.argtypes = [ct.c_void_p, ct.c_char_p] + [ct.POINTER(ct.c_uint8)] * 4 + [ct.POINTER(ct.c_uint64)]
        L.nbo_align.restype = ct.c_int32
        L.nbo_align.argtypes = [ct.c_void_p, ct.POINTER(OracleConfig), ct.c_int64, ct.c_char_p, ct.POINTER(ct.c_int64), ct.c_char_p,
                                ct.POINTER(ct.c_int64), ct.POINTER(ct.c_uint64), ct.POINTER(ct.c_int32), ct.POINTER(ct.c_uint8),
                                ct.POINTER(ct.c_uint8), ct.POINTER(ct.c_uint8), ct.POINTER(ct.c_int32), ct.c_int32]
        _LIB = L
    return _LIB


def _cfg(cfg):
    """Accepts an OracleConfig, a dict, or any ctypes struct with the same field names (e.g. nimble_amd NbConfig)."""
    if isinstance(cfg, OracleConfig):
        return cfg
    get = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
    return OracleConfig(*[get(name) for name, _ in OracleConfig._fields_])


class OracleIndex:
    def __init__(self, sequences, k=20):
        seqs = [s.encode() if isinstance(s, str) else bytes(s) for s in sequences]
        arr = (ct.c_char_p * len(seqs))(*seqs)
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        self._h = lib().nbo_index_build(len(seqs), arr, lens.ctypes.data_as(ct.POINTER(ct.c_int64)), int(k))
        if not self._h:
            raise ValueError("oracle index build failed")
        self.k, self.n_seqs = int(k), len(seqs)
        self.words = lib().nbo_index_words(self._h)

    @property
    def num_kmers(self):
        return lib().nbo_index_num_kmers(self._h)

    @property
    def num_colours(self):
        return lib().nbo_index_num_colours(self._h)

    def query(self, kmer):
        le, re, ri, li = ct.c_uint8(), ct.c_uint8(), ct.c_uint8(), ct.c_uint8()
        bits = np.zeros(self.words, dtype=np.uint64)
        rc = lib().nbo_index_query(self._h, kmer.encode() if isinstance(kmer, str) else kmer, ct.byref(le), ct.byref(re), ct.byref(ri),
                                   ct.byref(li), bits.ctypes.data_as(ct.POINTER(ct.c_uint64)))
        if rc == 0:
            return None
        members = np.nonzero(np.unpackbits(bits.view(np.uint8), bitorder="little"))[0].tolist()
        return {"lext": le.value, "rext": re.value, "r_internal": ri.value, "l_internal": li.value, "members": members}

    def align(self, cfg, blob1, off1, blob2=None, off2=None, n_threads=1):
        """Returns dict: set_bits[n, words] u64, score i32, reason u8, orientation u8, mates u8, cov[n, 4] i32."""
        off1 = np.ascontiguousarray(off1, dtype=np.int64)
        n = len(off1) - 1
        out = {"set_bits": np.zeros((n, self.words), dtype=np.uint64), "score": np.zeros(n, dtype=np.int32),
               "reason": np.zeros(n, dtype=np.uint8), "orientation": np.zeros(n, dtype=np.uint8),
               "mates": np.zeros(n, dtype=np.uint8), "cov": np.zeros((n, 4), dtype=np.int32)}
        p64 = ct.POINTER(ct.c_int64)
        o2 = None
        if blob2 is not None:
            off2 = np.ascontiguousarray(off2, dtype=np.int64)
            o2 = off2.ctypes.data_as(p64)
        c = _cfg(cfg)
        rc = lib().nbo_align(self._h, ct.byref(c), n, blob1, off1.ctypes.data_as(p64), blob2, o2,
                             out["set_bits"].ctypes.data_as(ct.POINTER(ct.c_uint64)), out["score"].ctypes.data_as(ct.POINTER(ct.c_int32)),
                             out["reason"].ctypes.data_as(ct.POINTER(ct.c_uint8)), out["orientation"].ctypes.data_as(ct.POINTER(ct.c_uint8)),
                             out["mates"].ctypes.data_as(ct.POINTER(ct.c_uint8)), out["cov"].ctypes.data_as(ct.POINTER(ct.c_int32)), int(n_threads))
        if rc != 0:
            raise RuntimeError(f"nbo_align failed ({rc})")
        return out

    def align_strs(self, cfg, reads, mates=None, n_threads=1):
        def cat(rs):
            bs = [r.encode() if isinstance(r, str) else bytes(r) for r in rs]
            off = np.zeros(len(bs) + 1, dtype=np.int64)
            np.cumsum([len(b) for b in bs], out=off[1:])
            return b"".join(bs), off
        b1, o1 = cat(reads)
        b2, o2 = cat(mates) if mates is not None else (None, None)
        return self.align(cfg, b1, o1, b2, o2, n_threads)

    def close(self):
        if getattr(self, "_h", None):
            lib().nbo_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

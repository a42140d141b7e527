"""The engine's own DEFLATE/gzip decoder (csrc/nb_inflate.cpp, what reads .gz FASTQ input) against zlib: every block type,
header field, member layout and failure mode, through the C-ABI entry point nb_gunzip_file."""
import ctypes as ct
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from nimble_amd import _lib


def gunzip(path, cap):
    out = np.empty(max(cap, 1), dtype=np.uint8)
    n = ct.c_int64()
    _lib.check(_lib.lib().nb_gunzip_file(str(path).encode(), out.ctypes.data, cap, ct.byref(n)))
    return out[:n.value].tobytes()


def fastq_like(rng, n):
    recs = []
    for i in range(n):
        L = int(rng.integers(30, 151))
        seq = "".join("ACGTN"[c] for c in rng.choice(5, L, p=[0.25, 0.25, 0.25, 0.24, 0.01]))
        qual = "".join(chr(33 + int(q)) for q in rng.integers(2, 41, L))
        recs.append(f"@read{i}/1 lane:{i % 7}\n{seq}\n+\n{qual}\n")
    return "".join(recs).encode()


def raw_member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, header=None):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    body = c.compress(data) + c.flush()
    head = header if header is not None else b"\x1f\x8b\x08\x00" + b"\0" * 4 + b"\x00\x03"
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


@pytest.fixture(scope="module")
def corpus():
    rng = np.random.default_rng(99)
    return {
        "empty": b"",
        "one_byte": b"A",
        "short_text": b"hello hello hello hello, deflate\n" * 3,
        "fastq": fastq_like(rng, 4000),
        "runs": b"I" * 100_000 + b"ACGT" * 50_000 + bytes(range(256)) * 300,
        "random": rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes(),                       # incompressible: stored blocks at level > 0 too
        "skewed": rng.choice(np.arange(256, dtype=np.uint8), 400_000, p=np.r_[[0.5], np.full(255, 0.5 / 255)]).tobytes(),   # long Huffman codes
        "far_matches": (lambda b: b + rng.integers(0, 256, 32_000, dtype=np.uint8).tobytes() + b)(rng.integers(0, 4, 700, dtype=np.uint8).tobytes()),
    }


@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_every_corpus_entry_round_trips(built, tmp_path, corpus, level):
    for name, data in corpus.items():
        p = tmp_path / f"{name}.gz"
        p.write_bytes(raw_member(data, level))
        assert gunzip(p, len(data) + 16) == data, (name, level)


def test_fixed_and_huffman_only_and_rle_strategies(built, tmp_path, corpus):
    for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
        for name in ("short_text", "fastq", "runs", "skewed"):
            p = tmp_path / "s.gz"
            p.write_bytes(raw_member(corpus[name], 6, strategy))
            assert gunzip(p, len(corpus[name]) + 16) == corpus[name], (name, strategy)


def test_header_fields_members_and_padding(built, tmp_path, corpus):
    data = corpus["fastq"]
    a, b, c = data[:100_000], data[100_000:100_001], data[100_001:]
    extra = b"XY" + struct.pack("<H", 5) + b"12345"
    h_extra = b"\x1f\x8b\x08" + bytes([4 | 8 | 16 | 2]) + b"\0" * 4 + b"\x00\x03" + struct.pack("<H", len(extra)) + extra + b"name.fq\0" + b"a comment\0" + b"\x12\x34"
    blob = raw_member(a, 6, header=h_extra) + raw_member(b"", 6) + raw_member(b, 1) + raw_member(c, 9) + b"\0" * 37
    p = tmp_path / "multi.gz"
    p.write_bytes(blob)
    assert gunzip(p, len(data) + 16) == data
    with gzip.open(tmp_path / "py.gz", "wb", compresslevel=5) as f:          # Python's writer: FNAME + mtime
        f.write(data)
    assert gunzip(tmp_path / "py.gz", len(data) + 16) == data
    # BGZF-style: thousands of small members
    (tmp_path / "bgzf.gz").write_bytes(b"".join(raw_member(data[i:i + 3000], 6) for i in range(0, len(data), 3000)))
    assert gunzip(tmp_path / "bgzf.gz", len(data) + 16) == data


def test_large_stream_crosses_pieces_with_full_window(built, tmp_path):
    """30 MB of text with matches at all distances up to 32 KiB: the decoder is resumed hundreds of times (1 MiB output pieces,
    4 MiB input windows) and must carry its 32 KiB history across every boundary."""
    rng = np.random.default_rng(5)
    words = [rng.integers(65, 91, int(rng.integers(3, 40)), dtype=np.uint8).tobytes() for _ in range(3000)]
    parts, size = [], 0
    while size < 30_000_000:
        w = words[int(rng.integers(0, len(words)))]
        parts.append(w); size += len(w)
        if rng.random() < 0.001:
            r = rng.integers(0, 256, int(rng.integers(1000, 70_000)), dtype=np.uint8).tobytes()   # incompressible stretches: stored blocks mid-stream
            parts.append(r); size += len(r)
    data = b"".join(parts)
    for level in (1, 9):
        p = tmp_path / f"big{level}.gz"
        p.write_bytes(raw_member(data, level))
        got = gunzip(p, len(data) + 16)
        assert len(got) == len(data) and got == data


def test_failures_are_reported_not_decoded(built, tmp_path, corpus):
    data = corpus["fastq"]
    good = raw_member(data, 6)
    cases = {
        "truncated_body": good[: len(good) // 2],
        "truncated_trailer": good[:-5],
        "bad_crc": good[:-8] + struct.pack("<I", (zlib.crc32(data) ^ 1) & 0xFFFFFFFF) + good[-4:],
        "bad_isize": good[:-4] + struct.pack("<I", len(data) + 1),
        "bad_magic": b"\x1f\x8c" + good[2:],
        "garbage_after": good + b"not a member",
        "flipped_bit": good[:5000] + bytes([good[5000] ^ 0x10]) + good[5001:],
        "reserved_block_type": b"\x1f\x8b\x08\x00" + b"\0" * 6 + b"\x07\x00" + b"\0" * 8,
    }
    for name, blob in cases.items():
        p = tmp_path / f"{name}.gz"
        p.write_bytes(blob)
        with pytest.raises(_lib.NimbleError):
            gunzip(p, len(data) + 1000)
    small = tmp_path / "small_cap.gz"
    small.write_bytes(good)
    with pytest.raises(_lib.NimbleError):
        gunzip(small, 1000)


def test_fastq_reader_on_gzip_equals_plain(built, tmp_path):
    from nimble_amd.align import FastqReader
    rng = np.random.default_rng(8)
    text = fastq_like(rng, 60_000)
    (tmp_path / "r.fastq").write_bytes(text)
    (tmp_path / "r.fastq.gz").write_bytes(raw_member(text[:1_000_000], 6) + raw_member(text[1_000_000:], 1))
    from tests.helpers import _bgzf_block                      # bgzip-style FASTQ: inflated block-parallel, like a BAM
    (tmp_path / "r.fastq.bgz").write_bytes(b"".join(_bgzf_block(text[i:i + 60000]) for i in range(0, len(text), 60000)) + _bgzf_block(b""))
    got = []
    for name in ("r.fastq", "r.fastq.gz", "r.fastq.bgz"):
        r = FastqReader(str(tmp_path / name), 3)
        rows = []
        while True:
            b = r.next_batch(25_000, 200)
            if b["n"] == 0:
                break
            rows.append((b["max_len"], b["words"].copy(), b["lens"].copy()))
        r.close()
        got.append(rows)
    for other in got[1:]:
        assert len(got[0]) == len(other)
        for (m0, w0, l0), (m1, w1, l1) in zip(got[0], other):
            assert m0 == m1 and (w0 == w1).all() and (l0 == l1).all()


def test_sync_and_full_flush_markers(built, tmp_path, corpus):
    """pigz-style streams: empty stored blocks (Z_SYNC_FLUSH / Z_FULL_FLUSH markers) between the compressed blocks, at odd bit
    offsets; also a stream that ends in such a marker followed by an empty final block."""
    data = corpus["fastq"] + corpus["runs"]
    for flush in (zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = b""
        for i in range(0, len(data), 7919):
            body += c.compress(data[i:i + 7919]) + c.flush(flush)
        body += c.flush()
        p = tmp_path / "flush.gz"
        p.write_bytes(b"\x1f\x8b\x08\x00" + b"\0" * 4 + b"\x00\x03" + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF))
        assert gunzip(p, len(data) + 16) == data

"""Bounds-checked run of the anchor-extend kernels (debug aid; run on the GPU box):
    NB_EXTRA_HIPCC_FLAGS="-DNB_ANCHOR_PROF -DNB_ANCHOR_CHECK" python scripts/anchor_check.py [loci alleles locus_len seed]
Every computed index of the kernels is compared with the size of its array; violations are counted per kind and the access is
redirected, so that an indexing bug shows up as a count instead of a GPU fault."""
import ctypes as ct
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nimble_amd import _build, _lib, engine, synth  # noqa: E402

_build.build(force=True)
loci, alleles, locus_len, seed = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (3, 90, 260, 61)
names, seqs, _ = synth.synth_library(loci, alleles, locus_len=locus_len, seed=seed)
strs = [synth.codes_to_str(s) for s in seqs]
codes = synth.synth_reads(seqs, 6000, 150, seed=12, err=0.012, n_rate=0.002)
reads = ["".join("ACGTN"[c] for c in row) for row in codes]
rng = np.random.default_rng(2)
for j in range(0, 600, 7):
    reads[j] = reads[j][:int(rng.integers(15, 150))]
words, lens = engine.pack_reads(reads, 150)
ix = engine.Index(strs, 20)
print(ix.stats())
dev = torch.device("cuda:0")
ctx = engine.Context(0, max_classes=1 << 16)
ctx.upload(ix)
d_reads = torch.from_numpy(words.view(np.int64)).to(dev)
d_lens = torch.from_numpy(lens.view(np.int16)).to(dev)
n = len(reads)
oc = torch.empty(n, dtype=torch.int32, device=dev); oi = torch.empty(n, dtype=torch.int32, device=dev); ov = torch.empty(n, dtype=torch.int64, device=dev)
ctx.align_batch(engine.make_config(), d_reads, d_lens, 150, oc, oi, ov, n=n)
ctx.sync()
fn = _lib.lib().nb_debug_anchor_prof
fn.restype = ct.c_int; fn.argtypes = [ct.POINTER(ct.c_ulonglong)]
buf = (ct.c_ulonglong * 16)()
assert fn(buf) == 0
kinds = {10: "pilot group", 11: "table slot", 12: "text word", 13: "sparse entry", 14: "class-row word", 15: "LDS record dword"}
print("kernels:", ctx.last_kernels())
print("violations:", {kinds[i]: int(buf[i]) for i in range(10, 16)})

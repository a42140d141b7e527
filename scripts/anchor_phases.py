"""Phase clock of the anchor-extend kernel (tuning aid; run on the GPU box):
    NB_EXTRA_HIPCC_FLAGS=-DNB_ANCHOR_PROF python scripts/anchor_phases.py [reads]
builds the library with the phase counters compiled in (the default build has none), maps one c2-shaped batch and prints the share of
wave-cycles per phase: stage, probes (+ stream_rc), walk rounds, other-strand scan, decide, intern, outputs/lists."""
import ctypes as ct
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nimble_amd import _build, _lib, engine, synth  # noqa: E402

_build.build(force=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
loci, alleles = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (20, 50)
read_len = int(sys.argv[4]) if len(sys.argv) > 4 else 150
names, seqs, _ = synth.synth_library(loci, alleles)
strs = [synth.codes_to_str(s) for s in seqs]
dev = torch.device("cuda:0")
reads = synth.synth_reads_packed_torch(seqs, n, read_len, 2, dev)
ix = engine.Index(strs, 20)
ctx = engine.Context(0, max_classes=1 << 20)
ctx.upload(ix)
cfg = engine.make_config()
oc = torch.empty(n, dtype=torch.int32, device=dev); oi = torch.empty(n, dtype=torch.int32, device=dev)
L = _lib.lib()
fn = L.nb_debug_anchor_prof
fn.restype = ct.c_int; fn.argtypes = [ct.POINTER(ct.c_ulonglong)]
buf = (ct.c_ulonglong * 16)()
hfn = L.nb_debug_anchor_hist
hfn.restype = ct.c_int; hfn.argtypes = [ct.POINTER(ct.c_ulonglong)]
hbuf = (ct.c_ulonglong * 128)()
for it in range(3):
    ctx.reset()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    ctx.align_batch(cfg, reads, None, read_len, oc, oi, None, n=n)
    t1.record()
    ctx.sync()
    assert fn(buf) == 0 and hfn(hbuf) == 0
print(f"batch: {t0.elapsed_time(t1):.3f} ms (all kernels, instrumented build)")
v = np.array(list(buf), dtype=np.float64)
tot = v[:7].sum()
for name, x in zip(("stage", "probes+rc", "rounds", "other-strand scan", "decide", "intern", "outputs+lists"), v[:7]):
    print(f"{name:20s} {100 * x / tot:6.2f} %   {x / max(v[9], 1):10.0f} cycles per wave-batch")
print("redo flags (reads): both-strand k-mer %d, opposite-strand hit %d, left extension %d, wide component %d, any %d" % tuple(int(x) for x in v[10:15]))
print(f"rounds per wave-batch: {v[8] / max(v[9], 1):.1f}; wave-batches: {int(v[9])}; kernels: {ctx.last_kernels()}")
hist = np.array(list(hbuf), dtype=np.float64)
lanes, waves = hist[:64], hist[64:]
print("rounds   reads(%)  cum(%)   wave-batches(%)  cum(%)")
cl = cw = 0.0
for r in range(64):
    if lanes[r] or waves[r]:
        cl += 100 * lanes[r] / max(lanes.sum(), 1); cw += 100 * waves[r] / max(waves.sum(), 1)
        print(f"{r:5d}  {100 * lanes[r] / max(lanes.sum(), 1):8.2f} {cl:7.2f}  {100 * waves[r] / max(waves.sum(), 1):12.2f} {cw:9.2f}")
print(f"mean rounds per read {np.dot(np.arange(64), lanes) / max(lanes.sum(), 1):.2f}, per wave-batch {np.dot(np.arange(64), waves) / max(waves.sum(), 1):.2f}")

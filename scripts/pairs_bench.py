#!/usr/bin/env python
"""Throughput of the general kernel (paired templates / tolerated mismatches) on the GPU box. Not the headline metric."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from nimble_amd import engine, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dev = torch.device("cuda", 0)
_, seqs, _ = synth.synth_library(20, 50)
index = engine.Index([synth.codes_to_str(s) for s in seqs], 20)
ctx = engine.Context(0, 1 << 21)
ctx.upload(index)
r1 = synth.synth_reads_packed_torch(seqs, n, 150, 11, dev)
r2 = synth.synth_reads_packed_torch(seqs, n, 150, 12, dev)
oc = torch.empty(n, dtype=torch.int32, device=dev)
oi = torch.empty(n, dtype=torch.int32, device=dev)
for name, cfg, mates in (("pairs 2x150, M=0", engine.make_config(intersect_level=1), r2), ("single-end 150, M=2", engine.make_config(num_mismatches=2), None)):
    for _ in range(2):
        ctx.align_batch(cfg, r1, None, 150, oc, oi, None, mates=mates, mate_max_len=150 if mates is not None else 0, n=n)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.align_batch(cfg, r1, None, 150, oc, oi, None, mates=mates, mate_max_len=150 if mates is not None else 0, n=n)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 3
    print(f"[general] {name}: {n} templates in {dt * 1e3:.1f} ms = {n / dt / 1e6:.1f} M templates/s")

import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from nimble_amd import engine, synth
names, seqs, _ = synth.synth_library(20, 50)
strs = [synth.codes_to_str(s) for s in seqs]
dev = torch.device("cuda:0")
n = 10_000_000
reads = synth.synth_reads_packed_torch(seqs, n, 150, 2, dev)
ix = engine.Index(strs, 20)
ctx = engine.Context(0, max_classes=1 << 20)
ctx.upload(ix)
cfg = engine.make_config()
oc = torch.empty(n, dtype=torch.int32, device=dev); oi = torch.empty(n, dtype=torch.int32, device=dev)
for it in range(3):
    ctx.reset()
    for _ in range(2):
        ctx.align_batch(cfg, reads, None, 150, oc, oi, None, n=n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows, counts = ctx.classes()
    t1 = time.perf_counter()
    print(f"classes(): {1e3 * (t1 - t0):.3f} ms for {rows.shape} rows", flush=True)

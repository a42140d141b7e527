"""Paired templates on the GPU box: 2 x 150 bp pairs against the c2 library, kernels of one nb_align_batch (HIP events).
    python scripts/pair_bench.py [pairs]          (NB_ANCHOR=0 for the general kernel alone)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nimble_amd import engine, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
loci, alleles = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (20, 50)
names, seqs, _ = synth.synth_library(loci, alleles)
strs = [synth.codes_to_str(s) for s in seqs]
dev = torch.device("cuda:0")


def pack(codes):                               # codes [m, 150] uint8 (4 = N) -> packed int64 [m, 8] (device layout of nb_pack_reads)
    m, L = codes.shape
    nwb, nwn = (L + 31) // 32, (L + 63) // 64
    isn = codes == 4
    two = torch.where(isn, torch.zeros_like(codes), codes).to(torch.int64)
    pad = torch.zeros((m, nwb * 32), dtype=torch.int64, device=codes.device); pad[:, :L] = two
    out = torch.empty((m, nwb + nwn), dtype=torch.int64, device=codes.device)
    out[:, :nwb] = (pad.view(m, nwb, 32) << (torch.arange(32, device=codes.device) * 2)[None, None, :]).sum(dim=2)
    npad = torch.zeros((m, nwn * 64), dtype=torch.int64, device=codes.device); npad[:, :L] = isn.to(torch.int64)
    out[:, nwb:] = (npad.view(m, nwn, 64) << torch.arange(64, device=codes.device)[None, None, :]).sum(dim=2)
    return out


def pairs(n, seed, frag=300, L=150, err=0.01, frac_random=0.10, chunk=1 << 20):
    """Fragments of 300 bases from a random allele and strand: r1 = its first 150 bases, r2 = the reverse complement of its last 150;
    1 % substitutions, 0.1 % N, 10 % of the pairs random (the distribution of synth.synth_reads, per mate)."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    lens = torch.tensor([len(s) for s in seqs], dtype=torch.int64, device=dev)
    offs = torch.zeros(len(seqs) + 1, dtype=torch.int64, device=dev); offs[1:] = torch.cumsum(lens, 0)
    flat = torch.from_numpy(np.concatenate(seqs).astype(np.uint8)).to(dev)
    ar = torch.arange(frag, device=dev)
    o1, o2 = [], []
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        al = torch.randint(0, len(seqs), (m,), generator=g, device=dev)
        st = (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * (lens[al] - frag + 1)).to(torch.int64)
        f = flat[(offs[al] + st)[:, None] + ar[None, :]]
        rev = torch.rand(m, generator=g, device=dev) < 0.5
        f = torch.where(rev[:, None], (3 - f).flip(1), f)
        a, b = f[:, :L], (3 - f[:, frag - L:]).flip(1)
        outs = []
        for x in (a, b):
            sub = torch.rand((m, L), generator=g, device=dev) < err
            x = torch.where(sub, (x + torch.randint(1, 4, (m, L), generator=g, device=dev, dtype=torch.uint8)) & 3, x)
            rnd = torch.rand(m, generator=g, device=dev) < frac_random / 2
            x = torch.where(rnd[:, None], torch.randint(0, 4, (m, L), generator=g, device=dev, dtype=torch.uint8), x)
            x = torch.where(torch.rand((m, L), generator=g, device=dev) < 0.001, torch.full_like(x, 4), x)
            outs.append(pack(x))
        o1.append(outs[0]); o2.append(outs[1])
    return torch.cat(o1), torch.cat(o2)


r1, r2 = pairs(n, 7)
ix = engine.Index(strs, 20)
ctx = engine.Context(0, max_classes=1 << 22)
ctx.upload(ix)
cfg = engine.make_config()
oc = torch.empty(n, dtype=torch.int32, device=dev); oi = torch.empty(n, dtype=torch.int32, device=dev)
ms = []
for it in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ctx.align_batch(cfg, r1, None, 150, oc, oi, None, mates=r2, mate_lens=None, mate_max_len=150, n=n)
    b.record(); ctx.sync()
    ms.append(a.elapsed_time(b))
print(f"{n} pairs: {np.mean(ms[1:]):.2f} ms per batch = {n / np.mean(ms[1:]) / 1e3:.0f} M pairs/s; kernels: {ctx.last_kernels()}; reasons: {ctx.reasons().tolist()}")

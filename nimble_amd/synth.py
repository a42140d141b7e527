"""Seeded synthetic HLA-like libraries and reads (SURVEY.md §8d recipe). No external data; numpy PCG64 only."""
import os

import numpy as np

LIB_SEED = 20260109
BASES = np.frombuffer(b"ACGTN", dtype=np.uint8)


def synth_library(n_loci, alleles_per_locus, locus_len=1098, seed=LIB_SEED, min_rate=0.005, max_rate=0.03):
    """Returns (names, seqs_codes list of uint8 arrays, lineage list). Each locus: random ancestor; alleles = ancestor with
    i.i.d. substitutions at a per-allele rate drawn uniformly from [min_rate, max_rate]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    names, seqs, lineage = [], [], []
    for locus in range(n_loci):
        anc = rng.integers(0, 4, size=locus_len, dtype=np.uint8)
        for a in range(alleles_per_locus):
            rate = rng.uniform(min_rate, max_rate)
            s = anc.copy()
            hit = rng.random(locus_len) < rate
            s[hit] = (s[hit] + rng.integers(1, 4, size=int(hit.sum()), dtype=np.uint8)) & 3
            names.append(f"L{locus:03d}*{a + 1:03d}")
            seqs.append(s)
            lineage.append(f"L{locus:03d}")
    return names, seqs, lineage


def codes_to_str(codes):
    return BASES[np.asarray(codes, dtype=np.uint8)].tobytes().decode()


def synth_reads(seqs, n, read_len, seed, frac_random=0.10, err=0.01, n_rate=0.001):
    """codes[n, read_len] uint8 (0..3, 4 = N): 90 % sampled from (allele, offset, strand) with substitution error,
    10 % uniform random; 0.1 % of bases replaced by N."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    if (lens < read_len).any():
        raise ValueError("reference sequence shorter than read_len")
    offs = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    flat = np.concatenate(seqs).astype(np.uint8)
    allele = rng.integers(0, len(seqs), size=n)
    start = (rng.random(n) * (lens[allele] - read_len + 1)).astype(np.int64)
    idx = (offs[allele] + start)[:, None] + np.arange(read_len, dtype=np.int64)[None, :]
    reads = flat[idx]
    rev = rng.random(n) < 0.5
    reads[rev] = (3 - reads[rev])[:, ::-1]
    sub = rng.random((n, read_len)) < err
    reads[sub] = (reads[sub] + rng.integers(1, 4, size=int(sub.sum()), dtype=np.uint8)) & 3
    rnd = rng.random(n) < frac_random
    reads[rnd] = rng.integers(0, 4, size=(int(rnd.sum()), read_len), dtype=np.uint8)
    reads[rng.random((n, read_len)) < n_rate] = 4
    return reads


def pack_codes(codes):
    """Vectorised twin of nb_pack_reads for fixed-length code arrays: codes[n, L] -> words[n, stride] uint64."""
    codes = np.asarray(codes, dtype=np.uint8)
    n, L = codes.shape
    nwb, nwn = (L + 31) // 32, (L + 63) // 64
    out = np.zeros((n, nwb + nwn), dtype=np.uint64)
    isn = codes == 4
    two = np.where(isn, 0, codes).astype(np.uint64)
    pad = np.zeros((n, nwb * 32), dtype=np.uint64)
    pad[:, :L] = two
    sh = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, None, :]
    out[:, :nwb] = np.bitwise_or.reduce(pad.reshape(n, nwb, 32) << sh, axis=2)
    npad = np.zeros((n, nwn * 64), dtype=np.uint64)
    npad[:, :L] = isn
    sh1 = np.arange(64, dtype=np.uint64)[None, None, :]
    out[:, nwb:] = np.bitwise_or.reduce(npad.reshape(n, nwn, 64) << sh1, axis=2)
    return out


def reads_to_ascii(codes):
    """codes[n, L] -> (blob bytes, offsets int64[n+1])."""
    codes = np.asarray(codes, dtype=np.uint8)
    n, L = codes.shape
    return BASES[codes].tobytes(), np.arange(n + 1, dtype=np.int64) * L


def synth_reads_packed_torch(seqs, n, read_len, seed, device, chunk=1 << 20, frac_random=0.10, err=0.01, n_rate=0.001,
                             return_codes_of_first=0):
    """Bench-scale twin of synth_reads + pack_codes on the GPU (torch Philox generator, seeded): returns the packed
    int64 tensor [n, stride] in the device layout; optionally also the uint8 codes of the first `return_codes_of_first`
    reads (for a parity spot check against the oracle). Same distribution as synth_reads, different PRNG stream."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    lens = torch.tensor([len(s) for s in seqs], dtype=torch.int64, device=device)
    offs = torch.zeros(len(seqs) + 1, dtype=torch.int64, device=device)
    offs[1:] = torch.cumsum(lens, 0)
    flat = torch.from_numpy(np.concatenate(seqs).astype(np.uint8)).to(device)
    nwb, nwn = (read_len + 31) // 32, (read_len + 63) // 64
    out = torch.empty((n, nwb + nwn), dtype=torch.int64, device=device)
    ar = torch.arange(read_len, dtype=torch.int64, device=device)
    sh2 = (torch.arange(32, dtype=torch.int64, device=device) * 2)[None, None, :]
    sh1 = torch.arange(64, dtype=torch.int64, device=device)[None, None, :]
    first_codes = None
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        allele = torch.randint(0, len(seqs), (m,), generator=g, device=device)
        if os.environ.get("NB_SYNTH_SORTED"):          # locality experiment (profiles/r01_tuning_notes.md): reads ordered by origin
            allele = torch.sort(allele).values
        start = (torch.rand(m, generator=g, device=device, dtype=torch.float64) * (lens[allele] - read_len + 1)).to(torch.int64)
        reads = flat[(offs[allele] + start)[:, None] + ar[None, :]]
        rev = torch.rand(m, generator=g, device=device) < 0.5
        reads = torch.where(rev[:, None], (3 - reads).flip(1), reads)
        sub = torch.rand((m, read_len), generator=g, device=device) < err
        reads = torch.where(sub, (reads + torch.randint(1, 4, (m, read_len), generator=g, device=device, dtype=torch.uint8)) & 3, reads)
        rnd = torch.rand(m, generator=g, device=device) < frac_random
        reads = torch.where(rnd[:, None], torch.randint(0, 4, (m, read_len), generator=g, device=device, dtype=torch.uint8), reads)
        isn = torch.rand((m, read_len), generator=g, device=device) < n_rate
        reads = torch.where(isn, torch.full_like(reads, 4), reads)
        if lo < return_codes_of_first:                 # codes of the first reads, across chunks
            part = reads[:return_codes_of_first - lo].cpu().numpy()
            first_codes = part if first_codes is None else np.concatenate([first_codes, part])
        two = torch.where(isn, torch.zeros_like(reads), reads).to(torch.int64)
        pad = torch.zeros((m, nwb * 32), dtype=torch.int64, device=device)
        pad[:, :read_len] = two
        out[lo:lo + m, :nwb] = (pad.view(m, nwb, 32) << sh2).sum(dim=2)       # disjoint bit fields: sum == or
        npad = torch.zeros((m, nwn * 64), dtype=torch.int64, device=device)
        npad[:, :read_len] = isn.to(torch.int64)
        out[lo:lo + m, nwb:] = (npad.view(m, nwn, 64) << sh1).sum(dim=2)
    return (out, first_codes) if return_codes_of_first else out

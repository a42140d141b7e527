"""Shared helpers for the parity tests: run the HIP path through the C-ABI and the oracle on the same inputs."""
import numpy as np

from nimble_amd import engine, synth


def run_gpu(index, cfg, words, lens, max_len, max_classes=1 << 16, device=0, mate_words=None, mate_lens=None, mate_max_len=0):
    import torch
    ctx = engine.Context(device, max_classes=max_classes)
    ctx.upload(index)
    dev = torch.device("cuda", device)
    n = words.shape[0]
    d_reads = torch.from_numpy(words.view(np.int64)).to(dev)
    d_lens = torch.from_numpy(lens.view(np.int16)).to(dev) if lens is not None else None
    out_class = torch.empty(n, dtype=torch.int32, device=dev)
    out_info = torch.empty(n, dtype=torch.int32, device=dev)
    out_cov = torch.empty(n, dtype=torch.int64, device=dev)
    d_mates = torch.from_numpy(mate_words.view(np.int64)).to(dev) if mate_words is not None else None
    d_mlens = torch.from_numpy(mate_lens.view(np.int16)).to(dev) if mate_lens is not None else None
    ctx.align_batch(cfg, d_reads, d_lens, max_len, out_class, out_info, out_cov, mates=d_mates, mate_lens=d_mlens,
                    mate_max_len=mate_max_len, n=n)
    ctx.sync()
    rows, counts = ctx.classes()
    res = {"class": out_class.cpu().numpy().view(np.uint32), "rows": rows, "counts": counts, "reasons": ctx.reasons()}
    res.update(engine.decode_info(out_info.cpu().numpy().view(np.uint32)))
    res["cov"] = engine.decode_cov(out_cov.cpu().numpy().view(np.uint64))
    W = index.words
    sets = np.zeros((n, W), dtype=np.uint64)
    called = res["class"] != engine.NO_CLASS
    sets[called] = rows[res["class"][called]]
    res["set_bits"] = sets
    ctx.close()
    return res


def assert_same(gpu, ora, what=""):
    for key in ("reason", "score", "orientation", "mates"):
        bad = np.nonzero(np.asarray(gpu[key]).astype(np.int64) != np.asarray(ora[key]).astype(np.int64))[0]
        assert bad.size == 0, f"{what}: {key} differs at {bad[:10]} gpu={np.asarray(gpu[key])[bad[:10]]} oracle={np.asarray(ora[key])[bad[:10]]}"
    bad = np.nonzero((gpu["cov"] != ora["cov"]).any(axis=1))[0]
    assert bad.size == 0, f"{what}: coverage differs at {bad[:10]} gpu={gpu['cov'][bad[:5]]} oracle={ora['cov'][bad[:5]]}"
    bad = np.nonzero((gpu["set_bits"] != ora["set_bits"]).any(axis=1))[0]
    assert bad.size == 0, f"{what}: result sets differ at {bad[:10]}"
    # counts: multiset {set -> count} from the oracle's per-read sets vs the device class table
    called = ora["reason"] == 0
    want = {}
    for row in ora["set_bits"][called]:
        want[row.tobytes()] = want.get(row.tobytes(), 0) + 1
    got = {r.tobytes(): int(c) for r, c in zip(gpu["rows"], gpu["counts"]) if c}
    assert got == want, f"{what}: class counts differ ({len(got)} vs {len(want)} classes)"
    hist = np.bincount(ora["reason"], minlength=len(gpu["reasons"]))
    assert (hist == gpu["reasons"]).all(), f"{what}: reason histogram {gpu['reasons']} vs {hist}"


def library_strings(n_loci, alleles, locus_len=1098, seed=synth.LIB_SEED):
    names, seqs, lineage = synth.synth_library(n_loci, alleles, locus_len, seed)
    return names, seqs, [synth.codes_to_str(s) for s in seqs], lineage

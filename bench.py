#!/usr/bin/env python
"""bench.py — M reads/s aligned on MI355X (BASELINE.json metric), reads resident in HBM.

Workload (BASELINE config 2): 10 M synthetic 150-bp single-end reads vs a 1 000-sequence HLA-like library
(20 loci x 50 alleles, SURVEY.md §8d recipe), per GPU. A "step" = one pass of the alignment kernel over the whole
batch. N > 1: one process per GPU (torchrun), reads sharded (each rank its own seeded batch, weak scaling), index
replicated, no collective while aligning; the timed region ends with the dictionary agreement (all-gather of 16-byte class
digests) + ONE RCCL all-reduce of the per-class count vector (nimble_amd/merge.py). Prints ONE JSON line on rank 0.

roofline.achieved = algorithmic bytes per read (ceil(2L/8) + ceil(L/8) + 8 = 65 B at L = 150) x reads per
launch / average kernel duration measured with HIP events on the launch stream. cpu_baseline = the C++ oracle
(oracle/nb_oracle.cpp, "port": a restatement, NOT the Rust nimble-aligner, which is unobtainable offline) timed on
rank 0's host cores on a bounded sample of the same batch.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# BASELINE.json configs that fit one GPU (SURVEY.md §8d sizes). c2 is the configuration the metric is quoted on.
PRESETS = {
    "c2": dict(reads=10_000_000, read_len=150, loci=20, alleles=50, seed=2, what="configs[1]: 150bp single-end vs 1k-seq HLA-like library"),
    "c3": dict(reads=100_000_000, read_len=91, loci=40, alleles=38, seed=3, what="configs[2]: 10x R2 (91bp cDNA) vs MHC+KIR-like library, per-read calls kept for per-cell counting"),
    "c5": dict(reads=100_000_000, read_len=150, loci=100, alleles=100, seed=5, max_classes=1 << 23, what="configs[4]: 150bp single-end vs 10k-seq library (index exceeds the Infinity Cache)"),
}


def cpu_baseline(oix, cfg, codes, strs, args, t_oidx):
    """The CPU path timed beside the GPU number (rank 0, N = 1). Two kinds:
      "port"       the C++ oracle (oracle/nb_oracle.cpp, a literal restatement of docs/SPEC.md — NOT the Rust nimble-aligner) on
                   min(cores, 32) threads: a calibration slice fixes the sample so that the ONE timed call runs about
                   --cpu-seconds (>= 3 s): thread start-up and the work quantum no longer dominate, the figure repeats to a
                   few per cent; reads/s per thread is reported next to the total;
      "reference"  only if NIMBLE_ALIGNER_BIN names an executable: the reference's own aligner timed with the argv the
                   reference builds (/root/reference/nimble/__main__.py:177-195). Never downloaded, never searched for."""
    from nimble_amd import synth
    cores = os.cpu_count() or 1
    ref = time_reference_binary(os.environ.get("NIMBLE_ALIGNER_BIN"), codes, strs, cores)
    if ref is not None:
        return ref
    threads = args.cpu_threads if args.cpu_threads > 0 else min(cores, 32)
    cal = min(codes.shape[0], max(2_000 * threads, 20_000))
    blob, off = synth.reads_to_ascii(codes[:cal])
    oix.align(cfg, blob, off, n_threads=threads)                       # warm-up: page in the index, spin the pool once
    t0 = time.perf_counter()
    oix.align(cfg, blob, off, n_threads=threads)
    rate = cal / max(time.perf_counter() - t0, 1e-6)
    nc = int(min(codes.shape[0], args.cpu_sample, max(cal, rate * args.cpu_seconds)))
    blob, off = synth.reads_to_ascii(codes[:nc])
    t0 = time.perf_counter()
    oix.align(cfg, blob, off, n_threads=threads)
    dt = time.perf_counter() - t0
    return {"value": nc / dt / 1e6, "unit": "M reads/s", "cores": threads, "kind": "port", "host_cores": cores,
            "reads_per_s_per_thread": nc / dt / threads,
            "sample": f"first {nc} reads of the same batch in ONE call of {dt:.1f} s wall on {threads} threads (calibrated on {cal} reads; "
                      f"alignment only, oracle index build {t_oidx:.1f} s not counted); C++ restatement of docs/SPEC.md, not the Rust nimble-aligner"}


def secondary_roofline(l2_requests, miss_requests, table_mb, kernel_s, n_reads, mixed=False):
    """Gather-rate ceiling from the committed microbenchmark (16-byte random requests, best over the occupancies measured): the row of
    the smallest table at least as large as this index's k-mer table. None without a PMC pass of this workload."""
    path = os.path.join(ROOT, "profiles", "r02_gather_microbench.jsonl")
    if not os.path.exists(path) or not (l2_requests or miss_requests):
        return None
    rows = [json.loads(l) for l in open(path) if l.strip()]
    sizes = sorted({r["table_MB"] for r in rows})
    size = next((s for s in sizes if s >= table_mb), sizes[-1])
    peak = max(r["G_requests_per_s"] for r in rows if r["table_MB"] == size and r["req_bytes"] == 16)
    if l2_requests and mixed:
        # anchor kernels: the requests go to several arrays (key words and text in the L2s, table entries around them, the sparse table
        # beyond): the L1->L2 side is priced against the rate of an L2-resident table (the fastest the path goes), the L2-miss side against
        # the rate of a table in the Infinity Cache
        l2_peak = max(r["G_requests_per_s"] for r in rows if r["req_bytes"] == 16 and 2.0 <= r["table_MB"] <= 4.0)
        ic_peak = max(r["G_requests_per_s"] for r in rows if r["req_bytes"] == 16 and 40.0 <= r["table_MB"] <= 60.0)
        ach, miss = l2_requests / kernel_s / 1e9, (miss_requests or 0) / kernel_s / 1e9
        return {"bound": "l1_to_l2_gather_requests", "achieved": ach, "peak": l2_peak, "unit": "G requests/s", "frac": ach / l2_peak,
                "requests_per_read": l2_requests / n_reads, "l2_misses_per_read": (miss_requests / n_reads) if miss_requests else None,
                "l2_miss_side": {"achieved": miss, "peak": ic_peak, "frac": miss / ic_peak, "unit": "G requests/s"},
                "peak_source": "scripts/gather_bench.hip, random 16-B requests from an L2-resident table (2.5-4 MB) / from a 49 MB table in the Infinity Cache "
                               "(profiles/r02_gather_microbench.jsonl)"}
    if l2_requests:
        ach = l2_requests / kernel_s / 1e9
        return {"bound": "l1_to_l2_gather_requests", "achieved": ach, "peak": peak, "unit": "G requests/s", "frac": ach / peak,
                "requests_per_read": l2_requests / n_reads, "l2_misses_per_read": (miss_requests / n_reads) if miss_requests else None,
                "peak_source": f"scripts/gather_bench.hip, random 16-B requests from a {size:g} MB table (profiles/r02_gather_microbench.jsonl)"}
    fabric = max(r["G_requests_per_s"] for r in rows if r["table_MB"] == sizes[-2 if len(sizes) > 1 else -1] and r["req_bytes"] == 32)
    ach = miss_requests / kernel_s / 1e9
    return {"bound": "l2_miss_requests", "achieved": ach, "peak": fabric, "unit": "G requests/s", "frac": ach / fabric, "misses_per_read": miss_requests / n_reads,
            "peak_source": "scripts/gather_bench.hip, random 32-B requests from a table beyond L2 (profiles/r02_gather_microbench.jsonl)"}


def time_reference_binary(path, codes, strs, cores, max_reads=2_000_000):
    """if a nimble-aligner binary is supplied through NIMBLE_ALIGNER_BIN, time it on a sample of the same batch with
    the reference's argv: [bin, "--input", fastq, "-c", N, "--strand_filter", "unstranded", "-r", lib.json, "-o", out]
    (/root/reference/nimble/__main__.py:177-189; bulk FASTQ input, one library). Returns the cpu_baseline object or None."""
    if not path:
        return None
    if not (os.path.isfile(path) and os.access(path, os.X_OK)):
        print(f"[bench] NIMBLE_ALIGNER_BIN={path!r} is not an executable file: ignored", file=sys.stderr)
        return None
    import subprocess
    import tempfile
    from nimble_amd import library, synth
    n = int(min(codes.shape[0], max_reads))
    with tempfile.TemporaryDirectory() as d:
        fa, lib, fq, out = (os.path.join(d, x) for x in ("lib.fasta", "lib.json", "reads.fastq", "out.tsv"))
        with open(fa, "w") as f:
            for i, s in enumerate(strs):
                f.write(f">seq{i}\n{s}\n")
        library.generate(fa, None, lib)
        L = codes.shape[1]
        with open(fq, "wb") as f:
            qual = b"I" * L
            text = synth.BASES[codes[:n]]
            for i in range(n):
                f.write(b"@r%d\n" % i + text[i].tobytes() + b"\n+\n" + qual + b"\n")
        argv = [path, "--input", fq, "-c", str(cores), "--strand_filter", "unstranded", "-r", lib, "-o", out]
        t0 = time.perf_counter()
        rc = subprocess.run(argv, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
        dt = time.perf_counter() - t0
    if rc != 0:
        print(f"[bench] NIMBLE_ALIGNER_BIN exited with {rc}: falling back to the C++ port", file=sys.stderr)
        return None
    return {"value": n / dt / 1e6, "unit": "M reads/s", "cores": cores, "kind": "reference",
            "sample": f"first {n} reads of the same batch as FASTQ, whole process ({dt:.1f} s wall incl. library load and TSV write), argv of "
                      "nimble/__main__.py:177-189"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(PRESETS), default="c2", help="BASELINE.json workload preset (default: the metric's config)")
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("--loci", type=int, default=None)
    ap.add_argument("--alleles", type=int, default=None)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--foreign", type=float, default=None, help="fraction of uniform random reads in the synthetic batch (default: synth's 0.10); sweeps only")
    ap.add_argument("--mismatches", type=int, default=0, help="num_mismatches of the library config (default 0: the metric's configuration); sweeps only")
    ap.add_argument("--err", type=float, default=None, help="substitution rate of the library-derived reads (default: synth's 0.01); sweeps only")
    ap.add_argument("--cpu-sample", type=int, default=12_000_000, help="upper bound of the reads timed on the CPU baseline (rank 0, N=1 only); the "
                    "actual sample is sized from a calibration slice so that the timed call runs about --cpu-seconds")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="target wall time of the timed CPU-baseline call")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (default min(host cores, 32))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-sample", type=int, default=20_000)
    ap.add_argument("--per-cell", action="store_true",
                    help="c3 only, after the timed region: synthetic cell/UMI ids per read, merge the per-read calls on the GPU and run the "
                         "report reduction (fused per-cell counting); reported under 'per_cell', not part of 'value'")
    ap.add_argument("--backend", default=os.environ.get("NB_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 logic on a 1-GPU box)")
    args = ap.parse_args()
    preset = PRESETS[args.config]
    for key in ("reads", "read_len", "loci", "alleles"):
        if getattr(args, key) is None:
            setattr(args, key, preset[key])

    import torch
    import torch.distributed as dist
    from nimble_amd import engine, merge, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_rank %= max(1, torch.cuda.device_count())     # gloo rehearsal: several ranks may share the one GPU of a dev box
    # NB_BENCH_REHEARSE_DIST=1 under torchrun with ONE rank: take the several-rank path (RCCL communicator, barriers, end-of-job reduce with its
    # collectives) on a one-GPU box; the line it prints is a rehearsal, not a measurement of the metric
    multi = world > 1 or os.environ.get("NB_BENCH_REHEARSE_DIST") == "1"
    if multi:
        if world == 1:
            os.environ["NB_MERGE_REHEARSE"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    if args.gpus != world:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")   # where collective buffers live

    # ---- library + index (identical on every rank) ----
    t0 = time.time()
    names, seqs, _ = synth.synth_library(args.loci, args.alleles)
    strs = [synth.codes_to_str(s) for s in seqs]
    index = engine.Index(strs, args.k)
    stats = index.stats()
    ctx = engine.Context(local_rank, max_classes=preset.get("max_classes", 1 << 21))
    ctx.upload(index)
    t_index = time.time() - t0

    # ---- reads: generated on the GPU, seed 2 + rank (seed 2 for c2, 4 + rank for c4) ----
    L, n = args.read_len, args.reads
    seed = preset["seed"] if not multi else 4 + rank
    n_keep = max(args.parity_sample, 0 if (args.no_cpu_baseline or multi) else args.cpu_sample) if rank == 0 else 0
    t0 = time.time()
    shape_kw = {}
    if args.foreign is not None:
        shape_kw["frac_random"] = args.foreign
    if args.err is not None:
        shape_kw["err"] = args.err
    res = synth.synth_reads_packed_torch(seqs, n, L, seed, dev, return_codes_of_first=min(n_keep, n), **shape_kw)
    d_reads, first_codes = res if n_keep else (res, None)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    out_class = torch.empty(n, dtype=torch.int32, device=dev)
    out_info = torch.empty(n, dtype=torch.int32, device=dev)
    cfg = engine.make_config(k=args.k, num_mismatches=args.mismatches)
    stream = torch.cuda.current_stream(dev)

    def step():
        ctx.align_batch(cfg, d_reads, None, L, out_class, out_info, None, stream=stream, n=n)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if multi:
        # the end-of-job reduce runs once, inside the timed region: its first call in a process loads a dozen tensor-op kernels and opens
        # the point-to-point channels of the communicator (hundreds of ms, nothing to do with the job): do that here, on the warm-up's table
        w_rows, w_counts = ctx.classes_device(stream)
        merge.reduce_class_counts_hashed(w_rows, w_counts, device=cdev, compute_device=dev)
        torch.cuda.synchronize()

    # ---- timed region: exactly K steps (+ the end-of-job count reduce) ----
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(stream)
        step()
        b.record(stream)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_steps = time.perf_counter() - t0
    # end-of-job reduce (nimble_amd/merge.py): dictionary agreement by 16-byte digests, ONE all-reduce of the dense count vector, rows to rank 0.
    # Several ranks: the class table stays on the device until the merged rows reach the root (internal numbering, converted there).
    if multi:
        t_red = time.perf_counter()
        d_rows, d_counts = ctx.classes_device(stream)
        glob_rows, glob_counts = merge.reduce_class_counts_hashed(d_rows, d_counts, device=cdev, compute_device=dev)
        if rank == 0:
            glob_rows = ctx.rows_to_external(glob_rows)
        rows, counts = None, None
        t_red = time.perf_counter() - t_red
    else:
        rows, counts = ctx.classes()
        glob_rows, glob_counts = rows, counts
    t_sync = time.perf_counter()
    torch.cuda.synchronize()
    t_tail = time.perf_counter()
    t_sync = t_tail - t_sync
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t_tail = time.perf_counter() - t_tail
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = [a.elapsed_time(b) for a, b in evs]

    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return

    if rows is None:
        rows, counts = ctx.classes()           # rank 0's own table, for the parity spot check below (outside the timed region)
    total_reads = n * world * args.steps
    value = total_reads / elapsed / 1e6
    bytes_per_read = (2 * L + 7) // 8 + (L + 7) // 8 + 8
    if stats["device_bytes"] > 256 << 20:          # index not cache-resident (c5): SURVEY.md §8d adds one 16-B slot per k-mer position
        bytes_per_read += (L - args.k + 1) * 16
    avg_kernel_s = float(np.mean(kern_ms)) / 1e3
    achieved = bytes_per_read * n / avg_kernel_s / 1e9
    reasons = ctx.reasons()

    # ---- parity spot check + CPU baseline (not timed above) ----
    parity = None
    cpu = None
    if first_codes is not None:
        from oracle import oracle
        t0 = time.time()
        oix = oracle.OracleIndex(strs, args.k)
        t_oidx = time.time() - t0
        ns = min(args.parity_sample, n, first_codes.shape[0])     # (the generator hands back codes of its first chunk only)
        if ns:
            blob, off = synth.reads_to_ascii(first_codes[:ns])
            ora = oix.align(cfg, blob, off, n_threads=os.cpu_count() or 1)
            cls = out_class[:ns].cpu().numpy().view(np.uint32)
            info = engine.decode_info(out_info[:ns].cpu().numpy().view(np.uint32))
            sets = np.zeros((ns, index.words), dtype=np.uint64)
            called = cls != engine.NO_CLASS
            sets[called] = rows[cls[called]]
            parity = bool((sets == ora["set_bits"]).all() and (info["score"] == ora["score"]).all()
                          and (info["reason"] == ora["reason"]).all() and (info["orientation"] == ora["orientation"]).all())
        if not multi and not args.no_cpu_baseline and args.cpu_sample:
            cpu = cpu_baseline(oix, cfg, first_codes, strs, args, t_oidx)

    # HBM-side traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the figure comes
    # from the committed rocprofv3 PMC passes of the SAME workload (scripts/profile_gpu.sh -> profiles/*_traffic.json).
    traffic, traffic_src, miss_requests, l2_requests, sq = None, None, None, None, None
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            t = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        # (counters of ANOTHER kernel say nothing about this run: the profile must be of the kernel that ran here)
        if (t.get("reads_per_gpu"), t.get("read_len"), t.get("n_seqs"), t.get("k"), t.get("kernel")) == (n, L, len(strs), args.k, ctx.last_kernels()[0]) \
                and "traffic_bytes_per_launch" in t:
            traffic, traffic_src = t["traffic_bytes_per_launch"], os.path.basename(f)
            miss_requests = t.get("TCC_MISS_sum")
            l2_requests = t.get("TCP_TCC_READ_REQ_sum")
            sq = t.get("sq_per_read")

    # ---- optional: per-cell counts for the 10x workload (BASELINE configs[2] asks for them), outside the timed region ----
    per_cell = None
    if args.per_cell and not multi:
        from nimble_amd import fused
        import tempfile
        t0 = time.perf_counter()
        g = torch.Generator(device=dev); g.manual_seed(33)
        n_cells = 10_000                                                  # 10 000 cells, a few reads per UMI
        # reads of one UMI come from one molecule, i.e. share their call: derive the molecule id from the read's class
        umi_id = (out_class.to(torch.int64) & 0xFFFFFFFF) * 64 + torch.randint(0, 64, (n,), generator=g, device=dev)
        cb_ids = (umi_id * 2654435761) % n_cells                          # a UMI belongs to one cell
        umi_codes = (umi_id * 40503) % (1 << 24)                          # 12-base UMI = 24 bits
        # class -> feature string id on the host (nb_class_features: names of the members; classes with > 10 features are rejected:
        # max_hits_to_report)
        import ctypes as ct
        uniq = sorted(set(names))
        fidx = {f: i for i, f in enumerate(uniq)}
        enc = [f.encode() for f in uniq]
        _, sorted_feats, creasons = engine.class_features(rows, np.array([fidx[f] for f in names], dtype=np.int32), (ct.c_char_p * len(enc))(*enc),
                                                          len(enc), 10, 0, max(len(e) for e in enc))
        feat_strings, cfid, index_of = [], np.full(rows.shape[0], -1, dtype=np.int64), {}
        for c, (key, rs) in enumerate(zip(sorted_feats, creasons.tolist())):
            if rs == 0 and key:
                fid = index_of.setdefault(key, len(feat_strings))
                if fid == len(feat_strings):
                    feat_strings.append(key)
                cfid[c] = fid
        t_map = time.perf_counter() - t0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cls_t = out_class.to(torch.int64) & 0xFFFFFFFF
        called = cls_t != engine.NO_CLASS
        fid_t = torch.full((n,), -1, dtype=torch.int64, device=dev)
        fid_t[called] = torch.from_numpy(cfid).to(dev)[cls_t[called]]
        ok = fid_t >= 0
        merged = fused.merge_rows(cb_ids[ok], umi_codes[ok], fid_t[ok], torch.ones(int(ok.sum()), dtype=torch.int64, device=dev))
        cols = [m.cpu().numpy() for m in merged]
        torch.cuda.synchronize(); t_merge = time.perf_counter() - t0
        t0 = time.perf_counter()
        with tempfile.TemporaryDirectory() as d:
            outp = os.path.join(d, "per_cell.tsv")
            from nimble_amd.report import report_rows
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):               # stdout carries exactly one JSON line
                report_rows(cols[0], cols[1], cols[2], cols[3], [f"CELL{i:05d}" for i in range(n_cells)], feat_strings, outp)
            n_out = sum(1 for _ in open(outp))
        per_cell = {"reads": int(n), "called_rows": int(ok.sum()), "merged_rows": int(cols[0].shape[0]), "class_to_feature_s": t_map,
                    "gpu_merge_s": t_merge, "reduce_s": time.perf_counter() - t0, "output_rows": n_out, "cells": n_cells}

    line = {
        "metric": "M reads/s aligned (150bp, 1k-seq HLA lib)" if args.config == "c2" else f"M reads/s aligned ({args.config})", "value": value, "unit": "M reads/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"{n} x {L}bp single-end reads per GPU vs {len(strs)}-seq HLA-like library; BASELINE {preset['what']}; k={args.k}",
                   "reads_per_gpu": n, "read_len": L, "n_seqs": len(strs), "k": args.k, "parallelism": f"reads sharded x{world}, index replicated"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                     "traffic_note": (f"bytes per launch, (FETCH_SIZE + WRITE_SIZE) KB from {traffic_src}: L2<->fabric requests (64 B per L2 miss; for scattered "
                                      "16..64-B gathers FETCH_SIZE was calibrated against a known request count, profiles/r02_fetch_calibration.txt: it counts "
                                      "64 B per missing request, no x2 correction applies); algorithmic bytes per launch = " + str(bytes_per_read * n)) if traffic else None,
                     "kernel": ctx.last_kernels()[0], "kernels_per_step": ctx.last_kernels(), "kernel_ms_avg": avg_kernel_s * 1e3,
                     "kernel_ms_note": "HIP events around one nb_align_batch call = every launch in kernels_per_step (the anchor kernel twice: all reads, then "
                                       "the stragglers' retry list); compare with the per-step total of the nb:: kernels in the rocprofv3 summary",
                     "algorithmic_bytes_per_read": bytes_per_read},
        # The nearest memory-side ceiling: the rate at which the chip serves RANDOM 16-byte gathers (one cuckoo bucket = one dwordx4 = one
        # L1->L2 request) from a table of this size, MEASURED by the standalone microbenchmark scripts/gather_bench.hip
        # (profiles/r02_gather_microbench.jsonl: 264 G requests/s from a 3 MB L2-resident table, 61 G from 49 MB in the Infinity Cache,
        # 55 G from 446 MB in HBM). achieved = L1->L2 read requests of the kernel (TCP_TCC_READ_REQ_sum of the committed PMC pass) / time.
        "secondary_roofline": secondary_roofline(l2_requests, miss_requests,
                                                 (stats["ph_table_bytes"] if ctx.last_kernels()[0].startswith(("nb_lane", "nb_anchor")) else stats["table_bytes"]) / 1e6, avg_kernel_s, n,
                                                 mixed=ctx.last_kernels()[0].startswith("nb_anchor")),
        # what binds the kernel (profiles/r01_tuning_notes.md): instruction issue. Instructions issued per read (SQ counters of the
        # committed PMC pass) against the issue slots a SIMD has per read at this occupancy: above 1.0 = ports saturated.
        "issue": ({"instructions_per_read": sq["SQ_ACTIVE_INST_ANY"], "valu": sq["SQ_INSTS_VALU"], "salu": sq["SQ_INSTS_SALU"], "vmem": sq["SQ_INSTS_VMEM"],
                   "simd_slots_per_read": sq["SQ_WAVE_CYCLES"] / sq["waves_per_simd"],
                   "issue_utilisation": sq["SQ_ACTIVE_INST_ANY"] / (sq["SQ_WAVE_CYCLES"] / sq["waves_per_simd"]),
                   "wait_for_issue_frac": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"], "source": traffic_src} if sq else None),
        "cpu_baseline": cpu,
        # identical to the oracle on a sample of this very batch: self-consistency of the two implementations of docs/SPEC.md (HIP vs
        # C++ restatement), NOT parity with the Rust nimble-aligner, which stays unpinned (no binary, source or vectors offline)
        "oracle_self_consistency_ok": parity,
        "index": {"kmers": stats["n_kmers"], "colours": stats["n_colours"], "table_MB": stats["table_bytes"] / 1e6, "compact_table": bool(stats["compact"]),
                  "perfect_hash_table_MB": stats["ph_table_bytes"] / 1e6, "perfect_hash_first_level_KB": stats["ph_pilot_bytes"] / 1e3,
                  "device_MB": stats["device_bytes"] / 1e6, "build_s": t_index},
        "per_cell": per_cell,
        "classes": int(glob_rows.shape[0]), "called_frac": float(reasons[0] / max(1, reasons.sum())), "readgen_s": t_gen,
    }
    if multi:
        line["end_of_job_reduce_ms"] = t_red * 1e3          # inside the timed region, once per job (K steps)
        line["timed_region_ms"] = {"steps_issued": t_issue * 1e3, "steps_done": t_steps * 1e3, "reduce": t_red * 1e3, "device_sync": t_sync * 1e3, "closing_barrier": t_tail * 1e3}
        if world == 1:
            line["rehearsal"] = "NB_BENCH_REHEARSE_DIST=1: several-rank path with one rank; not a measurement of the metric"
    print(json.dumps(line))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Command line: `python -m nimble_amd {generate,align,report,fastq-to-bam}` with the reference's flags.

Flag surface copied from /root/reference/nimble/__main__.py:389-431 (generate :389-392, align :394-401, report
:403-418, fastq-to-bam :424-431). `download` and `plot` exist in the reference but are outside the hot path (network
fetch of the Rust binary, HTML plotting): they are accepted and answered with an explanatory error."""
import argparse
import os
import sys

from .align import TenxOptions, align
from .library import generate
from .report import report


def build_parser():
    parser = argparse.ArgumentParser(description="nimble align")
    parser.add_argument("-v", "--version", action="version", version="nimble_amd 0.1 (MI355X engine, C-ABI 4)")
    sub = parser.add_subparsers(title="subcommands", dest="subcommand")

    g = sub.add_parser("generate")
    g.add_argument("--file", help="The file to process.", type=str, required=True)
    g.add_argument("--opt-file", help="The optional file to process.", type=str, default=None)
    g.add_argument("--output_path", help="The path to the output file.", type=str, required=True)

    a = sub.add_parser("align")
    a.add_argument("--reference", help="The reference genome to align to.", type=str, required=True)
    a.add_argument("--output", help="The path to the output file.", type=str, required=True)
    a.add_argument("--input", help="The input reads.", type=str, required=True, nargs="+")
    a.add_argument("-c", "--num_cores", help="The number of cores to use for alignment.", type=int, default=1)
    a.add_argument("--strand_filter", help="Filter reads based on strand information.", type=str, default="unstranded")
    a.add_argument("--trim", help="Configuration for trimming read-data, in the format <TARGET_LENGTH>:<STRICTNESS>, comma-separated, one entry for each passed library", type=str, default="")
    a.add_argument("--tmpdir", help="Path to a temporary directory for sorting .bam files", type=str, default=None)
    # extensions (neutral when absent): 10x R1 barcode reads straight from FASTQ, GPU choice
    a.add_argument("--tenx", help="Treat --input R1 R2 as 10x reads: R1 = cell barcode + UMI, R2 = cDNA.", action="store_true", default=False)
    a.add_argument("--map", help="Cell barcode whitelist file (one CB per line, .gz or plain text); enables Hamming-1 correction.", type=str, default=None)
    a.add_argument("--cb-length", help="Length of cell barcode (default: 16).", type=int, default=16)
    a.add_argument("--umi-length", help="Length of UMI (default: 12).", type=int, default=12)
    a.add_argument("--device", help="GPU index (default 0).", type=int, default=0)
    a.add_argument("--devices", help="Comma-separated GPU indices: one process per GPU (reads sharded, index replicated; counts reduced over RCCL, "
                                     "outputs identical to a single-GPU run). Under torchrun the ranks are taken from the environment instead.", type=str, default=None)
    a.add_argument("--report-output", help="Also write the per-cell counts `report` would produce from the per-read table (barcoded input only).", type=str, default=None)
    a.add_argument("--report-threshold", help="Threshold of the fused report (default: 0.05).", type=float, default=0.05)
    a.add_argument("--report-disable-thresholding", help="Disable thresholding in the fused report.", action="store_true", default=False)

    r = sub.add_parser("report")
    r.add_argument("-i", "--input", help="The input file.", type=str, required=True)
    r.add_argument("-o", "--output", help="The path to the output file.", type=str, required=True)
    r.add_argument("-s", "--summarize", help="CSV list of columns to summarize.", type=str, default=None)
    r.add_argument("-t", "--threshold", help="Proportional count threshold for filtering features (default: 0.05).", type=float, default=0.05)
    r.add_argument("--disable_thresholding", help="Disable the per-UMI proportional count thresholding algorithm.", action="store_true", default=False)

    f = sub.add_parser("fastq-to-bam")
    f.add_argument("--r1-fastq", help="Path to R1 FASTQ file.", type=str, required=True)
    f.add_argument("--r2-fastq", help="Path to R2 FASTQ file.", type=str, required=True)
    f.add_argument("--map", required=True, help="Cell barcode whitelist file (one CB per line, .gz or plain text)")
    f.add_argument("--output", help="Path for output BAM file.", type=str, required=True)
    f.add_argument("-c", "--num_cores", help="The number of cores to use for processing.", type=int, default=1)
    f.add_argument("--cb-length", help="Length of cell barcode (default: 16).", type=int, default=16)
    f.add_argument("--umi-length", help="Length of UMI (default: 12).", type=int, default=12)

    for name in ("download", "plot"):
        sub.add_parser(name, add_help=False)
    return parser


def _spawn_ranks(argv, devices):
    """`align --devices 0,1,...`: one child process per GPU, started BEFORE anything in this process touches a GPU; each child is the
    same command line with its rank in the environment (RANK / WORLD_SIZE / LOCAL_RANK / NB_DEVICES, rendezvous on 127.0.0.1). Returns
    the largest exit code (the reference's only status channel, __main__.py:198,211)."""
    import socket
    import subprocess
    if len(devices) < 1:
        print("Error -- --devices needs at least one GPU index", file=sys.stderr)
        return 1
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out, skip = [], False
    for a in argv:                              # the children get the command line without --devices
        if skip:
            skip = False
        elif a == "--devices":
            skip = True
        elif not a.startswith("--devices="):
            out.append(a)
    procs = []
    for r in range(len(devices)):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(len(devices)), LOCAL_RANK=str(r), NB_DEVICES=",".join(map(str, devices)),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if len(set(devices)) < len(devices):    # ranks sharing a GPU (rehearsal on a one-GPU box): RCCL needs one device per rank
            env.setdefault("NB_DIST_BACKEND", "gloo")
        procs.append(subprocess.Popen([sys.executable, "-m", "nimble_amd"] + out, env=env))
    return max(p.wait() for p in procs)


def main(argv=None):
    parser = build_parser()
    args, extra = parser.parse_known_args(argv)
    if args.subcommand in ("download", "plot"):
        print(f"Error -- `{args.subcommand}` is not part of this build: the aligner is the in-tree libnimble_amd.so (nothing to "
              "download) and plots are out of scope.", file=sys.stderr)
        return 2
    if extra:
        parser.error("unrecognized arguments: " + " ".join(extra))
    if args.subcommand == "generate":
        generate(args.file, args.opt_file, args.output_path)
        return 0
    if args.subcommand == "align" and args.devices and int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return _spawn_ranks(list(sys.argv[1:] if argv is None else argv), [int(x) for x in args.devices.split(",") if x != ""])
    if args.subcommand == "align":
        tenx = TenxOptions(args.map, args.cb_length, args.umi_length) if args.tenx else None
        return align(args.reference, args.output, args.input, args.num_cores, args.strand_filter, args.trim, args.tmpdir,
                     tenx=tenx, device_id=args.device, report_output=args.report_output, report_threshold=args.report_threshold,
                     report_disable_thresholding=args.report_disable_thresholding)
    if args.subcommand == "fastq-to-bam":
        from .fastq_to_bam import fastq_to_bam_with_barcodes
        fastq_to_bam_with_barcodes(args.r1_fastq, args.r2_fastq, args.map, args.output, args.num_cores, args.cb_length, args.umi_length)
        return 0
    if args.subcommand == "report":
        cols = args.summarize.split(",") if args.summarize else None
        report(args.input, args.output, cols, args.threshold, args.disable_thresholding)
        return 0
    parser.print_help()
    return 0


if __name__ == "__main__":
    sys.exit(main())

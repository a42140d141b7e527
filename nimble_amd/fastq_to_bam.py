"""`nimble fastq-to-bam`: 10x paired FASTQ -> unaligned BAM with CB/UB tags (host only, no GPU needed).

Same signature, filters and printed statistics as fastq_to_bam_with_barcodes
(/root/reference/nimble/fastq_barcode_processor.py:213-320); the work is done by nb_fastq_to_bam (csrc/nb_ingest.cpp): the
FASTQ files are read with the engine's own readers (read-ahead, own inflater), barcodes corrected with nb_whitelist_correct
(same rule as the Python twin tests/barcodes_twin.py), records written as BGZF blocks compressed on `num_cores` threads. Differences,
both deliberate: pairs are written in file order (the reference writes them in the completion order of its thread pool),
and no pysam / Biopython is needed. `align` reads the result back like any BAM — or skips it: `align --tenx` takes the two
FASTQ files directly."""
import ctypes as ct
import sys

from . import _lib


def fastq_to_bam_with_barcodes(r1_fastq, r2_fastq, cb_whitelist_file, output_bam, num_cores=1, cb_length=16, umi_length=12):
    L = _lib.lib()
    print("Loading cell barcode whitelist...")
    wl = ct.c_void_p()
    stats = (ct.c_int64 * 8)()
    try:
        _lib.check(L.nb_whitelist_load(str(cb_whitelist_file).encode(), ct.byref(wl)))
        _lib.check(L.nb_whitelist_set_threads(wl, int(num_cores)))
        print(f"Loaded {L.nb_whitelist_size(wl)} cell barcodes from whitelist")
        print(f"Processing paired FASTQ files with {num_cores} threads...")
        _lib.check(L.nb_fastq_to_bam(str(r1_fastq).encode(), str(r2_fastq).encode(), wl, str(output_bam).encode(), int(num_cores),
                                     int(cb_length), int(umi_length), stats))
    except Exception as e:   # the reference's error convention (fastq_barcode_processor.py:285-287)
        print(f"Error during processing: {e}", file=sys.stderr)
        sys.exit(1)
    finally:
        if wl:
            L.nb_whitelist_destroy(wl)
    total, written, perfect, corrected, none, mismatch, too_short, no_rest = (int(x) for x in stats)
    print("\n=== Processing Statistics ===")
    print(f"Total read pairs: {total}")
    print(f"Written pairs: {written}")
    print("\nCell Barcode Correction:")
    print(f"  Perfect matches: {perfect}")
    print(f"  Corrected (1-edit): {corrected}")
    print(f"  No valid correction: {none}")
    processed = perfect + corrected + none
    if processed > 0:
        print(f"  Correction rate: {100.0 * perfect / processed:.2f}% perfect, {100.0 * corrected / processed:.2f}% corrected, "
              f"{100.0 * none / processed:.2f}% dropped")
    print("\nOther filters:")
    print(f"  Name mismatch: {mismatch}")
    print(f"  Too short: {too_short}")
    print(f"  No remaining sequence: {no_rest}")
    print(f"\nOutput BAM written to: {output_bam}")
    return {"total_pairs": total, "written_pairs": written, "cb_perfect_match": perfect, "cb_corrected": corrected, "cb_no_correction": none,
            "name_mismatch": mismatch, "too_short": too_short, "no_remaining_seq": no_rest}

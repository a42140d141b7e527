"""Small host helpers that the reference keeps in nimble/utils.py."""
import os


def append_path_string(input_path, path_append_string):
    """append_path_string (/root/reference/nimble/utils.py:9-27): insert a suffix before the (multi-)extension,
    e.g. ("/a/b/out.tsv.gz", ".hla") -> "/a/b/out.hla.tsv.gz". Pinned by tests/golden/append_path_string.json."""
    filename = os.path.basename(input_path)
    root, ext = filename, ""
    while True:
        root, ext2 = os.path.splitext(root)
        if ext2 == "":
            break
        ext = ext2 + ext
    return os.path.join(os.path.dirname(input_path), root + path_append_string + ext)


def parse_trim(trim, n_libraries):
    """--trim "<TARGET_LENGTH>:<STRICTNESS>,..." one entry per library (/root/reference/nimble/__main__.py:400).
    Validated and returned; reads are not trimmed: the rule lives in the external aligner and is specified nowhere in the
    reference tree (docs/SPEC.md §5, open item)."""
    if trim == "" or trim is None:
        return None
    parts = trim.split(",")
    if len(parts) != n_libraries:
        raise ValueError(f"--trim has {len(parts)} entries for {n_libraries} libraries")
    out = []
    for p in parts:
        a, _, b = p.partition(":")
        out.append((int(a), float(b)))
    return out

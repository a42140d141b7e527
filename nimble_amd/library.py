"""Reference-library JSON: the `[config, data]` file `nimble generate` writes and the aligner reads.

Mirrors /root/reference/nimble/types.py (Config :10-25, Data :29-32), the writer in
/root/reference/nimble/__main__.py:45-110 (generate, process_file, collate_data) and the parsers in
/root/reference/nimble/parse.py:15-35 (parse_fasta) and :78-139 (parse_csv). The JSON text produced for a CSV library is
byte-identical to the reference's (tests/golden/generate_csv.json). FASTA is read without Biopython (absent here):
record id = first whitespace-delimited token after '>', sequence = concatenated lines, as Bio.SeqIO yields them.
`genbank://` cells (parse.py:120-130, network) are rejected.
"""
import csv
import json
import os
import pathlib
import sys

CONFIG_DEFAULTS = [  # key order of Config.__init__ (types.py:12-25) — json.dump preserves it
    ("score_threshold", 20), ("score_filter", 25), ("score_percent", 0.5), ("num_mismatches", 0),
    ("discard_multiple_matches", False), ("intersect_level", 0), ("group_on", ""), ("discard_multi_hits", 0),
    ("require_valid_pair", False), ("data_type", "RNA"), ("filters", []), ("max_hits_to_report", 10),
    ("trim_target_length", 50), ("trim_strictness", 0.9),
]
BASE_HEADERS = ["reference_genome", "sequence_name", "nt_length", "sequence"]  # types.py:31


def default_config():
    return {k: (list(v) if isinstance(v, list) else v) for k, v in CONFIG_DEFAULTS}


class Data:
    def __init__(self):
        self.headers = list(BASE_HEADERS)
        self.columns = [[], [], [], []]


def library_name_from_filename(path):
    """get_library_name_from_filename (utils.py:31-32)."""
    return pathlib.Path(path).stem.replace("_", " ")


def read_fasta(path):
    """[(id, sequence)] in file order."""
    recs, name, chunks = [], None, []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    recs.append((name, "".join(chunks)))
                tok = line[1:].split()
                name, chunks = (tok[0] if tok else ""), []
            elif name is not None:
                chunks.append(line.strip())
    if name is not None:
        recs.append((name, "".join(chunks)))
    return recs


def parse_fasta(path):
    """parse_fasta (parse.py:15-35)."""
    data, ref = Data(), library_name_from_filename(path)
    for rid, seq in read_fasta(path):
        data.columns[0].append(ref)
        data.columns[1].append(rid if rid else "null")
        data.columns[2].append(str(len(seq)))
        data.columns[3].append(seq)
    return data, default_config()


def parse_csv(path, has_sequences=True):
    """parse_csv (parse.py:78-139)."""
    data, ref = Data(), library_name_from_filename(path)
    genomes, names, lengths, seqs, meta = [], [], [], [], []
    with open(path) as f:
        reader = csv.reader(f, delimiter=",", quotechar='"')
        headers = next(reader)
        seq_idx = headers.index("sequence") if has_sequences else None
        name_idx = headers.index("name")
        headers.pop(name_idx)
        if has_sequences and name_idx < seq_idx:
            seq_idx -= 1
        if has_sequences:
            headers.pop(seq_idx)
        for row in reader:
            names.append(row.pop(name_idx))
            genomes.append(ref)
            if has_sequences:
                raw = row.pop(seq_idx)
                if "genbank://" in raw:
                    raise ValueError("genbank:// sequences need network access (reference: nimble/remote.py); not supported")
                seqs.append(raw)
                lengths.append(str(len(raw)))
            if not meta:
                meta = [[] for _ in headers]
            for i, col in enumerate(row):
                meta[i].append(col)
    data.headers.extend(headers)
    data.columns = [genomes, names, lengths, seqs] + meta
    return data, default_config()


def _process_file(file, paired):
    """process_file (__main__.py:70-85). Accepts .fa/.fna as well as the reference's .fasta."""
    if not file:
        return None, None, False
    suffix = pathlib.Path(file).suffix
    if suffix in (".fasta", ".fa", ".fna"):
        return (*parse_fasta(file), False)
    if suffix == ".csv":
        return (*parse_csv(file, not paired), True)
    return None, None, False


def _collate(data, metadata):
    """collate_data (__main__.py:88-110): FASTA supplies sequence + nt_length into the CSV's metadata rows."""
    ni, si, li = (data.headers.index(h) for h in ("sequence_name", "sequence", "nt_length"))
    mni, msi, mli = (metadata.headers.index(h) for h in ("sequence_name", "sequence", "nt_length"))
    n = len(data.columns[si])
    metadata.columns[msi] = ["" for _ in range(n)]
    metadata.columns[mli] = ["" for _ in range(n)]
    for frm, name in enumerate(data.columns[ni]):
        if name not in metadata.columns[mni]:
            print("Error -- record " + name + " is not found in both input files.")
            sys.exit()
        to = metadata.columns[mni].index(name)
        metadata.columns[msi][to] = data.columns[si][frm]
        metadata.columns[mli][to] = data.columns[li][frm]
    return metadata


def generate(file, opt_file, output_path):
    """generate (__main__.py:45-65): FASTA / CSV / CSV+FASTA -> lib.json, same text as the reference writes."""
    data, config, is_csv_req = _process_file(file, opt_file)
    data_opt, config_opt, is_csv_opt = _process_file(opt_file, file)
    final_config = config_opt if (data_opt is not None and is_csv_opt) else config
    if data_opt is not None:
        final = _collate(data_opt, data) if is_csv_req else (_collate(data, data_opt) if is_csv_opt else None)
    else:
        final = data
    if final is None or final_config is None:
        raise ValueError("generate: expected a .fasta and/or .csv input")
    with open(output_path, "w") as f:
        json.dump([final_config, {"headers": final.headers, "columns": final.columns}], f, indent=2)


class Library:
    """A loaded lib.json: config (with defaults filled in), sequences and the feature name of every sequence."""

    def __init__(self, path):
        with open(path) as f:
            doc = json.load(f)
        if not (isinstance(doc, list) and len(doc) == 2):
            raise ValueError(f"{path}: expected a [config, data] JSON array")
        cfg, data = doc
        self.path = path
        self.stem = os.path.splitext(os.path.basename(path))[0]
        self.config = default_config()
        self.config.update(cfg)
        headers, columns = data["headers"], data["columns"]
        if headers[:4] != BASE_HEADERS:
            raise ValueError(f"{path}: first four headers must be {BASE_HEADERS}")
        self.headers, self.columns = headers, columns
        self.names = list(columns[1])
        self.sequences = list(columns[3])
        if not self.sequences:
            raise ValueError(f"{path}: library has no sequences")
        group_on = self.config.get("group_on", "") or ""
        if group_on == "":
            self.features = list(self.names)
        else:
            if group_on not in headers:
                raise ValueError(f"{path}: group_on column {group_on!r} is not a header of the library")
            self.features = list(columns[headers.index(group_on)])
        if len(self.features) != len(self.sequences):
            raise ValueError(f"{path}: column lengths differ")
        self.k = int(cfg.get("k", 20)) if isinstance(cfg, dict) else 20   # optional extension key; docs/SPEC.md §1

    def feature_list(self, members):
        """docs/SPEC.md §5: distinct feature names of the set's members, in ascending sequence-index order."""
        seen, out = set(), []
        for m in members:
            f = self.features[m]
            if f not in seen:
                seen.add(f)
                out.append(f)
        return out

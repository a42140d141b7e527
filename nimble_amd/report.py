"""`nimble report`: per-read TSV -> per-cell / per-feature UMI counts. pandas-free host code, bit-compatible.

Mirrors /root/reference/nimble/__main__.py:213-302 (report, check_df_from_input, convert_df_to_proper_umi,
write_empty_df) and /root/reference/nimble/utils.py:119-224 (per_umi_thresholding, umi_intersection, intersect_lists),
including the float64 arithmetic the reference inherits from pandas: `groupby(...).sum()` on floats is
Kahan-compensated in row order (verified in SURVEY.md §7), the running `total_score` is a plain left-to-right sum, the
ratio test is `share / total < threshold` in IEEE double, and — a quirk kept on purpose — each thresholding round
re-derives the shares from the ORIGINAL rows minus only that round's `to_drop` (utils.py:146-173), not the union of
all rounds. Pinned by tests/golden/report_cases.json (outputs of the reference's own report()).

Deviation (documented, tested): when every row ends up filtered the reference raises KeyError at __main__.py:283; this
implementation writes the same empty output the reference writes for empty input (write_empty_df, __main__.py:299-302).
"""
import csv
import gzip
import os
import re

import numpy as np

# strings pandas.read_csv treats as missing by default
_NA = {"", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>", "N/A", "NA",
       "NULL", "NaN", "None", "n/a", "nan", "null"}

REQUIRED = ("nimble_features", "nimble_score", "r1_CB", "r1_UB")


class _Kahan:
    """pandas' group_sum accumulator (pandas/_libs/groupby.pyx): compensated sum in arrival order."""
    __slots__ = ("s", "c")

    def __init__(self):
        self.s = np.float64(0.0)
        self.c = np.float64(0.0)

    def add(self, v):
        y = v - self.c
        t = self.s + y
        c = (t - self.s) - y
        self.c = c if c == c and abs(c) != np.inf else np.float64(0.0)
        self.s = t


def write_empty(output):
    """write_empty_df (__main__.py:299-302): a zero-byte file."""
    print("No data to parse from input file, writing empty output.")
    open(output, "w").close()


def _parse_score(tok):
    try:
        return int(tok)
    except ValueError:
        return float(tok)


def read_rows(path):
    """check_df_from_input + the column selection / null filtering of convert_df_to_proper_umi.
    Returns None for "no data", else a list of (cb, umi, features_sorted, score)."""
    if os.path.getsize(path) == 0:
        return None
    if path.endswith(".gz"):      # pd.read_csv infers gzip from the extension (reference __main__.py:219)
        import gzip
        with gzip.open(path, "rt", newline="") as f:
            lines = f.read().split("\n")
    else:
        with open(path, newline="") as f:
            lines = f.read().split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    lines = [ln[:-1] if ln.endswith("\r") else ln for ln in lines]
    if not lines:
        return None
    header = lines[0].split("\t")
    missing = [c for c in REQUIRED if c not in header]
    body = [ln for ln in lines[1:] if ln != ""]   # pandas skips blank lines
    if not body:
        return None
    if missing:
        raise KeyError(f"input TSV lacks column(s) {missing}")
    ix = [header.index(c) for c in REQUIRED]
    rows, any_float = [], False
    for ln in body:
        fld = ln.split("\t")
        vals = [fld[i] if i < len(fld) else "" for i in ix]
        if any(v in _NA for v in vals):
            continue  # dropna + != '' (__main__.py:244-245)
        feats, score, cb, umi = vals
        sc = _parse_score(score)
        any_float |= isinstance(sc, float)
        rows.append((cb, umi, ",".join(sorted(feats.split(","))), sc))
    if any_float:
        rows = [(cb, umi, ft, float(sc)) for cb, umi, ft, sc in rows]
    return rows


def merge_rows(rows):
    """groupby(['cb','umi','features'])['nimble_score'].sum() (__main__.py:251): keys sorted, sums in file order."""
    acc, is_float = {}, bool(rows) and isinstance(rows[0][3], float)
    for cb, umi, ft, sc in rows:
        key = (cb, umi, ft)
        if is_float:
            acc.setdefault(key, _Kahan()).add(np.float64(sc))
        else:
            acc[key] = acc.get(key, 0) + sc
    out = []
    for key in sorted(acc):
        v = acc[key]
        out.append((key[0], key[1], key[2], float(v.s) if is_float else v))
    return out


def _shares(group, drop):
    """One pass of utils.py:125-133 / :158-173 over the UMI group's rows, skipping features in `drop`."""
    sums, total, any_row = {}, 0, False
    for _, _, ft, sc in group:
        feats = [f for f in ft.split(",") if f not in drop]
        if not feats:
            continue
        any_row = True
        spf = np.float64(sc) / np.float64(len(feats))
        total = total + sc
        for f in feats:
            sums.setdefault(f, _Kahan()).add(spf)
    return sums, total, any_row


def filter_umi_features(group, threshold):
    """filter_umi_features (utils.py:120-191) for one (cb, umi) group: returns the surviving feature set."""
    sums, total, _ = _shares(group, ())
    while True:
        if not sums:
            return set()
        with np.errstate(divide="ignore", invalid="ignore"):
            tot = np.float64(total)
            to_drop = {f for f, k in sums.items() if bool(k.s / tot < threshold)}
        if not to_drop:
            return set(sums)
        sums, total, any_row = _shares(group, to_drop)
        if not any_row:
            return set()


def report_native(input, output, threshold=0.05, disable_thresholding=False):
    """The same reduction in C++ (csrc/nb_report.cpp, nb_report_tsv): for 10^7..10^8-row tables."""
    import ctypes as ct
    from . import _lib
    st = (ct.c_int64 * 2)()
    _lib.check(_lib.lib().nb_report_tsv(os.fsencode(input), os.fsencode(output), float(threshold), int(bool(disable_thresholding)), st))
    if st[0]:
        print("No data to parse from input file, writing empty output.")
        return False
    print(f"Dropped {st[1]} UMIs due to empty intersections")
    return True


def report_rows(cb_ids, umi_keys, feat_ids, scores, cb_strings, feat_strings, output, threshold=0.05, disable_thresholding=False):
    """report() from arrays instead of a per-read TSV (nb_report_rows): row i = (cb_strings[cb_ids[i]], umi_keys[i],
    feat_strings[feat_ids[i]], scores[i]). Same output text as report() on the equivalent TSV."""
    import ctypes as ct
    from . import _lib
    cb_ids = np.ascontiguousarray(cb_ids, dtype=np.uint32); umi_keys = np.ascontiguousarray(umi_keys, dtype=np.uint64)
    feat_ids = np.ascontiguousarray(feat_ids, dtype=np.uint32); scores = np.ascontiguousarray(scores, dtype=np.int64)
    n = cb_ids.shape[0]
    cbs = (ct.c_char_p * max(1, len(cb_strings)))(*[s.encode() if isinstance(s, str) else s for s in cb_strings])
    fts = (ct.c_char_p * max(1, len(feat_strings)))(*[s.encode() if isinstance(s, str) else s for s in feat_strings])
    st = (ct.c_int64 * 2)()
    _lib.check(_lib.lib().nb_report_rows(n, cb_ids.ctypes.data, umi_keys.ctypes.data, feat_ids.ctypes.data, scores.ctypes.data, len(cb_strings), cbs,
                                         len(feat_strings), fts, float(threshold), int(bool(disable_thresholding)), os.fsencode(output), st))
    if st[0]:
        print("No data to parse from input file, writing empty output.")
    else:
        print(f"Dropped {st[1]} UMIs due to empty intersections")


_INT_RE = re.compile(r"[+-]?\d+\Z")
_TRUE, _FALSE = {"True", "TRUE", "true"}, {"False", "FALSE", "false"}


def _column_kind(values):
    """The dtype pandas' parser gives a column (None = missing cell): 'int', 'float', 'bool' or 'object'."""
    present = [v for v in values if v is not None]
    if not present:
        return "float"
    if all(_INT_RE.match(v) for v in present):
        return "int" if len(present) == len(values) else "float"      # a missing cell turns an integer column into floats
    try:
        for v in present:
            float(v)
        return "float"
    except ValueError:
        pass
    if len(present) == len(values) and all(v in _TRUE or v in _FALSE for v in present):
        return "bool"
    return "object"


def _column_key(kind, v):
    """(hashable key, text as pandas prints it inside an f-string) of one cell."""
    if kind == "int":
        return int(v), str(int(v))
    if kind == "float":
        x = float(v)
        return x, repr(x)
    if kind == "bool":
        return v in _TRUE, str(v in _TRUE)
    return v, v


def summarize_fields(input, columns, output_file):
    """`report -s col1,col2`: per UMI, the values of the named columns of the per-read table with their frequencies,
    "value(count); value(count)" by descending count (summarize_fields, /root/reference/nimble/__main__.py:307-310, applied to
    the table with r1_CB / r1_UB / nimble_features renamed to cb / umi / features, :236). Restated without pandas: column
    types as pandas' reader infers them (integers print bare, an integer column with a missing cell prints as floats), values
    in order of first appearance, then the same descending sort pandas' value_counts applies; UMI groups in sorted order, rows
    without a UMI left out. Held to outputs of the reference (tests/golden/summarize_cases.json)."""
    opener = gzip.open if str(input).endswith(".gz") else open
    with opener(input, "rt", newline="") as f:
        lines = f.read().split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    rename = {"r1_CB": "cb", "r1_UB": "umi", "nimble_features": "features"}
    header = [rename.get(h, h) for h in lines[0].rstrip("\r").split("\t")]
    for c in ["umi"] + list(columns):
        if c not in header:
            raise KeyError(c)
    want = [header.index(c) for c in ["umi"] + list(columns)]
    table = []
    for line in lines[1:]:
        line = line.rstrip("\r")
        if line == "":
            continue                                      # pandas skips blank lines
        cells = line.split("\t")
        cells += [""] * (len(header) - len(cells))
        table.append([None if cells[j] in _NA else cells[j] for j in want])
    kinds = [_column_kind([row[j] for row in table]) for j in range(len(want))]
    groups = {}
    for row in table:
        if row[0] is None:
            continue
        key, text = _column_key(kinds[0], row[0])
        groups.setdefault(key, (text, []))[1].append(row)
    with open(output_file, "w", newline="") as f:
        w = csv.writer(f, delimiter="\t", quoting=csv.QUOTE_MINIMAL, lineterminator="\n")
        w.writerow(["umi"] + list(columns))
        for key in sorted(groups):
            text, rows = groups[key]
            out = [text]
            for j in range(1, len(want)):
                counts, texts = {}, {}
                for row in rows:
                    if row[j] is None:
                        continue
                    k, t = _column_key(kinds[j], row[j])
                    counts[k] = counts.get(k, 0) + 1
                    texts.setdefault(k, t)
                keys = list(counts)
                c = np.array([counts[k] for k in keys], dtype=np.int64)
                # Series.sort_values(ascending=False) as pandas does it: argsort of the reversed array, reversed back
                order = np.arange(len(keys))[::-1][c[::-1].argsort(kind="quicksort")][::-1] if keys else []
                out.append("; ".join(f"{texts[keys[i]]}({counts[keys[i]]})" for i in order))
            w.writerow(out)


def report(input, output, summarize_columns_list=None, threshold=0.05, disable_thresholding=False, engine=None):
    """Same signature and output as the reference's report() (__main__.py:254). `engine`: "python" (this module) or
    "native" (nb_report_tsv); default from $NIMBLE_REPORT_ENGINE, else native. Both are pinned by the same fixtures."""
    engine = engine or os.environ.get("NIMBLE_REPORT_ENGINE", "native")
    if engine == "native":
        if report_native(input, output, threshold, disable_thresholding) and summarize_columns_list:
            summarize_fields(input, summarize_columns_list, "summarize." + output)
        return
    rows = read_rows(input)
    if rows is None:
        write_empty(output)
        return
    rows = merge_rows(rows)
    if not rows:
        write_empty(output)
        return
    # group by (cb, umi); rows are already sorted by (cb, umi, features)
    groups, cur = [], None
    for r in rows:
        if cur is None or (r[0], r[1]) != cur[0]:
            cur = ((r[0], r[1]), [])
            groups.append(cur)
        cur[1].append(r)
    per_umi = []   # ((cb, umi), intersected sorted feature list)
    for key, grp in groups:
        kept = []
        if disable_thresholding:
            kept = [set(ft.split(",")) for _, _, ft, _ in grp]
        else:
            survivors = filter_umi_features(grp, threshold)
            for _, _, ft, _ in grp:
                s = set(ft.split(",")) & survivors
                if s:
                    kept.append(s)     # rows whose filtered_features is '' are removed (utils.py:205)
        if kept:
            per_umi.append((key, sorted(set.intersection(*kept))))   # intersect_lists (utils.py:221-224)
    if not per_umi:
        write_empty(output)
        return
    empty = sum(1 for _, fl in per_umi if not fl)
    print(f"Dropped {empty} UMIs due to empty intersections")
    counts = {}
    for (cb, _), fl in per_umi:
        if fl:
            k = (cb, ",".join(fl))
            counts[k] = counts.get(k, 0) + 1
    with open(output, "w", newline="") as f:
        w = csv.writer(f, delimiter="\t", quoting=csv.QUOTE_MINIMAL, lineterminator="\n")
        for cb, feat in sorted(counts):
            w.writerow([feat, counts[(cb, feat)], cb])
    if summarize_columns_list:
        summarize_fields(input, summarize_columns_list, "summarize." + output)   # (the reference's naming, __main__.py:300)

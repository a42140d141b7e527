#!/usr/bin/env python
"""Generate tests/golden/*.json by RUNNING the reference's own Python (in the build container only).

Usage (never on the GPU box — /root/reference does not exist there):
    mkdir -p /tmp/stubs/Bio && touch /tmp/stubs/{distro,pysam,seaborn}.py /tmp/stubs/Bio/{__init__,SeqIO,Entrez}.py
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/stubs:/root/reference python /root/repo/tests/golden/make_golden.py

The empty stub modules only satisfy imports the exercised functions never touch (`distro` is imported and unused at
nimble/utils.py:3; pysam/seaborn/Bio are used only by sort_input_bam, plots and FASTA parsing). The fixtures are data:
inputs and the outputs the reference produced for them (report(): nimble/__main__.py:254-302; generate() on CSV:
__main__.py:45-65 + parse.py:78-139; append_path_string: utils.py:9-27). No reference source is copied.
"""
import json
import os
import random
import sys
import tempfile
import zlib

OUT = os.path.dirname(os.path.abspath(__file__))


def run_report(m, text, threshold, disable):
    with tempfile.TemporaryDirectory() as d:
        i, o = os.path.join(d, "in.tsv"), os.path.join(d, "out.tsv")
        with open(i, "w") as f:
            f.write(text)
        try:
            m.report(i, o, None, threshold, disable)
        except Exception as e:  # the reference itself raises on some inputs (e.g. every row filtered): record that
            return {"raised": type(e).__name__}
        return open(o).read() if os.path.exists(o) else None


def tsv(rows, header=("nimble_features", "nimble_score", "r1_CB", "r1_UB"), extra=None):
    cols = list(header) + (list(extra) if extra else [])
    lines = ["\t".join(cols)]
    for r in rows:
        lines.append("\t".join(str(x) for x in r))
    return "\n".join(lines) + "\n"


def handmade():
    c = []
    c.append(("pinned_survey", tsv([("A,B", 10, "cell1", "UMI1"), ("C,A", 20, "cell1", "UMI1"), ("B,D", 15, "cell1", "UMI2"), ("G", 50, "cell3", "UMI3")])))
    c.append(("basic_dominant", tsv([("A", 90, "c1", "u1"), ("B", 5, "c1", "u1"), ("A,B", 5, "c1", "u1")])))
    c.append(("all_minor", tsv([(f"F{i}", 1, "c1", "u1") for i in range(30)])))
    c.append(("tie_at_threshold", tsv([("A", 95, "c1", "u1"), ("B", 5, "c1", "u1")])))
    c.append(("zero_scores", tsv([("A", 0, "c1", "u1"), ("B", 0, "c1", "u1"), ("A,B", 0, "c1", "u2")])))
    c.append(("dup_feature_in_row", tsv([("A,A,B", 30, "c1", "u1"), ("B", 1, "c1", "u1")])))
    c.append(("huge_scores", tsv([("A", 10 ** 12, "c1", "u1"), ("B", 10 ** 10, "c1", "u1"), ("A,B", 3 * 10 ** 11, "c1", "u1")])))
    c.append(("float_scores", tsv([("A", 0.7, "c1", "u1"), ("B", 0.1, "c1", "u1"), ("A,B", 0.2, "c1", "u1"), ("C", 0.04, "c1", "u1")])))
    c.append(("unsorted_features_merge", tsv([("B,A", 3, "c1", "u1"), ("A,B", 4, "c1", "u1"), ("C,B,A", 2, "c1", "u1")])))
    c.append(("disjoint_sets_same_umi", tsv([("A", 10, "c1", "u1"), ("B", 10, "c1", "u1")])))
    c.append(("multi_cell", tsv([("A", 5, "c2", "u1"), ("A", 5, "c1", "u1"), ("A,B", 5, "c1", "u2"), ("B", 9, "c1", "u3"), ("B,A", 1, "c2", "u9")])))
    c.append(("cascade_drop", tsv([("A", 100, "c1", "u1"), ("B,C", 12, "c1", "u1"), ("C", 5, "c1", "u1"), ("D,B", 4, "c1", "u1")])))
    c.append(("empty_fields", "nimble_features\tnimble_score\tr1_CB\tr1_UB\nA\t5\tc1\tu1\n\t5\tc1\tu1\nB\t5\t\tu1\nB\t5\tc1\t\nC\t\tc1\tu2\nC\t7\tc1\tu2\n"))
    c.append(("extra_columns", tsv([("A", 5, "c1", "u1", 11, "x"), ("A,B", 6, "c1", "u1", 12, "y")], extra=("r1_POS", "r1_GN"))))
    c.append(("header_only", "nimble_features\tnimble_score\tr1_CB\tr1_UB\n"))
    c.append(("zero_bytes", ""))
    c.append(("all_rows_null", "nimble_features\tnimble_score\tr1_CB\tr1_UB\n\t5\tc1\tu1\n"))
    return c


def random_tables(n_cases, seed):
    rnd = random.Random(seed)
    feats = [f"HLA-{x}*{i:02d}" for x in "ABC" for i in range(1, 5)] + ["KIR2DL1", "KIR3DL2", "MICA"]
    cases = []
    for k in range(n_cases):
        rows = []
        ncb, numi = rnd.randint(1, 4), rnd.randint(1, 6)
        for _ in range(rnd.randint(1, 40)):
            fs = rnd.sample(feats, rnd.randint(1, 4))
            if rnd.random() < 0.1:
                fs.append(fs[0])  # duplicate inside a row
            rnd.shuffle(fs)
            score = rnd.choice([1, 1, 1, 2, 2, 3, 5, 8, 20, 91]) if k % 5 else round(rnd.random() * 10, 3)
            rows.append((",".join(fs), score, f"CB{rnd.randrange(ncb)}", f"UMI{rnd.randrange(numi)}"))
        cases.append((f"random_{seed}_{k}", tsv(rows)))
    return cases


def summarize_cases():
    rnd = random.Random(7)
    hdr = ("nimble_features", "nimble_score", "r1_CB", "r1_UB")
    c = []
    c.append(("strings", ["r1_GN"], tsv([("A", 1, "c1", "u1", "HLA-A"), ("A", 1, "c1", "u1", "HLA-B"), ("B", 1, "c2", "u1", "HLA-A"), ("B", 1, "c1", "u2", "HLA-C")], hdr, ("r1_GN",))))
    c.append(("ints", ["r1_POS"], tsv([("A", 1, "c1", "u1", 100), ("A", 1, "c1", "u1", 100), ("A", 1, "c1", "u1", 7), ("B", 1, "c1", "u2", 7)], hdr, ("r1_POS",))))
    c.append(("ints_with_missing", ["r1_POS"], tsv([("A", 1, "c1", "u1", 100), ("A", 1, "c1", "u1", ""), ("A", 1, "c1", "u2", ""), ("B", 1, "c1", "u3", 7)], hdr, ("r1_POS",))))
    c.append(("floats", ["x"], tsv([("A", 1, "c1", "u1", 0.5), ("A", 1, "c1", "u1", "0.50"), ("A", 1, "c1", "u1", "1e-05"), ("B", 1, "c1", "u2", 3)], hdr, ("x",))))
    c.append(("mixed_column", ["x"], tsv([("A", 1, "c1", "u1", 5), ("A", 1, "c1", "u1", "5"), ("A", 1, "c1", "u1", "five"), ("B", 1, "c1", "u2", "05")], hdr, ("x",))))
    c.append(("ties_first_seen", ["g"], tsv([("A", 1, "c1", "u1", v) for v in ("z", "y", "x", "y", "z", "w", "x")], hdr, ("g",))))
    c.append(("umi_missing_and_numeric", ["g"], tsv([("A", 1, "c1", 30, "a"), ("A", 1, "c1", 4, "b"), ("A", 1, "c1", "", "c"), ("A", 1, "c1", 4, "b")], hdr, ("g",))))
    c.append(("bools", ["b"], tsv([("A", 1, "c1", "u1", "True"), ("A", 1, "c1", "u1", "False"), ("A", 1, "c1", "u1", "True"), ("A", 1, "c1", "u2", "False")], hdr, ("b",))))
    c.append(("group_all_missing", ["g", "h"], tsv([("A", 1, "c1", "u1", "", "q"), ("A", 1, "c1", "u1", "", "q"), ("A", 1, "c1", "u2", "k", "")], hdr, ("g", "h"))))
    c.append(("two_columns_and_core", ["r1_GN", "cb", "features"], tsv([("B,A", 1, "c1", "u1", "G1"), ("A", 2, "c2", "u1", "G1"), ("A", 2, "c2", "u1", "G2"), ("C", 2, "c2", "u2", "G2")], hdr, ("r1_GN",))))
    c.append(("quoting", ["g"], tsv([("A", 1, "c1", "u1", 'say "hi"'), ("A", 1, "c1", "u1", "x;y")], hdr, ("g",))))
    for k in range(12):
        rows = []
        numi = rnd.randint(1, 5)
        nvals = rnd.choice([3, 6, 30])
        for _ in range(rnd.randint(5, 120)):
            rows.append(("A", 1, f"c{rnd.randrange(3)}", f"UMI{rnd.randrange(numi)}", f"G{rnd.randrange(nvals)}", rnd.choice([1, 2, 2, 3, 17, ""]) if k % 2 else rnd.randrange(4)))
        c.append((f"random_{k}", ["r1_GN", "r1_POS"], tsv(rows, hdr, ("r1_GN", "r1_POS"))))
    return c


def run_summarize(m, text, columns):
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "in.tsv"), "w").write(text)
        cwd = os.getcwd()
        os.chdir(d)                     # the reference writes to "summarize." + output: only a bare file name works
        try:
            m.report("in.tsv", "out.tsv", columns, 0.05, False)
            return open("summarize.out.tsv").read()
        except Exception as e:
            return {"raised": type(e).__name__}
        finally:
            os.chdir(cwd)


def reference_unit_vectors():
    """The reference's OWN 24 reducer unit tests (test/test.py; CI runs only 11 of them because the second
    `TestDataProcessing` class shadows the first) executed here with recording wrappers around
    per_umi_thresholding / umi_intersection / intersect_lists: every call's input and returned value is dumped as data.
    A recorded value is the test's expected vector whenever the test PASSES (its assertions compare exactly that value
    with the hand-written expectation), and all 24 must pass for this function to return. Nothing but frames/lists of
    strings and numbers is kept: no reference source text."""
    import unittest
    import pandas as pd
    import nimble.utils as u
    src = open("/root/reference/test/test.py").read()
    first = src.index("class TestDataProcessing(")
    second = src.index("class TestDataProcessing(", first + 1)
    src = src[:second] + src[second:].replace("class TestDataProcessing(", "class TestDataProcessingSecond(", 1)   # un-shadow, in memory
    ns = {"__name__": "reference_tests"}
    exec(compile(src, "reference_tests", "exec"), ns)
    calls, current = {}, [None]

    def frame(df):
        df = df.reset_index(drop=True)
        return {"columns": [str(c) for c in df.columns],
                "rows": [[(v if isinstance(v, (str, list)) else (None if pd.isna(v) else (int(v) if float(v).is_integer() and not isinstance(v, float) else float(v))))
                          for v in row] for row in df.itertuples(index=False, name=None)]}

    def wrap(name, fn):
        def inner(*args, **kw):
            if kw:                                   # (some tests pass threshold= by keyword)
                args = args + tuple(kw[k] for k in ("threshold",) if k in kw)
            rec = {"fn": name}
            if name == "intersect_lists":
                rec["input"] = [list(x) for x in args[0]]
            else:
                rec["input"] = frame(args[0].copy())
                if name == "per_umi_thresholding":
                    rec["threshold"] = float(args[1])
            out = fn(*args)
            rec["output"] = list(out) if name == "intersect_lists" else frame(out.copy())
            calls.setdefault(current[0], []).append(rec)
            return out
        return inner

    real_il = u.intersect_lists
    ns["intersect_lists"] = wrap("intersect_lists", real_il)
    ns["per_umi_thresholding"] = wrap("per_umi_thresholding", u.per_umi_thresholding)
    ns["umi_intersection"] = wrap("umi_intersection", u.umi_intersection)      # (its inner intersect_lists calls are not recorded)
    out = []
    for cls in (ns["TestDataProcessing"], ns["TestDataProcessingSecond"]):
        for meth in sorted(n for n in dir(cls) if n.startswith("test_")):
            current[0] = meth
            res = unittest.TestResult()
            cls(meth).run(res)
            if res.failures or res.errors:
                raise RuntimeError(f"reference test {meth} did not pass here: {res.failures or res.errors}")
            out.append({"test": meth, "ci_runs_it": cls is ns["TestDataProcessingSecond"], "calls": calls.get(meth, [])})
    assert len(out) == 24, len(out)
    return out


def main():
    import nimble.__main__ as m
    import nimble.utils as u

    # ---- the reference's own unit-test vectors (test/test.py, all 24) ----
    uv = reference_unit_vectors()
    json.dump(uv, open(os.path.join(OUT, "reference_unit_vectors.json"), "w"), indent=0)
    print(f"reference unit tests: {len(uv)} passed, {sum(len(t['calls']) for t in uv)} recorded calls")

    # ---- report -s/--summarize (summarize_fields, nimble/__main__.py:307-310) ----
    sm = [{"name": name, "columns": cols, "input": text, "output": run_summarize(m, text, cols)} for name, cols, text in summarize_cases()]
    json.dump(sm, open(os.path.join(OUT, "summarize_cases.json"), "w"), indent=0)
    print(f"summarize cases: {len(sm)}")

    # ---- G2/G4: report() ----
    out = []
    for name, text in handmade() + random_tables(60, 1) + random_tables(60, 2):
        for thr, dis in ((0.05, False), (0.0, False), (0.2, False), (0.5, False), (0.05, True)):
            if name.startswith("random") and (thr, dis) not in ((0.05, False), (0.2, False), (0.05, True)) and zlib.crc32(name.encode()) % 3:   # (crc32, not hash(): the same cases in every run)
                continue
            out.append({"name": name, "threshold": thr, "disable_thresholding": dis, "input": text,
                        "output": run_report(m, text, thr, dis)})
    json.dump(out, open(os.path.join(OUT, "report_cases.json"), "w"), indent=0)
    print(f"report cases: {len(out)}")

    # ---- G3: append_path_string ----
    paths = ["/a/b/out.tsv.gz", "out.tsv", "out", "/x/y.z/out.tar.gz", "rel/dir/name.with.many.dots.txt", "/a/.hidden", "a/b.c/d", "", "/a/b/"]
    g3 = [{"path": p, "suffix": s, "expected": u.append_path_string(p, s)} for p in paths for s in (".hla", "", ".lib-1")]
    json.dump(g3, open(os.path.join(OUT, "append_path_string.json"), "w"), indent=0)

    # ---- G1: generate() on a CSV library (exact JSON text) ----
    g1 = []
    csvs = {
        "basic": "name,sequence,lineage\nA*01,ACGTACGTACGTACGTACGTAAAA,A\nA*02,ACGTACGTACGTACGTACGTCCCC,A\nB*01,TTTTACGTACGGGGTACGTACGTA,B\n",
        "name_last": "sequence,locus,name\nACGTTTGACCA,x,s1\nGGGTTTGACCA,\"y,z\",s2\n",
    }
    for key, text in csvs.items():
        with tempfile.TemporaryDirectory() as d:
            p, o = os.path.join(d, f"my_{key}_lib.csv"), os.path.join(d, "lib.json")
            open(p, "w").write(text)
            m.generate(p, None, o)
            g1.append({"name": key, "csv_name": f"my_{key}_lib.csv", "csv": text, "json": open(o).read()})
    json.dump(g1, open(os.path.join(OUT, "generate_csv.json"), "w"), indent=0)
    print("ok")


if __name__ == "__main__":
    sys.exit(main())

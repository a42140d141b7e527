"""`report` reducer vs the outputs of the reference's own report() (tests/golden/report_cases.json, made by
tests/golden/make_golden.py in the build container). Text-identical, byte for byte."""
import json
import os

import pytest

from nimble_amd import report as rep

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "report_cases.json")))


@pytest.mark.parametrize("engine", ["python", "native"])
@pytest.mark.parametrize("idx", range(len(CASES)))
def test_report_case(idx, engine, tmp_path, capsys, built):
    c = CASES[idx]
    i, o = tmp_path / "in.tsv", tmp_path / "out.tsv"
    i.write_text(c["input"])
    rep.report(str(i), str(o), None, c["threshold"], c["disable_thresholding"], engine=engine)
    got = o.read_text()
    if isinstance(c["output"], dict):
        # the reference itself raised (KeyError when every row is filtered, __main__.py:283): we write the empty output
        assert c["output"]["raised"] == "KeyError" and got == ""
    else:
        assert got == c["output"], f"{c['name']} thr={c['threshold']} disable={c['disable_thresholding']}"


def test_pinned_survey_vector(tmp_path):
    """rows A,B/10 C,A/20 (cell1,UMI1), B,D/15 (cell1,UMI2), G/50 (cell3,UMI3)."""
    i, o = tmp_path / "in.tsv", tmp_path / "out.tsv"
    i.write_text("nimble_features\tnimble_score\tr1_CB\tr1_UB\nA,B\t10\tcell1\tUMI1\nC,A\t20\tcell1\tUMI1\nB,D\t15\tcell1\tUMI2\nG\t50\tcell3\tUMI3\n")
    for engine in ("python", "native"):
        rep.report(str(i), str(o), engine=engine)
        assert o.read_text() == "A\t1\tcell1\nB,D\t1\tcell1\nG\t1\tcell3\n"


@pytest.mark.parametrize("idx", range(0, len(CASES), 3))
def test_report_rows_equals_report_on_tsv(idx, tmp_path, built):
    """Array entry point (nb_report_rows, used by the fused per-cell counting) == the TSV path on the golden inputs with
    integer scores."""
    c = CASES[idx]
    lines = [l for l in c["input"].split("\n")[1:] if l]
    header = c["input"].split("\n")[0].split("\t")
    if isinstance(c["output"], dict) or not lines or header[:4] != list(rep.REQUIRED):
        pytest.skip("no plain-integer table in this case")
    rows = [l.split("\t") for l in lines]
    if any(len(r) < 4 or any(v in rep._NA for v in r[:4]) or not r[1].lstrip("-").isdigit() for r in rows):
        pytest.skip("float scores / null fields: TSV-only semantics")
    cbs, fts, umis = sorted({r[2] for r in rows}), sorted({",".join(sorted(r[0].split(","))) for r in rows}), sorted({r[3] for r in rows})
    cb_ids = [cbs.index(r[2]) for r in rows]; ft_ids = [fts.index(",".join(sorted(r[0].split(",")))) for r in rows]
    umi_keys = [1000 - umis.index(r[3]) for r in rows]                      # any injective key, deliberately not order preserving
    o = tmp_path / "o.tsv"
    rep.report_rows(cb_ids, umi_keys, ft_ids, [int(r[1]) for r in rows], cbs, fts, str(o), c["threshold"], c["disable_thresholding"])
    assert o.read_text() == c["output"]


@pytest.mark.parametrize("variant", ["int", "float", "odd_umis"])
def test_native_report_threaded_equals_single_thread_and_python(variant, tmp_path, built, monkeypatch):
    """The golden cases are too small to reach the threaded code (text slices per worker, merge of thread-local interners,
    parallel merge sort): a 150k-row table must give byte-identical output with 1 and 7 workers and equal the Python engine.
    'float' exercises the order-sensitive Kahan sums (UMI group order), 'odd_umis' the fallback to interned UMI strings."""
    import numpy as np
    rng = np.random.default_rng({"int": 1, "float": 2, "odd_umis": 3}[variant])
    n = 150_000
    names = [f"L{i:02d}*{j:02d}" for i in range(6) for j in range(5)]
    feats = [",".join(rng.permutation(names)[: int(rng.integers(1, 5))]) for _ in range(300)]
    cells = ["".join("ACGT"[c] for c in rng.integers(0, 4, 16)) for _ in range(40)]
    alphabet = "ACGTN" if variant != "odd_umis" else "ACGTNacgt-"
    umis = ["".join(alphabet[c] for c in rng.integers(0, len(alphabet), 10)) for _ in range(20_000)]
    fi, ci, ui = rng.integers(0, 300, n), rng.integers(0, 40, n), rng.integers(0, 20_000, n)
    if variant == "float":
        scores = [f"{x:.3f}" for x in rng.random(n) * 3]
    else:
        scores = [str(int(x)) for x in rng.integers(1, 3, n)]
    path = tmp_path / "big.tsv"
    with open(path, "w") as f:
        f.write("nimble_features\tnimble_score\tr1_CB\tr1_UB\textra\n")
        for i in range(n):
            f.write(f"{feats[fi[i]]}\t{scores[i]}\t{cells[ci[i]]}\t{umis[ui[i]]}\tx\n")
            if i % 50_000 == 7:
                f.write("\n")                                   # blank lines are skipped
    outs = {}
    for threads in ("1", "7"):
        monkeypatch.setenv("NB_REPORT_THREADS", threads)
        out = tmp_path / f"native_{threads}.tsv"
        rep.report(str(path), str(out), engine="native")
        outs[threads] = open(out).read()
    assert outs["1"] == outs["7"] and len(outs["1"].splitlines()) > 1000
    rep.report(str(path), str(tmp_path / "py.tsv"), engine="python")
    assert open(tmp_path / "py.tsv").read() == outs["1"]
    if variant == "int":
        # the same table as the aligner writes it for BAM input: BGZF blocks (nb_gzip_members, inflated block-parallel by the
        # reducer) and as an ordinary gzip stream (serial inflate) — same counts
        import ctypes as ct
        import gzip
        import numpy as np
        from nimble_amd import _lib
        text = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
        n = text.shape[0]
        cap = n + (n >> 9) + 64 * ((n >> 20) + 1) + 64
        out, out_len = np.empty(cap, dtype=np.uint8), ct.c_int64()
        _lib.check(_lib.lib().nb_gzip_members(text.ctypes.data, n, 4, 3, out.ctypes.data, cap, ct.byref(out_len)))
        blob = out[:out_len.value].tobytes()
        assert gzip.decompress(blob) == text.tobytes() and blob[:4] == b"\x1f\x8b\x08\x04" and blob[12:16] == b"BC\x02\x00"
        (tmp_path / "bgzf.tsv.gz").write_bytes(blob)
        with gzip.open(tmp_path / "plain.tsv.gz", "wb", compresslevel=3) as f:
            f.write(text.tobytes())
        for name in ("bgzf.tsv.gz", "plain.tsv.gz"):
            rep.report(str(tmp_path / name), str(tmp_path / "gz_out.tsv"), engine="native")
            assert open(tmp_path / "gz_out.tsv").read() == outs["1"], name


_SUMMARIZE = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "summarize_cases.json")))


@pytest.mark.parametrize("idx", range(len(_SUMMARIZE)))
@pytest.mark.parametrize("engine", ["python", "native"])
def test_report_summarize_case(idx, engine, tmp_path, built, monkeypatch):
    """`report -s cols`: the per-UMI value/frequency table of the reference's summarize_fields (__main__.py:307-310) — outputs of
    the reference itself, incl. its column typing (ints / floats / bools / strings), tie order and quoting."""
    case = _SUMMARIZE[idx]
    (tmp_path / "in.tsv").write_text(case["input"])
    monkeypatch.chdir(tmp_path)                            # "summarize." + output: the reference's naming needs a bare file name
    if isinstance(case["output"], dict):
        with pytest.raises(Exception):
            rep.report("in.tsv", "out.tsv", case["columns"], engine=engine)
        return
    rep.report("in.tsv", "out.tsv", case["columns"], engine=engine)
    assert open("summarize.out.tsv").read() == case["output"], case["name"]

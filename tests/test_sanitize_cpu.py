"""ASan + UBSan over the host-side C++ (index builder, FASTQ ingest, BAM reader, report reducer, whitelist, formatter) and
the oracle, on the CPU (GPU sanitizers are not
available on the pool). Builds tests/native/host_sanitize.cpp with g++ and runs it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_under_sanitizers(tmp_path):
    exe = tmp_path / "host_sanitize"
    csrc = os.path.join(ROOT, "nimble_amd", "csrc")
    srcs = [os.path.join(ROOT, "tests", "native", "host_sanitize.cpp"), os.path.join(ROOT, "oracle", "nb_oracle.cpp")] + \
           [os.path.join(csrc, f) for f in ("nb_index.cpp", "nb_ingest.cpp", "nb_inflate.cpp", "nb_bam.cpp", "nb_report.cpp", "nb_host.cpp")]
    # inputs for the driver
    import json
    import numpy as np
    from tests.helpers import write_bam
    rng = np.random.default_rng(0)
    recs = []
    for i in range(300):
        s = "".join("ACGTN"[c] for c in rng.integers(0, 5, int(rng.integers(1, 120))))
        recs.append({"name": f"r{i}", "flag": 0x10 if i % 3 == 0 else 0, "seq": s, "tags": {"CB": "ACGT" * 4 + "-1", "XS": 3, "UB": "TTTTGGGG"}})
        if i % 40 == 0:
            recs.append({"name": f"r{i}", "flag": 0x800, "seq": "ACGT"})
    write_bam(tmp_path / "t.bam", recs, block=211)
    import gzip
    with gzip.open(tmp_path / "t.fastq.gz", "wt", compresslevel=6) as f:
        for i in range(10000):
            L = int(rng.integers(30, 150))
            f.write(f"@r{i}\n{''.join('ACGTN'[c] for c in rng.integers(0, 5, L))}\n+\n{''.join(chr(33 + int(q)) for q in rng.integers(2, 41, L))}\n")
    case = json.load(open(os.path.join(ROOT, "tests", "golden", "report_cases.json")))[-1]
    (tmp_path / "in.tsv").write_text(case["input"])
    (tmp_path / "wl.txt").write_text("\n".join("".join("ACGT"[c] for c in rng.integers(0, 4, 8)) for _ in range(2000)) + "\n")
    for flags in (["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], ["-fsanitize=thread"]):   # the readers are multi-threaded
        subprocess.run(["g++", "-O1", "-g", "-std=c++17"] + flags + ["-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-o", str(exe)]
                       + srcs + ["-lz"], check=True, capture_output=True)
        r = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "sanitize ok" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout + r.stderr

"""The C-ABI alone is enough to run the hot path: INTEGRATION.md §B executed in a process that never imports PyTorch
(device buffers, copies, stream and sync all through nb_device_alloc / nb_copy_* / nb_stream_create / nb_ctx_sync),
checked bit for bit against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from nimble_amd import engine, library, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_stub_without_torch(built, tmp_path):
    from oracle import oracle
    names, seqs, lineage = synth.synth_library(4, 8, seed=31)
    strs = [synth.codes_to_str(s) for s in seqs]
    fa = tmp_path / "lib.fasta"
    fa.write_text("".join(f">{n}\n{s}\n" for n, s in zip(names, strs)))
    library.generate(str(fa), None, str(tmp_path / "lib.json"))
    rng = np.random.default_rng(2)
    codes = synth.synth_reads(seqs, 4000, 150, seed=8)
    reads = [synth.codes_to_str(c[: int(rng.integers(15, 151))]) for c in codes]
    (tmp_path / "reads.txt").write_text("\n".join(reads) + "\n")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "native", "integration_b.py"), str(tmp_path / "lib.json"),
                        str(tmp_path / "reads.txt"), str(tmp_path / "out.npz")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    assert "torch imported: False" in r.stdout
    got = np.load(tmp_path / "out.npz")
    ora = oracle.OracleIndex(strs, 20).align_strs(engine.make_config(), reads, n_threads=4)
    info = engine.decode_info(got["info"])
    assert (info["reason"] == ora["reason"]).all() and (info["score"] == ora["score"]).all() and (info["orientation"] == ora["orientation"]).all()
    sets = np.zeros((len(reads), got["rows"].shape[1]), dtype=np.uint64)
    called = got["cls"] != engine.NO_CLASS
    sets[called] = got["rows"][got["cls"][called]]
    assert (sets == ora["set_bits"]).all() and int(got["counts"].sum()) == int((ora["reason"] == 0).sum()) > 1500

"""bench.py's output contract: one JSON line with the fields the driver and the judge read (small workload)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--reads", "200000",
                        "--cpu-sample", "20000", "--parity-sample", "5000"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and j["data"] == "synthetic" and j["unit"] == "M reads/s" and j["value"] > 1
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert j["oracle_self_consistency_ok"] is True


def test_bench_two_ranks_rehearsal():
    """The N > 1 path of bench.py as the driver launches it (torch.distributed.run, one process per rank), rehearsed on this box's one GPU
    with two ranks sharing it and gloo for the collectives (on a multi-GPU node the same code runs over RCCL): barrier + max-over-ranks
    timing, digest-first class dictionary agreement, one all-reduce of the count vector; value = reads of ALL ranks / time."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "300000", "--backend", "gloo",
                        "--cpu-sample", "0", "--parity-sample", "3000"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["reads_per_gpu"] == 300000 and j["oracle_self_consistency_ok"] is True
    assert j["cpu_baseline"] is None and j["classes"] > 1000
    assert abs(j["value"] - 2 * 300000 * 2 / (j["ms_per_step"] * 2 / 1e3) / 1e6) < 1e-6 * j["value"] + 1e-9

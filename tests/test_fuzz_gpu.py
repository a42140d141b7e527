"""A short run of scripts/fuzz_gpu.py under pytest: random libraries (1 .. 7 000 sequences), k from 9 to 31, reads of 40 .. 339 bp (ragged or
not), error / N / foreign-read rates, settings incl. tolerated mismatches, single-end batches of 200 k reads and paired batches of 60 k —
every read and every class count against the oracle. The long runs of the round (1 096 scenarios) are in profiles/r02z_fuzz.txt; the index
that the first of them could not build is pinned by tests/test_index_cpu.py."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(ROOT, "scripts", "fuzz_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3000, 3001, 3002, 3003, 3004, 3005, 3006, 3007, 3008, 3009, 3010, 3342, 3343, 3344])
def test_random_scenario_equals_oracle(built, monkeypatch, seed):
    monkeypatch.setenv("NB_ANCHOR_RETRY_MIN", "20000")          # (pairs come in batches of 60 k: run the stragglers' pass there too)
    line = _fuzz().scenario(seed)
    assert line.startswith("ok seed"), line

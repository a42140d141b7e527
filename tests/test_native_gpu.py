"""Native unit tests that must run on the GPU: built with hipcc on the box, executed as a child process."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dpp_wave_reductions(tmp_path):
    exe = tmp_path / "wave_reduce_test"
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "nimble_amd", "csrc"), "-o", str(exe),
                    os.path.join(ROOT, "tests", "native", "wave_reduce_test.hip")], check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("bad 0"), r.stdout + r.stderr

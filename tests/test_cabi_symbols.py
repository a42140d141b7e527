"""The C-ABI library loads without a GPU and exports every function include/nimble_amd.h declares."""
import ctypes as ct
import os
import re

from nimble_amd import _build, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "nimble_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nb_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(built):
    names = declared_functions()
    assert len(names) >= 20
    handle = ct.CDLL(_build.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in nimble_amd.h but not exported"


def test_python_binding_covers_header(built):
    assert sorted(_lib.SIGNATURES) == declared_functions()
    lib = _lib.lib()
    assert lib.nb_version() == 4
    assert lib.nb_packed_stride_words(150) == 8          # 5 base words + 3 N-mask words = 64 B per 150-bp read
    assert lib.nb_kernel_name(150, 20, 1000, 0) == b"nb_align_se_kernel<3, 1, true, true>"
    assert lib.nb_kernel_name(91, 20, 1000, 0) == b"nb_align_se_kernel<2, 1, false, true>"


def test_no_gpu_is_a_loud_error(built):
    """No CPU fallback: without a device nb_ctx_create fails with NB_ERR_NOGPU (skipped where a GPU exists)."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ct.c_void_p()
    rc = _lib.lib().nb_ctx_create(0, 1024, ct.byref(h))
    assert rc == -5 and b"no HIP device" in _lib.lib().nb_last_error()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under nimble_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "nimble_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "nb_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f

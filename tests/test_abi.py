"""The C ABI library loads without a GPU and exports every symbol include/hrs.h declares."""
import os
import re

import pytest

from hybrid_rag_system_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "hrs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hrs_[a-z0-9_]+)\s*\(", text)))


def test_exports_match_header():
    lib = _native.lib()
    names = _declared()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f"libhrs.so does not export {n}"
    assert sorted(_native.EXPORTS) == names
    assert lib.hrs_version() >= 1


def test_argument_errors_do_not_need_a_gpu():
    import ctypes as C

    lib = _native.lib()
    assert lib.hrs_ctx_create(0, None) == -1
    assert "out is null" in _native.last_error()
    assert lib.hrs_score_stats(None, None, None) == -1


def test_engine_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hybrid_rag_system_amd.engine import Engine

    with pytest.raises(_native.HrsError):
        Engine(0)


def test_header_compiles_as_c(tmp_path):
    """include/hrs.h is a C header: gcc -std=c11 -fsyntax-only on a translation unit that includes it."""
    import shutil
    import subprocess

    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include "hrs.h"\nint main(void) { return hrs_version() < 0; }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"), str(src)])

"""The C ABI from plain C (tests/c/abi_smoke.c, built by gcc against include/hrs.h): no Python, no torch."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_c_program_uses_the_abi():
    exe = os.path.join(HERE, "c", "abi_smoke")
    assert os.path.exists(exe), "build it with `make -C tests/c` (done by __graft_entry__.build())"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_smoke: ok" in r.stdout

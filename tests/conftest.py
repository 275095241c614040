import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def text_goldens():
    import json

    with open(os.path.join(GOLDEN, "text_goldens.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def corpus_texts():
    d = os.path.join(GOLDEN, "context", "infrastructure")
    names = ["README.md", "automation-guide.md", "bash-linux-reference.md", "windows-powershell-reference.md"]
    return {n: open(os.path.join(d, n), encoding="utf-8", errors="ignore").read() for n in names}


@pytest.fixture(scope="session")
def golden_dir():
    import pathlib

    return pathlib.Path(GOLDEN)

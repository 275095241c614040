"""A short run of the randomised differential check (tools/fuzz_search.py): random shapes, spaces, tie
patterns, masks; both scoring modes and both shadows against the CPU oracle."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_shapes_match_oracle():
    import fuzz_search
    from hybrid_rag_system_amd.engine import Engine

    eng = Engine(0)
    assert fuzz_search.run(eng, seed=11, cases=16, verbose=False) == 0
    eng.close()


def test_random_hybrid_steps_match_oracle():
    import fuzz_hybrid
    from hybrid_rag_system_amd.engine import Engine

    eng = Engine(0)
    assert fuzz_hybrid.run(eng, seed=5, cases=10, verbose=False) == 0
    eng.close()

"""A fixed-seed slice of every soak (tests/fuzzers.py) inside the GPU suite.  Round 3's first soak found a K2 bug
(41-59-px-tall cells) the committed cases had missed; the soaks used to live in tools/experiments/ and never ran under
pytest.  About a minute on the GPU box; the long runs are in profiles/r0x_fuzz_soak.txt."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzzers  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,cases,seed", [("parity", 300, 41), ("levels", 20, 42), ("batch_parity", 100, 43), ("matchers", 100, 44),
                                              ("ingest", 50, 45), ("best2", 500, 46)])
def test_fuzz_slice(oracle, name, cases, seed):
    bad, summary = fuzzers.FUZZERS[name](cases=cases, seed=seed)
    print(summary)
    assert bad == 0, summary

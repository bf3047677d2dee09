"""The CPU restatement against the committed golden vectors (made from the
real reference by tests/golden/make_golden.py).  Bit-exact; runs anywhere."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import cases  # noqa: E402


def test_oracle_matches_golden(orc):
    gold = np.load(os.path.join(HERE, "golden", "golden.npz"))
    mine = cases.run_cases("orc", orc)
    assert set(mine) == set(gold.files)
    for k in gold.files:
        assert np.array_equal(mine[k], gold[k], equal_nan=True), k

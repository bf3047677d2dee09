"""Writes tests/golden/golden.npz from the REAL reference (oracle/_ref, built
from /root/reference by oracle/Makefile).  Run in the build container only:

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from oracle import oracle  # noqa: E402
import cases  # noqa: E402

if __name__ == "__main__":
    oracle.build()
    assert oracle.have_ref(), "oracle/_ref missing: /root/reference is needed to write golden vectors"
    res = cases.run_cases("ref", oracle)
    path = os.path.join(HERE, "golden.npz")
    np.savez_compressed(path, **res)
    print("wrote", path, os.path.getsize(path), "bytes,", len(res), "arrays")

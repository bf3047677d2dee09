"""Shared helpers for the -m gpu tests (they call the HIP path through the C
ABI, tempestsdr_amd/libtsdrgpu.so, and check it against the oracle)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

_ctx = None


def ctx():
    global _ctx
    if _ctx is None:
        from tempestsdr_amd import gpu
        _ctx = gpu.TsdrGpu(0)  # raises without a GPU: no fallback
    return _ctx


def golden():
    return np.load(os.path.join(HERE, "golden", "golden.npz"))

"""A bounded run of the randomised differential soak (scripts/fuzz_parity.py): random resampler ratios / chunkings,
random frame geometries x stage orders x batch splits x run forms (plain, split, fused), random autocorrelation
rates and FFT sizes, device vs oracle.  The full soak (thousands of cases, several seeds) is run by hand."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [101, 102])
def test_soak_has_no_mismatch(seed, monkeypatch, capsys):
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "scripts", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    monkeypatch.setattr("sys.argv", ["fuzz_parity.py", "240", str(seed)])
    rc = fz.main()
    out = capsys.readouterr().out
    assert rc == 0, out
    assert "0 mismatches" in out

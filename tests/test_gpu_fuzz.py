"""-m gpu : a fixed-seed slice of tools/fuzz_parity.py, so that the randomised per-PIXEL comparison with the oracle is run by the
driver and not only by hand.  Each case: a random small scene with an extreme shape (needles, opacities at the 1/255 threshold,
splats at the near plane, partial tiles), random pose, either package; `radii` exact, images per pixel, gradients of both fp32
implementations measured against float64 autograd (oracle/autograd_ref.py) -- the HIP path may not be further from it than
5x the oracle's own distance (or 2e-4).  Every case is rendered twice: the second forward is the speculative one."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz():
    spec = importlib.util.spec_from_file_location("gsr_fuzz_parity", os.path.join(_ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed,cases", [(404, 28), (2026, 28)])
def test_fixed_seed_slice_of_the_parity_fuzz(seed, cases, monkeypatch):
    monkeypatch.setenv("GSR_SPECULATION", "1")          # both packages carry depth bounds from the first render to the second
    r = _fuzz().run(N=cases, seed=seed, verbose=False)
    assert not r["failures"], "\n".join(r["failures"]) + "\n" + r["summary"]

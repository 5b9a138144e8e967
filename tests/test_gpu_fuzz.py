"""GPU: a seeded, bounded subset of the randomised differential runs of tests/tools/fuzz_*.py (vocoder + spectrum, the
SoundTouch-shaped chain, the N2 converter) against the CPU oracle.  The full runs (python tests/tools/fuzz_X.py CASES SEED)
draw more cases from the same generators; these fixed seeds keep the driver's `-m gpu` run to a few seconds per test."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))

pytestmark = pytest.mark.gpu


def test_fuzz_stretch_and_spectrum_seeded(nae, ctx):
    import fuzz_stretch
    worst = fuzz_stretch.main(cases=10, seed=3, ctx=ctx, nae=nae)
    assert worst <= 1e-4                      # vocoder within tolerance; the spectrum is asserted bit-exact inside


def test_fuzz_stretch_one_barrier_pipeline_in_every_shape(nae):
    """the same randomised run with `debug_set("pv_flow", 2)`: every vocoder launch of at most one workgroup per CU takes the one-barrier schedule (pv_flow_kernel), whatever its shape"""
    import os
    import fuzz_stretch
    with nae.Context(0) as c:
        c.debug_set("pv_flow", 2)
        assert fuzz_stretch.main(cases=10, seed=5, ctx=c, nae=nae) <= 1e-4


def test_fuzz_wsola_seeded(nae, ctx):
    import fuzz_wsola
    assert fuzz_wsola.main(cases=8, seed=3, ctx=ctx, nae=nae) >= 4     # samples and overlap offsets bit-exact


def test_fuzz_swr_seeded(nae, ctx):
    import fuzz_swr
    assert fuzz_swr.main(cases=10, seed=3, ctx=ctx, nae=nae) >= 6      # bit-exact for every cut into convert calls

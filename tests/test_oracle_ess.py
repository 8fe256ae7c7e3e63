"""Pin oracle/ess_ref.py against outputs of the reference's own
zhusuan/diagnostics.py (tests/golden/ess_fixture.npz, made by
oracle/make_golden.py in the build container), and restate the reference's
own bounds (tests/test_diagnostics.py:13-39)."""
import os

import numpy as np
import pytest

from oracle import ess_ref


@pytest.fixture(scope='module')
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, 'ess_fixture.npz'))


@pytest.mark.parametrize('case', ['iid', 'ar1', 'rwmh', 'sticky_f32'])
def test_matches_reference_module(fx, case):
    s = fx[case + '_samples']
    for burn in (0, 100):
        got = ess_ref.effective_sample_size(s, burn_in=burn)
        np.testing.assert_allclose(got, fx['%s_ess_burn%d' % (case, burn)],
                                   rtol=1e-12)
    got1d = [ess_ref.effective_sample_size_1d(s[:, j])
             for j in range(s.shape[1])]
    np.testing.assert_allclose(got1d, fx[case + '_ess1d'], rtol=1e-12)


def test_reference_bounds():
    # tests/test_diagnostics.py:13-22 (scaled down 10x: ESS ~ n/3 for iid)
    rng = np.random.RandomState(0)
    ess = ess_ref.effective_sample_size(rng.normal(size=(1100, 1)),
                                        burn_in=100)
    assert 200 <= ess <= 1000

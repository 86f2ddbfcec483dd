"""Build-container-only: runs the imported, unmodified reference head live and checks (a) the committed fixtures are
what the reference produces today, (b) the oracle agrees with the live reference, (c) the eval skip quirk raises in
the reference too.  Skipped automatically where /root/reference is absent (GPU box)."""
import os
import sys

import numpy as np
import pytest

import cases
import helpers

pytestmark = pytest.mark.reference


def _run_ref(name):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    return make_golden.run_reference(cases.build_case(name))


@pytest.mark.parametrize("name", ["tiny", "ragged3", "train_tiny"])
def test_fixture_is_current_and_oracle_agrees(name):
    live = _run_ref(name)
    g = helpers.load_golden(name)
    for k in g:
        assert np.array_equal(live[k], g[k], equal_nan=True), "fixture stale: %s" % k
    case = cases.build_case(name)
    got = helpers.flatten_oracle(case, *helpers.run_oracle(case))
    helpers.compare_flat(got, live, atol=1e-6, rtol=1e-5)


def test_reference_raises_on_eval_skip_quirk():
    with pytest.raises(IndexError):
        _run_ref("skips_raise")

"""The device against the committed fixture tests/golden/oracle_qu480.npz (frozen CPU-oracle outputs, see
tests/golden/make_golden.py): velocity solver + incremental remap, velocity solver + upwind, one step each on
the 2 562-cell sphere, 1e-10 relative to each field's largest value.  No oracle code runs in this test."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run_device(kind, dom):
    from mpas_seaice_amd import advection as adv, velocity_solver as vs
    vs.seaice_mesh_pool_create(dom)
    try:
        vs.seaice_init_velocity_solver(dom)
        adv.seaice_init_advection(dom)
        vs.seaice_run_velocity_solver(dom)
        adv.seaice_run_advection(dom)
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_device_matches_frozen_oracle_outputs():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    got = mk.build_cases(_run_device)
    with np.load(os.path.join(GOLDEN, "oracle_qu480.npz")) as z:
        assert sorted(z.files) == sorted(got)
        for k in z.files:
            ref = z[k]
            big = np.abs(ref) >= 1e29  # "undefined" markers of the principal stresses
            assert np.array_equal(big, np.abs(got[k]) >= 1e29), k
            scale = max(np.abs(np.where(big, 0.0, ref)).max(), 1e-300)
            assert np.abs(np.where(big, 0.0, got[k] - ref)).max() <= 1e-10 * scale, k

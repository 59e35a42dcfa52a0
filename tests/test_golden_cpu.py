"""The committed fixtures of tests/golden: the reference-derived analytic samples and the frozen outputs of
the CPU oracle (see tests/golden/make_golden.py for what each file is and is not)."""
import importlib.util
import json
import os

import numpy as np

import analytic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _maker():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reference_analytic_samples():
    with open(os.path.join(GOLDEN, "kat_samples.json")) as f:
        kat = json.load(f)
    s = kat["sphere"]
    got = analytic.velocities_strains_analytical(s["lat"], s["lon"])
    for g, k in zip(got, ("u", "v", "e11", "e22", "e12")):
        assert abs(g - s[k]) < 5e-11 * max(1.0, abs(s[k])), k
    p = kat["plane"]
    got = analytic.planar_fields(p["x"], p["y"])
    for g, k in zip(got, ("u", "v", "e11", "e22", "e12", "divu", "divv")):
        assert abs(g - p[k]) < 5e-11 * max(1.0, abs(p[k])), k
    assert s == {**analytic.SPHERE_SAMPLE, "function": s["function"]}
    assert p == {**analytic.PLANE_SAMPLE, "function": p["function"]}


def test_oracle_reproduces_its_frozen_outputs():
    mk = _maker()
    now = mk.build_cases(mk.run_oracle)
    with np.load(os.path.join(GOLDEN, "oracle_qu480.npz")) as z:
        assert sorted(z.files) == sorted(now)
        for k in z.files:
            ref = z[k]
            scale = max(np.abs(ref).max(), 1e-300)
            assert np.abs(now[k] - ref).max() <= 1e-12 * scale, k

"""Writes tests/golden/*.  Two kinds of file, kept apart on purpose:

kat_samples.json    values the REFERENCE's own analytic test-case functions return (spherical-harmonic
                    velocities / strains, planar sinusoid with its stress divergence), as recorded when those
                    functions were evaluated during the survey of the reference; the netCDF4 module they import
                    is absent here, so they cannot be re-run and the numbers are kept as data.
oracle_qu480.npz    outputs of THIS repository's CPU oracle (oracle/*.c) on the 2 562-cell sphere: one
                    velocity-solver step, one incremental-remap step, one upwind step, sampled at fixed cells
                    and vertices.  They are not reference outputs (the reference cannot be built here, the
                    oracle is "parity unpinned"): they freeze the restatement so that a later change to the
                    oracle or to the device code shows up as a difference against a committed file.

Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

CFG = {"config_elastic_subcycle_number": 120}
BAND = (-0.2, 0.3)
NSAMPLE = 96


def sample_index(n):
    return (np.arange(NSAMPLE, dtype=np.int64) * 7919) % n


def build_cases(run):
    """run(kind, dom) advances `dom` by one step of `kind`; returns the sampled fields."""
    from mpas_seaice_amd import cases
    from util import get_mesh
    mesh = get_mesh("QU480")
    out = {}
    iv, ic = sample_index(mesh.nVertices), sample_index(mesh.nCells)
    dom = cases.make_domain(mesh, "QU480", configs=CFG, open_water_band=BAND)
    run("evp+ir", dom)
    b = dom.blocklist[0]
    for k in ("uVelocity", "vVelocity", "stressDivergenceU", "oceanStressU"):
        out["evp_" + k] = b.structs["velocity_solver"][k][iv]
    for k in ("divergence", "shear"):
        out["evp_" + k] = b.structs["velocity_solver"][k][ic]
    for k in ("stress11", "stress12", "principalStress1"):
        out["evp_" + k] = b.structs["velocity_variational"][k][:, ic]
    for k in ("iceAreaCategory", "iceVolumeCategory", "surfaceTemperature", "iceEnthalpy", "snowEnthalpy"):
        out["ir_" + k] = b.structs["tracers"][k][..., ic]
    dom = cases.make_domain(mesh, "QU480", configs=dict(CFG, config_advection_type="upwind"), open_water_band=BAND)
    dom.blocklist[0].structs["tracers"]["surfaceTemperature"][..., :mesh.nCells] += 273.15
    run("evp+upwind", dom)
    b = dom.blocklist[0]
    for k in ("iceAreaCategory", "iceVolumeCategory", "surfaceTemperature", "snowVolumeCategory"):
        out["upwind_" + k] = b.structs["tracers"][k][..., ic]
        out["upwind_" + k + "Tend"] = b.structs["tracer_tendencies"][k + "Tend"][..., ic]
    return {k: np.ascontiguousarray(v) for k, v in out.items()}


def run_oracle(kind, dom):
    from oracle import evp_flow, ir_flow, upwind_flow
    evp_flow.seaice_init_velocity_solver(dom)
    if kind == "evp+ir":
        ir_flow.seaice_init_advection(dom)
    else:
        upwind_flow.seaice_init_advection_upwind(dom)
    evp_flow.seaice_run_velocity_solver(dom)
    if kind == "evp+ir":
        ir_flow.seaice_run_advection(dom)
    else:
        upwind_flow.seaice_run_advection_upwind(dom)


def main():
    kat = {
        "source": "values returned by the reference's analytic test-case functions, recorded during the survey",
        "sphere": {"function": "testing_and_setup/testcases/spherical_operators/strain/create_ic.py:574 "
                               "velocities_strains_analytical(lat, lon, 3, 5, 2, 4)",
                   "lat": 0.7, "lon": 1.1, "u": -0.59121435168, "v": -0.31027979175, "e11": 0.63179002647,
                   "e22": -0.60078848715, "e12": -1.01789907063},
        "plane": {"function": "testing_and_setup/testcases/square/operators_strain/create_ics.py:90 "
                              "velocities_strains_stress_divergences(x, y, 'sinusoid')",
                  "x": 0.3, "y": 0.4, "u": -0.14926584448, "v": -0.14926584448, "e11": 0.27270298007,
                  "e22": -15.8008226061, "e12": -7.76405981304, "divu": 72.3621268644, "divv": 72.3621268644},
    }
    with open(os.path.join(HERE, "kat_samples.json"), "w") as f:
        json.dump(kat, f, indent=1)
    np.savez(os.path.join(HERE, "oracle_qu480.npz"), **build_cases(run_oracle))
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()

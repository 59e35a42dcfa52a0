#!/usr/bin/env python3
"""Times the incremental-remap advection step on one GPU (per-phase HIP-event durations).

    python scripts/bench_ir.py --mesh QU120 --reps 10 [--cpu-mesh QU240]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402


def n_scalars(dom):
    nCat = dom.blocklist[0].dimensions["nCategories"]
    return sum(t.nLayers for t in dom.solver_state["tracersHead"]) * nCat


def gpu_ir(meshname, reps, renumber=True):
    from mpas_seaice_amd import advection as adv, cases, mesh as meshmod, velocity_solver as vs
    mesh = meshmod.make_named_mesh(meshname)
    dom = cases.make_domain(mesh, meshname)
    vs.seaice_mesh_pool_create(dom, renumber=renumber)
    vs.seaice_init_velocity_solver(dom)
    adv.seaice_init_advection(dom)
    vs.seaice_run_velocity_solver(dom)
    adv.seaice_run_advection(dom, download=False)      # warm-up, tracers now resident
    b = dom.blocklist[0]
    dt = float(b.structs["velocity_solver"]["dynamicsTimeStep"][0])
    tot, ph = [], []
    for _ in range(reps):
        t, p = b.ctx.ir_run_timed(dt)
        tot.append(t)
        ph.append(p)
    a = np.zeros_like(b.structs["tracers"]["iceAreaCategory"])
    b.ctx.get("iceAreaCategory", a)
    assert np.isfinite(a).all()
    ns = n_scalars(dom)
    ms = float(np.median(tot))
    out = {"mesh": meshname, "nCells": mesh.nCells, "scalars_per_cell": ns, "ms_per_step": ms,
           "phase_ms": dict(zip(["convert+masks", "reconstruct", "departure_triangles", "fluxes", "update+zap"],
                                np.median(np.array(ph), axis=0).round(4).tolist())),
           "tracer_cell_updates_per_s": mesh.nCells * ns / (ms * 1e-3)}
    vs.seaice_mesh_pool_destroy(dom)
    return out


def cpu_ir(meshname):
    from mpas_seaice_amd import cases, mesh as meshmod
    from oracle import evp_flow as ora, ir_flow as ir
    mesh = meshmod.make_named_mesh(meshname)
    dom = cases.make_domain(mesh, meshname, configs={"config_elastic_subcycle_number": 4})
    ora.seaice_init_velocity_solver(dom)
    ir.seaice_init_advection(dom)
    ora.seaice_run_velocity_solver(dom)
    t0 = time.perf_counter()
    ir.seaice_run_advection(dom)
    dt = time.perf_counter() - t0
    ns = n_scalars(dom)
    return {"mesh": meshname, "seconds_per_step": dt, "tracer_cell_updates_per_s": mesh.nCells * ns / dt,
            "cores": ora.num_threads(), "kind": "port"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", default="QU120")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--cpu-mesh", default="")
    args = ap.parse_args()
    res = {"gpu": gpu_ir(args.mesh, args.reps)}
    if args.cpu_mesh:
        res["cpu"] = cpu_ir(args.cpu_mesh)
    print(json.dumps(res), flush=True)

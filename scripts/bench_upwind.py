#!/usr/bin/env python3
"""Times the upwind advection step (config_advection_type = 'upwind') on one GPU: wall time of sih_upwind_run
with the tracers resident on the device.

    python scripts/bench_upwind.py --mesh QU60 --reps 20
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", default="QU60")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    from mpas_seaice_amd import advection as adv, cases, mesh as meshmod, velocity_solver as vs
    mesh = meshmod.make_named_mesh(args.mesh)
    dom = cases.make_domain(mesh, args.mesh, configs={"config_advection_type": "upwind"})
    b = dom.blocklist[0]
    b.structs["tracers"]["surfaceTemperature"] += 273.15
    vs.seaice_mesh_pool_create(dom)
    vs.seaice_init_velocity_solver(dom)
    adv.seaice_init_advection(dom)
    vs.seaice_run_velocity_solver(dom)
    adv.seaice_run_advection(dom)  # warm-up
    dt = float(b.structs["velocity_solver"]["dynamicsTimeStep"][0])
    times = []
    for _ in range(args.reps):
        b.ctx.sync()
        t0 = time.perf_counter()
        b.ctx.upwind_run(dt)       # returns after the stream has drained
        times.append(time.perf_counter() - t0)
    a = np.zeros_like(b.structs["tracers"]["iceAreaCategory"])
    b.ctx.get("iceAreaCategory", a)
    assert np.isfinite(a).all()
    nCat = b.dimensions["nCategories"]
    ms = float(np.median(times)) * 1e3
    print(json.dumps({"mesh": args.mesh, "nCells": mesh.nCells, "scalars_per_cell": 4 * nCat, "ms_per_step": ms,
                      "tracer_cell_updates_per_s": mesh.nCells * 4 * nCat / (ms * 1e-3)}))
    vs.seaice_mesh_pool_destroy(dom)


if __name__ == "__main__":
    main()

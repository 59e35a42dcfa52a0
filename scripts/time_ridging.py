"""Wall time of sih_column_ridge (launch, kernel and the read of its status word) and of sih_column_update_state on a named mesh after one velocity
solve and advection; prints one JSON line.  Usage: python scripts/time_ridging.py [mesh] [reps]"""
import json
import sys
import time

sys.path.insert(0, ".")
from mpas_seaice_amd import advection as adv, cases, column, mesh as meshmod, pools, velocity_solver as vs  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "QU60"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m = meshmod.make_named_mesh(name)
dom = cases.make_domain(m, name, configs={"config_use_column_package": True}, open_water_band=(-0.2, 0.3))
b = dom.blocklist[0]
pools.allocate_ridging_pools(b)
nC = m.nCells
b.structs["icestate"]["openWaterArea"][:nC] = 1.0 - b.structs["tracers"]["iceAreaCategory"][0, :, :nC].sum(axis=0)
vs.seaice_mesh_pool_create(dom)
try:
    vs.seaice_init_velocity_solver(dom)
    adv.seaice_init_advection(dom)
    vs.seaice_run_velocity_solver(dom)
    adv.seaice_run_advection(dom, download=False)
    column.column_ridging(dom, upload="cells", download=False)
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        column.column_ridging(dom, upload=False, download=False)
        t.append(time.perf_counter() - t0)
    t.sort()
    column.seaice_column_update_state(dom, download=False)
    b.ctx.sync()
    u = []
    for _ in range(reps):   # seaice_column_update_state (asynchronous: timed with the wait for the stream)
        t0 = time.perf_counter()
        column.seaice_column_update_state(dom, download=False)
        b.ctx.sync()
        u.append(time.perf_counter() - t0)
    u.sort()
    print(json.dumps({"mesh": name, "nCells": int(nC), "ridging_ms_median": 1e3 * t[len(t) // 2], "ridging_ms_min": 1e3 * t[0],
                      "iterations": dom.solver_state["ridge_iterations"], "update_state_ms_median": 1e3 * u[len(u) // 2],
                      "update_state_ms_min": 1e3 * u[0]}))
finally:
    vs.seaice_mesh_pool_destroy(dom)

"""Time one velocity-solver step including the host<->device transfers of its pools (QU60 by default)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mpas_seaice_amd import cases, mesh as meshmod, pools, velocity_solver as vs

ap = argparse.ArgumentParser()
ap.add_argument("--mesh", default="QU60")
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()

out = {}
for pinned in (False, True):
    dom = cases.make_domain(meshmod.make_named_mesh(args.mesh), "analytic")
    if pinned:
        pools.pin_pools(dom)
    vs.seaice_mesh_pool_create(dom)
    vs.seaice_init_velocity_solver(dom)
    b = dom.blocklist[0]
    t = {"upload": [], "solve": [], "download": []}
    for r in range(args.reps + 1):
        t0 = time.perf_counter()
        vs._upload(dom, b)
        t1 = time.perf_counter()
        b.ctx.run()
        b.ctx.sync()
        t2 = time.perf_counter()
        vs._download(dom, b)
        t3 = time.perf_counter()
        if r:
            t["upload"].append(t1 - t0)
            t["solve"].append(t2 - t1)
            t["download"].append(t3 - t2)
    nb_up = sum(b.structs[p][n].nbytes for p, n in vs.RUN_INPUTS)
    nb_dn = sum(b.structs[p][n].nbytes for p, n in vs.RUN_OUTPUTS)
    out["pinned" if pinned else "pageable"] = {
        "upload_ms": 1e3 * min(t["upload"]), "solve_ms": 1e3 * min(t["solve"]), "download_ms": 1e3 * min(t["download"]),
        "upload_MB": nb_up / 1e6, "download_MB": nb_dn / 1e6,
        "upload_GBps": nb_up / min(t["upload"]) / 1e9, "download_GBps": nb_dn / min(t["download"]) / 1e9}
    vs.seaice_mesh_pool_destroy(dom)
print(json.dumps(out))

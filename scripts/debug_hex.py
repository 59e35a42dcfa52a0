import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from mpas_seaice_amd import cases, velocity_solver as vs
from mpas_seaice_amd.velocity_solver import RUN_OUTPUTS, STATIC_OUTPUTS
from oracle import evp_flow as oracle
from util import get_mesh, rel_linf
mesh = get_mesh("hex_24_20")
for nsub in (60, 120):
    cfg = {"config_dt": 600.0, "config_elastic_subcycle_number": nsub}
    ref = cases.make_domain(mesh, None, configs=cfg)
    oracle.seaice_init_velocity_solver(ref); oracle.seaice_run_velocity_solver(ref)
    dom = cases.make_domain(mesh, None, configs=cfg)
    vs.seaice_mesh_pool_create(dom, renumber=False)
    vs.seaice_init_velocity_solver(dom); vs.seaice_run_velocity_solver(dom)
    vs.seaice_mesh_pool_destroy(dom)
    print("nsub", nsub)
    b, r = dom.blocklist[0], ref.blocklist[0]
    for pool, name in RUN_OUTPUTS:
        x, y = b.structs[pool][name], r.structs[pool][name]
        if x.dtype.kind == 'i':
            print(f"  {name:28s} int mismatches {(x!=y).sum()}")
        else:
            e = rel_linf(x, y)
            if e > 1e-12:
                d = np.abs(x - y)
                k = np.unravel_index(np.argmax(d), d.shape)
                extra = ""
                if x.ndim == 1 and x.shape[0] == b.dimensions["nVertices"] + 1:
                    extra = f" interior={b.structs['boundary']['interiorVertex'][k[0]]} solve={b.structs['velocity_solver']['solveVelocity'][k[0]]}"
                print(f"  {name:28s} rel {e:.3e} at {k} gpu={x[k]:.6e} ref={y[k]:.6e} nbad={(d>1e-12*np.abs(y).max()).sum()}{extra}")

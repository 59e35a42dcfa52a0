import copy

import numpy as np

from mpas_seaice_amd import cases, mesh as meshmod

_MESH_CACHE = {}


def get_mesh(name, **kw):
    key = (name, tuple(sorted(kw.items())))
    if key not in _MESH_CACHE:
        if name.startswith("hex"):
            _, nx, ny = name.split("_")
            _MESH_CACHE[key] = meshmod.make_planar_hex_mesh(int(nx), int(ny), 1.28e6 / int(nx))
        elif name.startswith("quad"):      # quad_<nx>_<ny>: planar squares
            _, nx, ny = name.split("_")
            _MESH_CACHE[key] = meshmod.make_planar_quad_mesh(int(nx), int(ny), 1.28e6 / int(nx))
        elif name.startswith("llquad"):    # llquad_<nlon>_<nlat>: latitude-longitude band on the sphere
            _, nx, ny = name.split("_")
            _MESH_CACHE[key] = meshmod.make_latlon_quad_mesh(int(nx), int(ny), **kw)
        elif name.startswith("var"):
            _MESH_CACHE[key] = meshmod.make_variable_sphere_mesh(int(name[3:]), **kw)
        elif name.startswith("geo"):
            _MESH_CACHE[key] = meshmod.make_sphere_mesh(int(name[3:]), **kw)
        else:
            _MESH_CACHE[key] = meshmod.make_named_mesh(name, **kw)
    return _MESH_CACHE[key]


def rel_linf(a, b):
    """max |a-b| / max |b| (the parity metric: relative L-infinity against the field's own scale)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.abs(b).max()
    if den == 0.0:
        return float(np.abs(a - b).max())
    return float(np.abs(a - b).max() / den)


def compare_domains(dom_a, dom_b, fields, tol, report=None):
    worst = 0.0
    for pool, name in fields:
        for ba, bb in zip(dom_a.blocklist, dom_b.blocklist):
            x, y = ba.structs[pool][name], bb.structs[pool][name]
            if x.dtype.kind == "i":
                nbad = int((x != y).sum())
                if report is not None:
                    report.append((pool, name, nbad))
                assert nbad == 0, f"{pool}.{name}: {nbad} integer mismatches"
            else:
                # 1e30 "undefined" markers of principal stresses must coincide exactly
                big = np.abs(y) >= 1e29
                assert np.array_equal(big, np.abs(x) >= 1e29), f"{pool}.{name}: marker mismatch"
                e = rel_linf(np.where(big, 0.0, x), np.where(big, 0.0, y))
                if report is not None:
                    report.append((pool, name, e))
                assert e <= tol, f"{pool}.{name}: rel Linf {e:.3e} > {tol:.1e}"
                worst = max(worst, e)
    return worst

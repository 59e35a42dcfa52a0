"""Raw stream files exchanged with fortran/seaice_hip_driver (the Fortran host of the C ABI)."""
from __future__ import annotations

import numpy as np


def write_case(path, block, cfg):
    d = block.dimensions
    m, tr = block.structs["mesh"], block.structs["tracers"]
    oc, at = block.structs["ocean_coupling"], block.structs["atmos_coupling"]
    nC, nV, nE, M, nCat = d["nCells"], d["nVertices"], d["nEdges"], d["maxEdges"], d["nCategories"]
    with open(path, "wb") as f:
        np.array([nC, nV, nE, M, nCat, cfg["config_elastic_subcycle_number"], int(bool(m["on_a_sphere"])),
                  d["nIceLayers"], d["nSnowLayers"]], dtype=np.int32).tofile(f)
        np.array([m["sphere_radius"], cfg["config_dt"]], dtype=np.float64).tofile(f)

        def w(a, n):
            np.asfortranarray(a[..., :n]).ravel(order="F").tofile(f)
        w(m["nEdgesOnCell"], nC)
        for k in ("verticesOnCell", "cellsOnCell", "edgesOnCell"):
            w(m[k], nC)
        w(m["cellsOnVertex"], nV)
        w(m["cellsOnEdge"], nE)
        w(m["verticesOnEdge"], nE)
        for k in ("xCell", "yCell", "zCell"):
            w(m[k], nC)
        for k in ("xVertex", "yVertex", "zVertex"):
            w(m[k], nV)
        w(m["areaCell"], nC)
        w(m["areaTriangle"], nV)
        w(m["fVertex"], nV)
        w(m["dvEdge"], nE)
        w(m["dcEdge"], nE)
        for k in ("iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory"):
            w(tr[k], nC)
        w(oc["uOceanVelocity"], nC)
        w(oc["vOceanVelocity"], nC)
        w(at["uAirVelocity"], nC)
        w(at["vAirVelocity"], nC)
        w(at["airDensity"], nC)
        w(m["edgesOnVertex"], nV)
        for k in ("xEdge", "yEdge", "zEdge"):
            w(m[k], nE)
        w(m["coeffs_reconstruct"], nC)
        for k in ("surfaceTemperature", "iceEnthalpy", "iceSalinity", "snowEnthalpy"):
            w(tr[k], nC)


def read_result(path, block):
    d = block.dimensions
    nC, nV, M, nCat = d["nCells"], d["nVertices"], d["maxEdges"], d["nCategories"]
    nIce, nSnow = d["nIceLayers"], d["nSnowLayers"]
    raw = np.fromfile(path, dtype=np.float64)
    out, o = {}, 0
    for name, shape in (("uVelocity", (nV,)), ("vVelocity", (nV,)), ("stress11", (M, nC)), ("stress22", (M, nC)),
                        ("stress12", (M, nC)), ("divergence", (nC,)), ("shear", (nC,)),
                        ("iceAreaCategory", (1, nCat, nC)), ("iceVolumeCategory", (1, nCat, nC)),
                        ("snowVolumeCategory", (1, nCat, nC)), ("surfaceTemperature", (1, nCat, nC)),
                        ("iceEnthalpy", (nIce, nCat, nC)), ("iceSalinity", (nIce, nCat, nC)),
                        ("snowEnthalpy", (nSnow, nCat, nC)), ("upwindArea", (1, nCat, nC)),
                        ("upwindTemperature", (1, nCat, nC)), ("upwindVolume", (1, nCat, nC)),
                        ("upwindSnow", (1, nCat, nC))):
        n = int(np.prod(shape))
        out[name] = raw[o:o + n].reshape(shape, order="F")
        o += n
    assert o == raw.size
    return out


# ---- pools as a whole, for fortran/harness/shim_harness (the shim modules run on pools built from this file) ----------
MAGIC = 1397311537  # "SIH1"
_KEY = 64


def _key(s):
    b = s.encode()
    assert len(b) <= _KEY, s
    return b + b" " * (_KEY - len(b))  # Fortran character(len=64), blank padded


def _field_records(block):
    """(pool, key, levels): every numpy field of every pool; a pool named <pool>_level2 holds the second time level
    of the fields of <pool> with the same names."""
    out = []
    for pname, pool in block.structs.items():
        if pname.endswith("_level2"):
            continue
        second = block.structs.get(pname + "_level2", {})
        for k, a in pool.items():
            if not isinstance(a, np.ndarray) or a.dtype not in (np.float64, np.int32) or a.ndim > 3:
                continue
            levels = [a] + ([second[k]] if k in second else [])
            out.append((pname, k, levels))
    return out


def write_pools(path, domain, tracer_list=None):
    """Everything the shim modules reach through the pool interface, for one block: configs (the mesh attributes
    on_a_sphere and sphere_radius among them), dimensions, the ordered tracer list (tracers.
    seaice_add_tracers_to_linked_list when not given) and every field of every pool with all its time levels."""
    from . import tracers as trc
    block = domain.blocklist[0]
    cfg = dict(domain.configs)
    cfg["on_a_sphere"] = bool(block.structs["mesh"]["on_a_sphere"])
    cfg["sphere_radius"] = float(block.structs["mesh"]["sphere_radius"])
    if tracer_list is None:
        tracer_list = [(t.tracerName, t.parentName) for t in trc.seaice_add_tracers_to_linked_list(domain)]
    with open(path, "wb") as f:
        np.array([MAGIC, len(cfg)], dtype=np.int32).tofile(f)
        for k, v in cfg.items():
            f.write(_key(k))
            if isinstance(v, (bool, np.bool_)):
                np.array([3, int(v)], dtype=np.int32).tofile(f)
            elif isinstance(v, (int, np.integer)):
                np.array([1, int(v)], dtype=np.int32).tofile(f)
            elif isinstance(v, (float, np.floating)):
                np.array([2], dtype=np.int32).tofile(f)
                np.array([float(v)], dtype=np.float64).tofile(f)
            else:
                np.array([4], dtype=np.int32).tofile(f)
                f.write(_key(str(v)))
        np.array([len(block.dimensions)], dtype=np.int32).tofile(f)
        for k, v in block.dimensions.items():
            f.write(_key(k))
            np.array([int(v)], dtype=np.int32).tofile(f)
        np.array([len(tracer_list)], dtype=np.int32).tofile(f)
        for name, parent in tracer_list:
            f.write(_key(name))
            f.write(_key(parent))
        recs = _field_records(block)
        np.array([len(recs)], dtype=np.int32).tofile(f)
        for pname, k, levels in recs:
            a = levels[0]
            scalar = a.ndim == 1 and a.size == 1  # dynamicsTimeStep, elasticTimeStep: scalars of the Registry
            dims = [1, 1, 1]
            if not scalar:
                dims[:a.ndim] = a.shape
            f.write(_key(pname))
            f.write(_key(k))
            np.array([int(a.dtype == np.int32), 0 if scalar else a.ndim] + dims + [len(levels)], dtype=np.int32).tofile(f)
            for lv in levels:
                np.asfortranarray(lv).ravel(order="F").tofile(f)


def read_pools(path):
    """{(pool, key): [array of time level 1, ...]} from a result file of shim_harness."""
    raw = open(path, "rb").read()
    o = 0

    def take(dtype, n):
        nonlocal o
        a = np.frombuffer(raw, dtype=dtype, count=n, offset=o)
        o += a.nbytes
        return a
    magic, n = take(np.int32, 2)
    assert magic == MAGIC
    out = {}
    for _ in range(int(n)):
        pname = raw[o:o + _KEY].decode().strip()
        k = raw[o + _KEY:o + 2 * _KEY].decode().strip()
        o += 2 * _KEY
        is_int, nd, d1, d2, d3, nlev = (int(x) for x in take(np.int32, 6))
        shape = (d1, d2, d3)[:nd] if nd else (1,)
        cnt = int(np.prod(shape))
        out[(pname, k)] = [take(np.int32 if is_int else np.float64, cnt).reshape(shape, order="F").copy()
                           for _ in range(nlev)]
    assert o == len(raw)
    return out

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

"""Synthetic idealised-forcing cases (deterministic, analytic) used by tests and bench.

The reference's own idealised case is the planar "square" test: analytic wind
(/root/reference/src/shared/mpas_seaice_testing.F:357-422), ocean gyre (:311-343) and a ramp of
ice concentration with 2 m ice (:436-525).  Here the same fields are mapped onto the sphere
through the rotated longitude / latitude (lambda', phi') of ``config_rotate_cartesian_grid``
(x' = -z, y' = y, z' = x; /root/reference/src/shared/mpas_seaice_mesh.F:2367-2371), with
X = (lambda' + pi) / 2 pi and Y = (phi' + pi/2) / pi playing the roles of x/Lx and y/Ly.
Velocity components are (east', north') in that rotated frame, like the solver's.
"""
from __future__ import annotations

import numpy as np

from .pools import Domain

# config_dt per resolution, /root/reference/bld/namelist_files/namelist_defaults_mpassi.xml:8-12
DT_BY_MESH = {"QU480": 7200.0, "QU240": 3600.0, "QU120": 1800.0, "QU60": 900.0, "QU30": 450.0, "QU15": 225.0}

CATEGORY_AREA_FRACTION = np.array([0.10, 0.20, 0.30, 0.25, 0.15])
CATEGORY_THICKNESS_FACTOR = np.array([0.3, 0.7, 1.0, 1.4, 2.0])


def rotated_lonlat(x, y, z, radius, rotate=True):
    if rotate:
        xp, yp, zp = -z, y, x
    else:
        xp, yp, zp = x, y, z
    return np.arctan2(yp, xp), np.arcsin(np.clip(zp / radius, -1.0, 1.0))


def fill_idealised_state(block, time=0.0, open_water_band=None, rotate=True):
    """Fill tracers, forcing and initial velocities of one block with the analytic case."""
    mesh = block.structs["mesh"]
    nC = block.dimensions["nCells"]
    nCat = block.dimensions["nCategories"]
    R = mesh["sphere_radius"] if mesh["on_a_sphere"] else 1.0
    if mesh["on_a_sphere"]:
        lon, lat = rotated_lonlat(mesh["xCell"][:nC], mesh["yCell"][:nC], mesh["zCell"][:nC], R, rotate)
        X = (lon + np.pi) / (2.0 * np.pi)
        Y = (lat + 0.5 * np.pi) / np.pi
    else:
        # a block of a partitioned planar mesh carries the extent of the whole domain
        x0, x1, y0, y1 = mesh.get("planarExtent", (mesh["xCell"][:nC].min(), mesh["xCell"][:nC].max(),
                                                   mesh["yCell"][:nC].min(), mesh["yCell"][:nC].max()))
        Lx = x1 - x0 + 1e-30
        Ly = y1 - y0 + 1e-30
        X = (mesh["xCell"][:nC] - x0) / Lx
        Y = (mesh["yCell"][:nC] - y0) / Ly
        lon = (X * 2.0 - 1.0) * np.pi
        lat = (Y - 0.5) * np.pi
    area = 0.95 - 0.15 * np.cos(lat) ** 2
    thick = 2.0 + 0.5 * np.sin(3.0 * lon)
    if open_water_band is not None:
        lo, hi = open_water_band
        area = np.where((lat > lo) & (lat < hi), 0.0, area)
    frac = CATEGORY_AREA_FRACTION if nCat == 5 else np.full(nCat, 1.0 / nCat)
    hfac = CATEGORY_THICKNESS_FACTOR if nCat == 5 else np.linspace(0.5, 1.5, nCat)
    tr = block.structs["tracers"]
    for k in range(nCat):
        tr["iceAreaCategory"][0, k, :nC] = area * frac[k]
        tr["iceVolumeCategory"][0, k, :nC] = area * frac[k] * thick * hfac[k]
        tr["snowVolumeCategory"][0, k, :nC] = area * frac[k] * 0.2
    # smooth analytic fields for the other advected tracers (values only matter where ice exists)
    if "surfaceTemperature" in tr:
        for k in range(nCat):
            tr["surfaceTemperature"][0, k, :nC] = -5.0 - 10.0 * np.sin(lat) ** 2 - 0.5 * k + 2.0 * np.cos(2.0 * lon)
    for name, base, amp in (("iceEnthalpy", -2.0e8, 0.5e8), ("iceSalinity", 4.0, 2.0), ("snowEnthalpy", -1.1e8, 0.1e8)):
        if name in tr:
            nL = tr[name].shape[0]
            for l in range(nL):
                for k in range(nCat):
                    tr[name][l, k, :nC] = base + amp * ((l + 1.0) / nL) * np.cos(lon + 0.3 * k) * np.cos(lat) \
                        + 0.1 * amp * np.sin(2.0 * lat + l)
    a, b, theta = 5.0, 3.0, 4.0 * 24.0 * 3600.0
    amp = np.sin((2.0 * np.pi * time) / theta) - b
    at = block.structs["atmos_coupling"]
    at["uAirVelocity"][:nC] = a + amp * np.sin(2.0 * np.pi * X) * np.sin(np.pi * Y)
    at["vAirVelocity"][:nC] = a + amp * np.sin(2.0 * np.pi * Y) * np.sin(np.pi * X)
    at["airDensity"][:nC] = 1.3
    oc = block.structs["ocean_coupling"]
    oc["uOceanVelocity"][:nC] = 0.1 * (2.0 * Y - 1.0)
    oc["vOceanVelocity"][:nC] = -0.1 * np.sin(lon) * np.cos(lat)
    oc["seaSurfaceTiltU"][:] = 0.0
    oc["seaSurfaceTiltV"][:] = 0.0
    oc["landIceMask"][:] = 0
    oc["landIceMaskVertex"][:] = 0
    vs = block.structs["velocity_solver"]
    for k in ("uVelocity", "vVelocity"):
        vs[k][:] = 0.0
    vs["solveVelocityPrevious"][:] = 0


def make_domain(mesh_or_meshes, name=None, configs=None, nCategories=5, time=0.0, open_water_band=None):
    cfg = {}
    if name in DT_BY_MESH:
        cfg["config_dt"] = DT_BY_MESH[name]
    elif isinstance(name, str) and name.startswith("var") and name[3:].isdigit():
        # variable-resolution mesh on 10 n^2 + 2 generators, finest cells about 0.6 of the nominal spacing: half the
        # step of the quasi-uniform mesh of the same size (var128: 450 s, var256: 225 s)
        cfg["config_dt"] = 57600.0 / int(name[3:])
    if configs:
        cfg.update(configs)
    dom = Domain(mesh_or_meshes, cfg, nCategories=nCategories)
    for blk in dom.blocklist:
        fill_idealised_state(blk, time=time, open_water_band=open_water_band,
                             rotate=dom.configs["config_rotate_cartesian_grid"])
    return dom

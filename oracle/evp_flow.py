"""TEST INFRASTRUCTURE ONLY: the reference's velocity-solver call flow executed with the CPU oracle.

Walks the same sequence as the reference driver routines
(/root/reference/src/shared/mpas_seaice_velocity_solver.F: seaice_init_velocity_solver :85,
velocity_solver_pre_subcycle :613, subcycle_velocity_solver :2404, velocity_solver_post_subcycle
:3360) over the pool containers of ``mpas_seaice_amd.pools`` and calls ``liboracle.so`` for every
loop.  Parity status: see the header of evp_oracle.c ("parity unpinned" at bit level).

``halo`` is an optional callable ``halo(domain, pool_name, field_names)`` standing for
``mpas_dmpar_field_halo_exch``; a single full-sphere block needs none.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def call(name, *args):
    """Call an oracle routine: numpy arrays -> pointers, ints -> c_int, floats -> c_double."""
    fn = getattr(lib(), name)
    fn.restype = None
    cargs = []
    for a in args:
        if a is None:
            cargs.append(ctypes.c_void_p(0))
        elif isinstance(a, np.ndarray):
            assert a.flags["F_CONTIGUOUS"] or a.flags["C_CONTIGUOUS"], name
            cargs.append(ctypes.c_void_p(a.ctypes.data))
        elif isinstance(a, (bool, np.bool_)):
            cargs.append(ctypes.c_int(int(a)))
        elif isinstance(a, (int, np.integer)):
            cargs.append(ctypes.c_int(int(a)))
        elif isinstance(a, (float, np.floating)):
            cargs.append(ctypes.c_double(float(a)))
        else:
            raise TypeError(f"{name}: unsupported argument {type(a)}")
    fn(*cargs)


def num_threads():
    f = lib().ora_num_threads
    f.restype = ctypes.c_int
    return int(f())


def set_num_threads(n):
    lib().ora_set_num_threads(ctypes.c_int(int(n)))


RELATION = {"evp": 1, "evp_revised": 2, "linear": 3}


# ------------------------------------------------------------------------------------ init
def seaice_init_mesh_boundary(domain):
    """interior_vertices of init_boundary (/root/reference/src/shared/mpas_seaice_mesh.F:372-488)."""
    for b in domain.blocklist:
        d = b.dimensions
        m = b.structs["mesh"]
        call("ora_interior_vertices", b.structs["boundary"]["interiorVertex"], d["nVerticesSolve"],
             d["vertexDegree"], d["nCells"], m["cellsOnVertex"])
    if halo_needed(domain):
        domain.halo(domain, "boundary", ["interiorVertex"])


def halo_needed(domain):
    return getattr(domain, "halo", None) is not None and domain.configs["config_use_halo_exch"]


def weak_scheme(cfg):
    """velocity_solver.F:169-187: both schemes 'variational' (default) or both 'weak'; the mixed ones
    (weak strain interpolated to the variational divergence, :2768-2784) are not restated."""
    pair = (cfg["config_strain_scheme"], cfg["config_stress_divergence_scheme"])
    if pair == ("weak", "weak"):
        return True
    assert pair in (("variational", "variational"), ("weak", "variational")), pair
    return False


def hybrid_scheme(cfg):
    """weak strain interpolated to the corners, variational stress and divergence (velocity_solver.F:2768-2784)"""
    return (cfg["config_strain_scheme"], cfg["config_stress_divergence_scheme"]) == ("weak", "variational")


def seaice_init_velocity_solver(domain):
    cfg = domain.configs
    st = domain.solver_state
    seaice_init_mesh_boundary(domain)
    from mpas_seaice_amd.special_boundaries import seaice_init_special_boundaries
    seaice_init_special_boundaries(domain)  # host-side index translation only (blocks have no device context)
    for b in domain.blocklist:
        vs = b.structs["velocity_solver"]
        vs["dynamicsTimeStep"][0] = cfg["config_dt"] / float(cfg["config_dynamics_subcycle_number"])
        vs["elasticTimeStep"][0] = vs["dynamicsTimeStep"][0] / float(cfg["config_elastic_subcycle_number"])
    # seaice_init_evp, constitutive_relation.F:75-164
    dvmin = 1.0e30
    for b in domain.blocklist:
        vs = b.structs["velocity_solver"]
        st["dampingTimescale"] = 0.36 * vs["dynamicsTimeStep"][0]
        dvmin = min(dvmin, float(b.structs["mesh"]["dvEdge"][:b.dimensions["nEdgesSolve"]].min()))
    if getattr(domain, "allreduce_min", None) is not None:
        dvmin = domain.allreduce_min(dvmin)
    for b in domain.blocklist:
        gamma = 0.25 * 1.0e11 * b.structs["velocity_solver"]["dynamicsTimeStep"][0]
        st["numericalInertiaCoefficient"] = (2.0 * 0.86 * 5.5e-3 * gamma) / dvmin ** 2
    for b in domain.blocklist:
        d = b.dimensions
        m = b.structs["mesh"]
        vv = b.structs["velocity_variational"]
        nC, nV, M, D = d["nCells"], d["nVertices"], d["maxEdges"], d["vertexDegree"]
        rot = cfg["config_rotate_cartesian_grid"]
        call("ora_metric_terms", vv["tanLatVertexRotatedOverRadius"], nV, m["xVertex"], m["yVertex"], m["zVertex"],
             float(m["sphere_radius"]), rot, bool(cfg["config_include_metric_terms"] and m["on_a_sphere"]))
        call("ora_cell_vertices_at_vertex", vv["cellVerticesAtVertex"], nV, D, M, nC, m["nEdgesOnCell"],
             m["verticesOnCell"], m["cellsOnVertex"])
        xl = np.zeros((M, nC), order="F")
        yl = np.zeros((M, nC), order="F")
        call("ora_calc_local_coords", xl, yl, nC, M, m["nEdgesOnCell"], m["verticesOnCell"], m["xVertex"],
             m["yVertex"], m["zVertex"], m["xCell"], m["yCell"], m["zCell"], rot, bool(m["on_a_sphere"]))
        if cfg["config_variational_basis"] == "pwl":  # variational.F:296-312
            call("ora_pwl_init", nC, M, m["nEdgesOnCell"], m["edgesOnCell"], m["dvEdge"], m["areaCell"], xl, yl,
                 vv["basisGradientU"], vv["basisGradientV"], vv["basisIntegralsMetric"], vv["basisIntegralsU"],
                 vv["basisIntegralsV"])
        else:
            assert cfg["config_variational_basis"] == "wachspress"
            qu, qv, qw, qn = integration_rule(cfg)
            call("ora_wachspress_init", nC, M, m["nEdgesOnCell"], xl, yl, vv["basisGradientU"], vv["basisGradientV"],
                 vv["basisIntegralsU"], vv["basisIntegralsV"], vv["basisIntegralsMetric"], qu.size, qu, qv, qw, qn)
        call("ora_variational_denominator", vv["variationalDenominator"], nV, D, M, m["nEdgesOnCell"],
             m["areaTriangle"], m["cellsOnVertex"], vv["cellVerticesAtVertex"], vv["basisIntegralsMetric"],
             cfg["config_variational_denominator_type"] == "alternate")
        b._xLocal, b._yLocal = xl, yl
        if weak_scheme(cfg) or hybrid_scheme(cfg):  # seaice_init_velocity_solver_weak, velocity_solver_weak.F:48-94
            vw = b.structs["velocity_weak"]
            iv = b.structs["boundary"]["interiorVertex"]
            geo = (m["xCell"], m["yCell"], m["zCell"], m["xEdge"], m["yEdge"], m["zEdge"], m["xVertex"], m["yVertex"],
                   m["zVertex"], float(m["sphere_radius"]), bool(m["on_a_sphere"]), rot, True)
            call("ora_normal_vectors_polygon", nC, M, m["nEdgesOnCell"], m["edgesOnCell"], m["verticesOnEdge"],
                 m["cellsOnEdge"], *geo, vw["normalVectorPolygon"], vw["latCellRotated"])
            call("ora_normal_vectors_triangle", nV, d["nVerticesSolve"], D, m["edgesOnVertex"], m["cellsOnEdge"],
                 m["verticesOnEdge"], iv, *geo, vw["normalVectorTriangle"], vw["latVertexRotated"])


# ------------------------------------------------------------------------------------ run
def _interp_c2v(b, vert, cell):
    d = b.dimensions
    m = b.structs["mesh"]
    call("ora_interpolate_cell_to_vertex", vert, cell, d["nVerticesSolve"], d["vertexDegree"], m["cellsOnVertex"],
         m["areaCell"])


def integration_rule(cfg):
    """get_integration_factors (wachspress.F:1223-1285): points, weights and normalisation on the unit triangle"""
    types = {"dunavant": 1, "trapezoidal": 2, "fekete": 3}
    t = cfg["config_wachspress_integration_type"]
    if t not in types:
        raise ValueError("Unknown Wachspress integration type for the oracle: " + str(t))
    u, v, w = np.zeros(256), np.zeros(256), np.zeros(256)
    norm = np.zeros(1)
    fn = lib().ora_integration_rule
    fn.restype = ctypes.c_int
    n = fn(ctypes.c_int(types[t]), ctypes.c_int(int(cfg["config_wachspress_integration_order"])),
           ctypes.c_void_p(u.ctypes.data), ctypes.c_void_p(v.ctypes.data), ctypes.c_void_p(w.ctypes.data),
           ctypes.c_void_p(norm.ctypes.data))
    if n == 0:
        raise ValueError("integration rule not restated: %s order %s" % (t, cfg["config_wachspress_integration_order"]))
    return u[:n].copy(), v[:n].copy(), w[:n].copy(), float(norm[0])


def column_strength_options(cfg):
    """kstrength, krdg_partic, krdg_redist, mu_rdg, Cf from the namelist strings (mpas_seaice_column.F:11432-11460)"""
    table = {"config_ice_strength_formulation": {"Hibler79": 0, "Rothrock75": 1},
             "config_ridging_participation_function": {"Thorndike75": 0, "exponential": 1},
             "config_ridging_redistribution_function": {"Hibler80": 0, "exponential": 1}}
    out = []
    for key, m in table.items():
        if cfg[key] not in m:
            raise ValueError(key + " unknown: " + str(cfg[key]))
        out.append(m[cfg[key]])
    return out + [float(cfg["config_ridiging_efolding_scale"]), float(cfg["config_ratio_ridging_work_to_PE"])]


def velocity_solver_pre_subcycle(domain):
    cfg = domain.configs
    no_column = (not cfg["config_use_column_package"]) or (
        cfg["config_use_column_package"] and not cfg["config_use_column_vertical_thermodynamics"])
    ex = halo_needed(domain)
    for b in domain.blocklist:
        d = b.dimensions
        tr, ta, ic = b.structs["tracers"], b.structs["tracers_aggregate"], b.structs["icestate"]
        call("ora_aggregate_mass_and_area", d["nCells"], d["nCategories"], tr["iceAreaCategory"],
             tr["iceVolumeCategory"], tr["snowVolumeCategory"], ta["iceAreaCell"], ta["iceVolumeCell"],
             ta["snowVolumeCell"], ic["totalMassCell"])
        if no_column:
            ic["iceAreaCellInitial"][:] = ta["iceAreaCell"]
    if ex:
        domain.halo(domain, "icestate", ["iceAreaCellInitial", "totalMassCell"])
    for b in domain.blocklist:
        ic = b.structs["icestate"]
        _interp_c2v(b, ic["iceAreaVertex"], ic["iceAreaCellInitial"])
        _interp_c2v(b, ic["totalMassVertex"], ic["totalMassCell"])
    if cfg["config_calc_velocity_masks"]:
        for b in domain.blocklist:
            d = b.dimensions
            m, vs, ic, oc = b.structs["mesh"], b.structs["velocity_solver"], b.structs["icestate"], b.structs["ocean_coupling"]
            call("ora_stress_calculation_mask", vs["solveStress"], d["nCells"], d["maxEdges"], m["nEdgesOnCell"],
                 m["cellsOnCell"], ic["iceAreaCellInitial"], ic["totalMassCell"], oc["landIceMask"])
            call("ora_velocity_calculation_mask", vs["solveVelocity"], d["nVerticesSolve"], d["nVertices"],
                 b.structs["boundary"]["interiorVertex"], oc["landIceMaskVertex"], ic["iceAreaVertex"],
                 ic["totalMassVertex"])
    if ex:
        domain.halo(domain, "velocity_solver", ["solveVelocity"])
    for b in domain.blocklist:
        d = b.dimensions
        vs, oc = b.structs["velocity_solver"], b.structs["ocean_coupling"]
        _interp_c2v(b, vs["uOceanVelocityVertex"], oc["uOceanVelocity"])
        _interp_c2v(b, vs["vOceanVelocityVertex"], oc["vOceanVelocity"])
        call("ora_new_ice_velocities", d["nVerticesSolve"], d["nVertices"] + 1, vs["solveVelocity"],
             vs["solveVelocityPrevious"], vs["uVelocity"], vs["vVelocity"], vs["uVelocityInitial"],
             vs["vVelocityInitial"], vs["uOceanVelocityVertex"], vs["vOceanVelocityVertex"],
             vs["stressDivergenceU"], vs["stressDivergenceV"], vs["oceanStressU"], vs["oceanStressV"])
    if ex:
        domain.halo(domain, "velocity_solver", ["uVelocity", "vVelocity"])
    for b in domain.blocklist:
        d = b.dimensions
        vs, ta = b.structs["velocity_solver"], b.structs["tracers_aggregate"]
        if no_column:
            call("ora_ice_strength_hibler", d["nCellsSolve"], vs["solveStress"], ta["iceAreaCell"],
                 ta["iceVolumeCell"], vs["icePressure"])
        else:  # colpkg_ice_strength, velocity_solver.F:1437-1458
            k = column_strength_options(cfg)
            tr = b.structs["tracers"]
            call("ora_ice_strength_column", d["nCellsSolve"], d["nCategories"], vs["solveStress"], ta["iceAreaCell"],
                 ta["iceVolumeCell"], b.structs["icestate"]["openWaterArea"], tr["iceAreaCategory"],
                 tr["iceVolumeCategory"], k[0], k[1], k[2], k[3], k[4], vs["icePressure"])
    if ex:
        domain.halo(domain, "velocity_solver", ["icePressure"])
    for b in domain.blocklist:
        d = b.dimensions
        vs, at, ta = b.structs["velocity_solver"], b.structs["atmos_coupling"], b.structs["tracers_aggregate"]
        if no_column:
            call("ora_constant_air_stress", d["nCellsSolve"], at["uAirVelocity"], at["vAirVelocity"], at["airDensity"],
                 ta["iceAreaCell"], vs["airStressCellU"], vs["airStressCellV"])
        if not cfg["config_use_air_stress"]:
            vs["airStressCellU"][:] = 0.0
            vs["airStressCellV"][:] = 0.0
    if ex:
        domain.halo(domain, "velocity_solver", ["airStressCellU", "airStressCellV"])
    for b in domain.blocklist:
        d = b.dimensions
        m, vs, ic, oc = b.structs["mesh"], b.structs["velocity_solver"], b.structs["icestate"], b.structs["ocean_coupling"]
        _interp_c2v(b, vs["airStressVertexU"], vs["airStressCellU"])
        _interp_c2v(b, vs["airStressVertexV"], vs["airStressCellV"])
        call("ora_coriolis_force_coefficient", d["nVerticesSolve"], ic["totalMassVertex"], m["fVertex"],
             vs["totalMassVertexfVertex"])
        call("ora_ocean_stress", d["nVerticesSolve"], d["nVertices"] + 1, cfg["config_use_ocean_stress"],
             vs["solveVelocity"], vs["uOceanVelocityVertex"], vs["vOceanVelocityVertex"], m["fVertex"],
             vs["oceanStressU"], vs["oceanStressV"])
    mode = 0
    if cfg["config_use_surface_tilt"]:
        mode = 1 if cfg["config_geostrophic_surface_tilt"] else 2
    if mode == 2 and ex:
        domain.halo(domain, "ocean_coupling", ["seaSurfaceTiltU", "seaSurfaceTiltV"])
    for b in domain.blocklist:
        d = b.dimensions
        m, vs, ic, oc = b.structs["mesh"], b.structs["velocity_solver"], b.structs["icestate"], b.structs["ocean_coupling"]
        if mode == 2:
            _interp_c2v(b, vs["seaSurfaceTiltVertexU"], oc["seaSurfaceTiltU"])
            _interp_c2v(b, vs["seaSurfaceTiltVertexV"], oc["seaSurfaceTiltV"])
        call("ora_surface_tilt", d["nVerticesSolve"], d["nVertices"] + 1, mode, vs["solveVelocity"],
             ic["totalMassVertex"], m["fVertex"], vs["uOceanVelocityVertex"], vs["vOceanVelocityVertex"],
             vs["seaSurfaceTiltVertexU"], vs["seaSurfaceTiltVertexV"], vs["surfaceTiltForceU"],
             vs["surfaceTiltForceV"])
        vv = b.structs["velocity_variational"]
        if weak_scheme(cfg):  # velocity_solver.F:2347-2380: the same with one value per cell
            vw = b.structs["velocity_weak"]
            call("ora_init_subcycle_variables", d["nCells"] + 1, d["nVerticesSolve"], d["nVertices"] + 1, 1,
                 _pad_mask(vs["solveStress"]), vs["solveVelocity"], vs["stressDivergenceU"], vs["stressDivergenceV"],
                 vs["uVelocity"], vs["vVelocity"], vs["oceanStressCoeff"], vw["strain11"], vw["strain22"],
                 vw["strain12"], vw["stress11"], vw["stress22"], vw["stress12"])
            continue
        call("ora_init_subcycle_variables", d["nCells"] + 1, d["nVerticesSolve"], d["nVertices"] + 1, d["maxEdges"],
             _pad_mask(vs["solveStress"]), vs["solveVelocity"], vs["stressDivergenceU"], vs["stressDivergenceV"],
             vs["uVelocity"], vs["vVelocity"], vs["oceanStressCoeff"], vv["strain11"], vv["strain22"],
             vv["strain12"], vv["stress11"], vv["stress22"], vv["stress12"])


def _pad_mask(a):
    return a


def _subcycle_args_weak(domain, b):
    cfg, st = domain.configs, domain.solver_state
    d = b.dimensions
    m, vs, vw, ic = b.structs["mesh"], b.structs["velocity_solver"], b.structs["velocity_weak"], b.structs["icestate"]
    return (d["nCells"], d["nVerticesSolve"], d["nVertices"], d["maxEdges"], d["vertexDegree"],
            m["nEdgesOnCell"], m["verticesOnCell"], m["edgesOnCell"], m["verticesOnEdge"], m["cellsOnVertex"],
            m["edgesOnVertex"], m["cellsOnEdge"], m["dvEdge"], m["dcEdge"], m["areaCell"], m["areaTriangle"],
            float(m["sphere_radius"]) if m["on_a_sphere"] else 0.0, vs["solveStress"], vs["solveVelocity"],
            vw["normalVectorPolygon"], vw["normalVectorTriangle"], vw["latCellRotated"], vw["latVertexRotated"],
            vs["icePressure"], ic["totalMassVertex"], vs["totalMassVertexfVertex"], ic["iceAreaVertex"],
            vs["airStressVertexU"], vs["airStressVertexV"], vs["surfaceTiltForceU"], vs["surfaceTiltForceV"],
            vs["oceanStressU"], vs["oceanStressV"], vs["uOceanVelocityVertex"], vs["vOceanVelocityVertex"],
            vs["uVelocityInitial"], vs["vVelocityInitial"], vs["uVelocity"], vs["vVelocity"], vw["strain11"],
            vw["strain22"], vw["strain12"], vw["stress11"], vw["stress22"], vw["stress12"], vw["replacementPressure"],
            vs["stressDivergenceU"], vs["stressDivergenceV"], vs["oceanStressCoeff"], float(vs["elasticTimeStep"][0]),
            float(vs["dynamicsTimeStep"][0]), float(st["dampingTimescale"]),
            float(st["numericalInertiaCoefficient"]), RELATION[cfg["config_constitutive_relation_type"]],
            bool(cfg["config_use_ocean_stress"]), cfg["config_ocean_stress_type"] == "quadratic")


def _subcycle_args_hybrid(domain, b):
    cfg, st = domain.configs, domain.solver_state
    d = b.dimensions
    m, vs, vw, ic = b.structs["mesh"], b.structs["velocity_solver"], b.structs["velocity_weak"], b.structs["icestate"]
    vv, wv = b.structs["velocity_variational"], b.structs["velocity_weak_variational"]
    return (d["nCells"], d["nVerticesSolve"], d["nVertices"] + 1, d["maxEdges"], d["vertexDegree"],
            m["nEdgesOnCell"], m["verticesOnCell"], m["edgesOnCell"], m["verticesOnEdge"], m["cellsOnVertex"],
            vv["cellVerticesAtVertex"], m["dvEdge"], m["areaCell"],
            float(m["sphere_radius"]) if m["on_a_sphere"] else 0.0, vs["solveStress"], vs["solveVelocity"],
            vw["normalVectorPolygon"], vw["latCellRotated"], vv["basisIntegralsU"], vv["basisIntegralsV"],
            vv["basisIntegralsMetric"], vv["variationalDenominator"], vv["tanLatVertexRotatedOverRadius"],
            vs["icePressure"], ic["totalMassVertex"], vs["totalMassVertexfVertex"], ic["iceAreaVertex"],
            vs["airStressVertexU"], vs["airStressVertexV"], vs["surfaceTiltForceU"], vs["surfaceTiltForceV"],
            vs["oceanStressU"], vs["oceanStressV"], vs["uOceanVelocityVertex"], vs["vOceanVelocityVertex"],
            vs["uVelocityInitial"], vs["vVelocityInitial"], vs["uVelocity"], vs["vVelocity"], vw["strain11"],
            vw["strain22"], vw["strain12"], wv["strain11Vertex"], wv["strain22Vertex"], wv["strain12Vertex"],
            vv["strain11"], vv["strain22"], vv["strain12"], vv["stress11"], vv["stress22"], vv["stress12"],
            vv["replacementPressure"], vs["stressDivergenceU"], vs["stressDivergenceV"], vs["oceanStressCoeff"],
            float(vs["elasticTimeStep"][0]), float(vs["dynamicsTimeStep"][0]), float(st["dampingTimescale"]),
            float(st["numericalInertiaCoefficient"]), RELATION[cfg["config_constitutive_relation_type"]],
            bool(cfg["config_use_ocean_stress"]), cfg["config_ocean_stress_type"] == "quadratic")


def _subcycle_call(domain, b, n):
    if hybrid_scheme(domain.configs):
        call("ora_evp_subcycle_loop_hybrid", n, *_subcycle_args_hybrid(domain, b))
    elif weak_scheme(domain.configs):
        call("ora_evp_subcycle_loop_weak", n, *_subcycle_args_weak(domain, b))
    else:
        call("ora_evp_subcycle_loop", n, *_subcycle_args(domain, b))


def _subcycle_args(domain, b):
    cfg, st = domain.configs, domain.solver_state
    d = b.dimensions
    m, vs, vv, ic = b.structs["mesh"], b.structs["velocity_solver"], b.structs["velocity_variational"], b.structs["icestate"]
    return (d["nCells"], d["nVerticesSolve"], d["nVertices"] + 1, d["maxEdges"], d["vertexDegree"],
            m["nEdgesOnCell"], m["verticesOnCell"], m["cellsOnVertex"], vv["cellVerticesAtVertex"],
            vs["solveStress"], vs["solveVelocity"], vv["basisGradientU"], vv["basisGradientV"],
            vv["basisIntegralsU"], vv["basisIntegralsV"], vv["basisIntegralsMetric"], vv["variationalDenominator"],
            vv["tanLatVertexRotatedOverRadius"], vs["icePressure"], ic["totalMassVertex"],
            vs["totalMassVertexfVertex"], ic["iceAreaVertex"], vs["airStressVertexU"], vs["airStressVertexV"],
            vs["surfaceTiltForceU"], vs["surfaceTiltForceV"], vs["oceanStressU"], vs["oceanStressV"],
            vs["uOceanVelocityVertex"], vs["vOceanVelocityVertex"], vs["uVelocityInitial"], vs["vVelocityInitial"],
            vs["uVelocity"], vs["vVelocity"], vv["strain11"], vv["strain22"], vv["strain12"], vv["stress11"],
            vv["stress22"], vv["stress12"], vv["replacementPressure"], vs["stressDivergenceU"],
            vs["stressDivergenceV"], vs["oceanStressCoeff"], float(vs["elasticTimeStep"][0]),
            float(vs["dynamicsTimeStep"][0]), float(st["dampingTimescale"]),
            float(st["numericalInertiaCoefficient"]), RELATION[cfg["config_constitutive_relation_type"]],
            bool(cfg["config_use_ocean_stress"]), cfg["config_ocean_stress_type"] == "quadratic",
            bool(cfg["config_average_variational_strain"]), m["areaCell"])


def _set_special_boundaries(domain):
    """seaice_set_special_boundaries_velocity / _velocity_masks, special_boundaries.F:253-401"""
    cfg = domain.configs
    for b in domain.blocklist:
        sb, vs = b.structs["special_boundaries"], b.structs["velocity_solver"]
        nV, nC = b.dimensions["nVertices"], b.dimensions["nCells"]
        if cfg["config_use_special_boundaries_velocity"]:
            call("ora_set_special_boundaries_velocity", nV, sb["vertexBoundaryType"], sb["vertexBoundarySourceLocal"],
                 vs["uVelocity"], vs["vVelocity"])
        if cfg["config_use_special_boundaries_velocity_masks"]:
            vs["solveVelocity"][:nV] = sb["solveVelocitySpecialBoundaries"][:nV]
            vs["solveStress"][:nC] = sb["solveStressSpecialBoundaries"][:nC]


def subcycle_velocity_solver(domain, n_subcycles=None):
    cfg = domain.configs
    n = cfg["config_elastic_subcycle_number"] if n_subcycles is None else n_subcycles
    if cfg["config_use_special_boundaries_velocity"] or cfg["config_use_special_boundaries_velocity_masks"]:
        # velocity_solver.F:2441-2458: before the loop and after every iteration (after its halo update)
        _set_special_boundaries(domain)
        for _ in range(n):
            for b in domain.blocklist:
                _subcycle_call(domain, b, 1)
            if halo_needed(domain):
                domain.halo(domain, "velocity_solver", ["uVelocity", "vVelocity"])
            _set_special_boundaries(domain)
        return
    if not halo_needed(domain):
        for b in domain.blocklist:
            _subcycle_call(domain, b, n)
        return
    for _ in range(n):
        for b in domain.blocklist:
            _subcycle_call(domain, b, 1)
        domain.halo(domain, "velocity_solver", ["uVelocity", "vVelocity"])


def velocity_solver_post_subcycle(domain):
    cfg = domain.configs
    for b in domain.blocklist:
        d = b.dimensions
        m, vs, vv, ic = b.structs["mesh"], b.structs["velocity_solver"], b.structs["velocity_variational"], b.structs["icestate"]
        rg = b.structs["ridging"]
        if weak_scheme(cfg):  # velocity_solver.F:3416-3418, 3497-3516
            vw = b.structs["velocity_weak"]
            call("ora_final_divergence_shear_weak", d["nCellsSolve"], vw["strain11"], vw["strain22"], vw["strain12"],
                 vs["divergence"], vs["shear"], bool(cfg["config_use_column_package"]), rg["ridgeConvergence"],
                 rg["ridgeShear"])
            call("ora_principal_stresses_weak", d["nCellsSolve"], vw["stress11"], vw["stress22"], vw["stress12"],
                 vw["replacementPressure"], vw["principalStress1"], vw["principalStress2"])
        else:
            call("ora_final_divergence_shear", d["nCells"], d["maxEdges"], m["nEdgesOnCell"], vs["solveStress"],
                 vv["strain11"], vv["strain22"], vv["strain12"], vs["divergence"], vs["shear"],
                 bool(cfg["config_use_column_package"]), rg["ridgeConvergence"], rg["ridgeShear"])
            call("ora_principal_stresses", d["nCellsSolve"], d["maxEdges"], m["nEdgesOnCell"], vv["stress11"],
                 vv["stress22"], vv["stress12"], vv["replacementPressure"], vv["principalStress1"],
                 vv["principalStress2"])
        call("ora_ocean_stress_coefficient", d["nVerticesSolve"], d["nVertices"] + 1, cfg["config_use_ocean_stress"],
             cfg["config_ocean_stress_type"] == "quadratic", vs["solveVelocity"], vs["oceanStressCoeff"],
             vs["uOceanVelocityVertex"], vs["vOceanVelocityVertex"], vs["uVelocity"], vs["vVelocity"],
             ic["iceAreaVertex"])
        call("ora_ocean_stress_final_vertex", d["nVerticesSolve"], d["nVertices"] + 1, cfg["config_use_ocean_stress"],
             vs["solveVelocity"], vs["oceanStressCoeff"], vs["uOceanVelocityVertex"], vs["vOceanVelocityVertex"],
             vs["uVelocity"], vs["vVelocity"], m["fVertex"], ic["iceAreaVertex"], vs["oceanStressU"],
             vs["oceanStressV"])
    if cfg["config_use_ocean_stress"]:
        if halo_needed(domain):
            domain.halo(domain, "velocity_solver", ["oceanStressU", "oceanStressV"])
        for b in domain.blocklist:
            d = b.dimensions
            m, vs, ic = b.structs["mesh"], b.structs["velocity_solver"], b.structs["icestate"]
            iv = b.structs["boundary"]["interiorVertex"]
            for cellf, vertf in (("oceanStressCellU", "oceanStressU"), ("oceanStressCellV", "oceanStressV")):
                call("ora_interpolate_vertex_to_cell", vs[cellf], vs[vertf], d["nCellsSolve"], d["maxEdges"],
                     m["nEdgesOnCell"], m["verticesOnCell"], m["areaTriangle"], iv)
            call("ora_ocean_stress_final_rescale", d["nVerticesSolve"], vs["solveVelocity"], ic["iceAreaVertex"],
                 vs["oceanStressU"], vs["oceanStressV"])
    else:
        for b in domain.blocklist:
            b.structs["velocity_solver"]["oceanStressCellU"][:] = 0.0
            b.structs["velocity_solver"]["oceanStressCellV"][:] = 0.0


def seaice_run_velocity_solver(domain, clock=None):
    if domain.configs["config_use_velocity_solver"]:
        velocity_solver_pre_subcycle(domain)
        subcycle_velocity_solver(domain)
        velocity_solver_post_subcycle(domain)

"""Tracer hierarchy of the incremental-remap advection: host mirror of
/root/reference/src/shared/mpas_seaice_advection_incremental_remap_tracers.F.

The reference keeps a linked list of ``tracer_type`` (:30-137) headed by the mass-like field
``iceAreaCategory``; every other entry names its parent, and the list is ordered so that parents
come before children (``set_tracer_order`` :860-922: level 0, then 1, 2, 3, each level in the order
the tracers were added).  Here the list is a Python list of :class:`Tracer`.
"""
from __future__ import annotations

# tracers that are always advected (:218-224) followed by the optional ones in the order the
# reference adds them (:298-475, biogeochemistry included); each entry: (tracerName, parentName, package that activates it)
TRACER_TABLE = [
    ("iceAreaCategory", "None", None),
    ("iceVolumeCategory", "iceAreaCategory", None),
    ("snowVolumeCategory", "iceAreaCategory", None),
    ("surfaceTemperature", "iceAreaCategory", None),
    ("iceEnthalpy", "iceVolumeCategory", None),
    ("iceSalinity", "iceVolumeCategory", None),
    ("snowEnthalpy", "snowVolumeCategory", None),
    ("iceAge", "iceVolumeCategory", "pkgColumnTracerIceAge"),
    ("firstYearIceArea", "iceAreaCategory", "pkgColumnTracerFirstYearIce"),
    ("levelIceArea", "iceAreaCategory", "pkgColumnTracerLevelIce"),
    ("levelIceVolume", "iceVolumeCategory", "pkgColumnTracerLevelIce"),
    ("pondArea", "levelIceArea|iceAreaCategory", "pkgColumnTracerPonds"),
    ("pondDepth", "pondArea", "pkgColumnTracerPonds"),
    ("pondLidThickness", "pondArea", "pkgColumnTracerLidThickness"),
    ("snowIceMass", "snowVolumeCategory", "pkgColumnTracerEffectiveSnowDensity"),
    ("snowLiquidMass", "snowVolumeCategory", "pkgColumnTracerEffectiveSnowDensity"),
    ("snowDensity", "snowVolumeCategory", "pkgColumnTracerEffectiveSnowDensity"),
    ("snowGrainRadius", "snowVolumeCategory", "pkgColumnTracerSnowGrainRadius"),
    ("snowScatteringAerosol", "snowVolumeCategory", "pkgColumnTracerAerosols"),
    ("snowBodyAerosol", "snowVolumeCategory", "pkgColumnTracerAerosols"),
    ("iceScatteringAerosol", "iceVolumeCategory", "pkgColumnTracerAerosols"),
    ("iceBodyAerosol", "iceVolumeCategory", "pkgColumnTracerAerosols"),
    ("brineFraction", "iceVolumeCategory", "pkgTracerBrine"),
    # skeletal-layer and vertical biogeochemistry, vertical aerosols and salinity (:347-475)
    ("skeletalAlgaeConc", "iceAreaCategory", "pkgTracerSkeletalAlgae"),
    ("skeletalNitrateConc", "iceAreaCategory", "pkgTracerSkeletalNitrate"),
    ("skeletalDOCConc", "iceAreaCategory", "pkgTracerSkeletalCarbon"),
    ("skeletalDICConc", "iceAreaCategory", "pkgTracerSkeletalCarbon"),
    ("skeletalDONConc", "iceAreaCategory", "pkgTracerSkeletalDON"),
    ("skeletalAmmoniumConc", "iceAreaCategory", "pkgTracerSkeletalAmmonium"),
    ("skeletalSilicateConc", "iceAreaCategory", "pkgTracerSkeletalSilicate"),
    ("skeletalDMSConc", "iceAreaCategory", "pkgTracerSkeletalDMS"),
    ("skeletalDMSPpConc", "iceAreaCategory", "pkgTracerSkeletalDMS"),
    ("skeletalDMSPdConc", "iceAreaCategory", "pkgTracerSkeletalDMS"),
    ("skeletalNonreactiveConc", "iceAreaCategory", "pkgTracerSkeletalNonreactive"),
    ("skeletalHumicsConc", "iceAreaCategory", "pkgTracerSkeletalHumics"),
    ("skeletalParticulateIronConc", "iceAreaCategory", "pkgTracerSkeletalIron"),
    ("skeletalDissolvedIronConc", "iceAreaCategory", "pkgTracerSkeletalIron"),
    ("mobileFraction", "brineFraction", "pkgTracerMobileFraction"),
    ("verticalAlgaeSnow", "snowVolumeCategory", "pkgTracerVerticalAlgae"),
    ("verticalAlgaeIce", "brineFraction", "pkgTracerVerticalAlgae"),
    ("verticalNitrateSnow", "snowVolumeCategory", "pkgTracerVerticalNitrate"),
    ("verticalNitrateIce", "brineFraction", "pkgTracerVerticalNitrate"),
    ("verticalDOCSnow", "snowVolumeCategory", "pkgTracerVerticalCarbon"),
    ("verticalDICSnow", "snowVolumeCategory", "pkgTracerVerticalCarbon"),
    ("verticalDOCIce", "brineFraction", "pkgTracerVerticalCarbon"),
    ("verticalDICIce", "brineFraction", "pkgTracerVerticalCarbon"),
    ("verticalDONSnow", "snowVolumeCategory", "pkgTracerVerticalDON"),
    ("verticalDONIce", "brineFraction", "pkgTracerVerticalDON"),
    ("verticalAmmoniumSnow", "snowVolumeCategory", "pkgTracerVerticalAmmonium"),
    ("verticalAmmoniumIce", "brineFraction", "pkgTracerVerticalAmmonium"),
    ("verticalSilicateSnow", "snowVolumeCategory", "pkgTracerVerticalSilicate"),
    ("verticalSilicateIce", "brineFraction", "pkgTracerVerticalSilicate"),
    ("verticalDMSSnow", "snowVolumeCategory", "pkgTracerVerticalDMS"),
    ("verticalDMSPpSnow", "snowVolumeCategory", "pkgTracerVerticalDMS"),
    ("verticalDMSPdSnow", "snowVolumeCategory", "pkgTracerVerticalDMS"),
    ("verticalDMSIce", "brineFraction", "pkgTracerVerticalDMS"),
    ("verticalDMSPpIce", "brineFraction", "pkgTracerVerticalDMS"),
    ("verticalDMSPdIce", "brineFraction", "pkgTracerVerticalDMS"),
    ("verticalNonreactiveSnow", "snowVolumeCategory", "pkgTracerVerticalNonreactive"),
    ("verticalNonreactiveIce", "brineFraction", "pkgTracerVerticalNonreactive"),
    ("verticalHumicsSnow", "snowVolumeCategory", "pkgTracerVerticalHumics"),
    ("verticalHumicsIce", "brineFraction", "pkgTracerVerticalHumics"),
    ("verticalParticulateIronSnow", "snowVolumeCategory", "pkgTracerVerticalIron"),
    ("verticalDissolvedIronSnow", "snowVolumeCategory", "pkgTracerVerticalIron"),
    ("verticalParticulateIronIce", "brineFraction", "pkgTracerVerticalIron"),
    ("verticalDissolvedIronIce", "brineFraction", "pkgTracerVerticalIron"),
    ("verticalAerosolsIce", "brineFraction", "pkgTracerZAerosols"),
    ("verticalAerosolsSnow", "snowVolumeCategory", "pkgTracerZAerosols"),
    ("verticalSalinity", "brineFraction", "pkgTracerZSalinity"),
]


class Tracer:
    """One element of the list (tracer_type, :30-137): the fields the host needs."""

    def __init__(self, name, parentName):
        self.tracerName = name
        self.parentName = parentName
        self.parent = None
        self.hasChild = False
        self.nParents = 0
        self.ndims = 2      # 2: (nCategories, nCells); 3: (nLayers, nCategories, nCells)
        self.nLayers = 1

    def __repr__(self):
        return f"Tracer({self.tracerName}, parent={self.parentName}, nParents={self.nParents}, L={self.nLayers})"


def seaice_add_tracers_to_linked_list(domain, extra=()):
    """Ordered tracer list for a domain (:158-526).

    A tracer of the table is active when the ``tracers`` pool of the first block holds a field of that
    name (the reference asks the pool whether the field's package is active, :752-764).  ``extra`` adds
    (tracerName, parentName) pairs outside the table.  ``config_use_level_meltponds`` chooses the
    parent of pondArea (:311-316).
    """
    pool = domain.blocklist[0].structs["tracers"]
    level_ponds = bool(domain.configs.get("config_use_level_meltponds", False))
    initial = []
    for name, parent, _pkg in TRACER_TABLE:
        if "|" in parent:
            a, b = parent.split("|")
            parent = a if level_ponds else b
        initial.append(Tracer(name, parent))
    for name, parent in extra:
        initial.append(Tracer(name, parent))
    # create_tracers_connectivity :601-637
    byname = {t.tracerName: t for t in initial}
    for t in initial:
        p = byname.get(t.parentName)
        if p is not None:
            t.parent = p
            p.hasChild = True
    # set_parent_numbers :650-679
    for t in initial:
        n, p = 0, t.parent
        while p is not None:
            n += 1
            p = p.parent
        t.nParents = n
    # remove_inactive_tracers :733-846
    active = []
    for t in initial:
        if t.tracerName in pool:
            active.append(t)
    for t in active:
        if t.parent is not None and t.parent not in active:
            raise RuntimeError("IR tracers: Cannot have an inactive parent with an active child: "
                               f"{t.parent.tracerName} / {t.tracerName}")
    if not active:
        raise RuntimeError("IR tracers: No active tracers for IR")
    # hasChild as seen by the run: the reference keeps the flag set at connectivity time even if the
    # child was later removed (:625), so it is left untouched here.
    # set_tracer_order :860-922
    maxLevels = max(t.nParents for t in active)
    ordered = []
    for level in range(maxLevels + 1):
        ordered += [t for t in active if t.nParents == level]
    if ordered[0].nParents != 0 or sum(1 for t in ordered if t.nParents == 0) != 1:
        raise RuntimeError("IR tracers: exactly one mass-like field (no parent) is required")
    for t in ordered:
        if t.nParents > 3:
            raise RuntimeError("Tracer has > 3 parents: " + t.tracerName)
    # seaice_set_tracer_array_pointers :936-1118: a leading dimension of size 1 makes the tracer "2D"
    for t in ordered:
        a = pool[t.tracerName]
        t.nLayers = int(a.shape[0]) if a.ndim == 3 else 1
        t.ndims = 2 if t.nLayers == 1 else 3
    if ordered[0].ndims != 2:
        raise RuntimeError("IR: mass-like field must be 2D")
    for t in ordered:
        if t.parent is not None and t.parent.nLayers not in (1, t.nLayers):
            raise RuntimeError(f"IR tracers: {t.tracerName} and its parent have different layer counts")
    domain.solver_state["tracersHead"] = ordered
    return ordered

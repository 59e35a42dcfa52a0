// Time-step kernels of the EVP velocity solver: pre-subcycle, the subcycle pair (fused cell kernel +
// fused vertex kernel) and post-subcycle.
//
// Device data are structure-of-arrays: corner quantity (s, cell) at [s * padC + cell], so one thread
// per cell (or vertex) gives fully coalesced 8-byte-per-lane streams for every array; the only
// gathers are the vertex values a cell reads through verticesOnCell and the per-corner partial sums
// a vertex reads through its three (cell, corner) pairs.
//
// Arithmetic follows the cited reference statements in the same order; the file is compiled with
// -ffp-contract=off (the kernels are bandwidth-bound, fused multiply-add buys nothing here).
#include <cmath>

#include "sih_internal.h"

SihField *sih_find_field(sih_ctx *c, const char *name);

// velocity_solver.F:61-65; constitutive_relation.F:41-47; column/constants/cice/ice_constants_colpkg.F90
#define SIN_TURN 0.0
#define COS_TURN 1.0
#define AREA_MIN 0.001
#define MASS_MIN 0.01
#define ECC2 4.0
#define PUNY 1.0e-11
#define RHOI 917.0
#define RHOS 330.0
#define RHOW 1026.0
#define DRAGIO 0.00536
#define PSTAR 2.75e4
#define CSTAR 20.0
#define GRAVIT 9.80616

void sih_bind_dev(sih_ctx *c)
{
    auto fi = [&](const char *n) { return (int *)sih_find_field(c, n)->d; };
    auto fd = [&](const char *n) { return (double *)sih_find_field(c, n)->d; };
    SihDev &d = c->dev;
    d.nCells = c->nCells;
    d.nCellsSolve = c->nCellsSolve;
    d.nVertices = c->nVertices;
    d.nVerticesSolve = c->nVerticesSolve;
    d.M = c->M;
    d.D = c->D;
    d.nCat = c->cfg.nCategories;
    d.padC = c->padC;
    d.padV = c->padV;
    d.nEdgesOnCell = fi("_nEdgesOnCell");
    d.voc = fi("_voc");
    d.coc = fi("_coc");
    d.cov = fi("_cov");
    d.cvav = fi("cellVerticesAtVertex");
    d.gidx = fi("_gidx");
    d.interiorVertex = fi("interiorVertex");
    d.vActive = fi("_vertexActive");
    d.landIceMask = fi("landIceMask");
    d.landIceMaskVertex = fi("landIceMaskVertex");
    d.solveStress = fi("solveStress");
    d.solveVelocity = fi("solveVelocity");
    d.solveVelocityPrevious = fi("solveVelocityPrevious");
    d.areaCell = fd("areaCell");
    d.areaTriangle = fd("areaTriangle");
    d.fVertex = fd("fVertex");
    d.tanLat = fd("tanLatVertexRotatedOverRadius");
    d.varDenom = fd("variationalDenominator");
    d.gradU = fd("basisGradientU");
    d.gradV = fd("basisGradientV");
    d.gradU3 = fd("_gradU3");
    d.gradV3 = fd("_gradV3");
    d.nEdgesDerive = (const int *)sih_find_field(c, "_nEdgesDerive")->d;
    d.airTiltU = fd("_airTiltU");
    d.airTiltV = fd("_airTiltV");
    d.SU = fd("basisIntegralsU");
    d.SV = fd("basisIntegralsV");
    d.SM = fd("basisIntegralsMetric");
    d.u = fd("uVelocity");
    d.v = fd("vVelocity");
    d.uInit = fd("uVelocityInitial");
    d.vInit = fd("vVelocityInitial");
    d.strain11 = fd("strain11");
    d.strain22 = fd("strain22");
    d.strain12 = fd("strain12");
    d.stress11 = fd("stress11");
    d.stress22 = fd("stress22");
    d.stress12 = fd("stress12");
    d.Prep = fd("replacementPressure");
    d.prin1 = fd("principalStress1");
    d.prin2 = fd("principalStress2");
    d.partU = fd("_partU");
    d.partV = fd("_partV");
    d.divU = fd("stressDivergenceU");
    d.divV = fd("stressDivergenceV");
    d.icePressure = fd("icePressure");
    d.airCellU = fd("airStressCellU");
    d.airCellV = fd("airStressCellV");
    d.airVertU = fd("airStressVertexU");
    d.airVertV = fd("airStressVertexV");
    d.uOcnVert = fd("uOceanVelocityVertex");
    d.vOcnVert = fd("vOceanVelocityVertex");
    d.ocnStressU = fd("oceanStressU");
    d.ocnStressV = fd("oceanStressV");
    d.ocnCoeff = fd("oceanStressCoeff");
    d.ocnStressCellU = fd("oceanStressCellU");
    d.ocnStressCellV = fd("oceanStressCellV");
    d.tiltVertU = fd("seaSurfaceTiltVertexU");
    d.tiltVertV = fd("seaSurfaceTiltVertexV");
    d.tiltForceU = fd("surfaceTiltForceU");
    d.tiltForceV = fd("surfaceTiltForceV");
    d.massF = fd("totalMassVertexfVertex");
    d.divergence = fd("divergence");
    d.shear = fd("shear");
    d.ridgeConv = fd("ridgeConvergence");
    d.ridgeShear = fd("ridgeShear");
    d.iceAreaVertex = fd("iceAreaVertex");
    d.totalMassVertex = fd("totalMassVertex");
    d.iceAreaCellInitial = fd("iceAreaCellInitial");
    d.totalMassCell = fd("totalMassCell");
    d.iceAreaCell = fd("iceAreaCell");
    d.iceVolumeCell = fd("iceVolumeCell");
    d.snowVolumeCell = fd("snowVolumeCell");
    d.iceAreaCat = fd("iceAreaCategory");
    d.iceVolCat = fd("iceVolumeCategory");
    d.snowVolCat = fd("snowVolumeCategory");
    d.uOcn = fd("uOceanVelocity");
    d.vOcn = fd("vOceanVelocity");
    d.tiltU = fd("seaSurfaceTiltU");
    d.tiltV = fd("seaSurfaceTiltV");
    d.uAir = fd("uAirVelocity");
    d.vAir = fd("vAirVelocity");
    d.airDensity = fd("airDensity");
    d.relation = c->cfg.constitutiveRelation;
    d.useOcean = c->cfg.useOceanStress;
    d.quadratic = c->cfg.oceanStressType == 1;
    d.useAir = c->cfg.useAirStress;
    d.tiltMode = c->cfg.surfaceTilt;
    d.useColumn = c->cfg.useColumnPackage;
    d.columnStrength = c->cfg.useColumnPackage && c->cfg.useColumnVerticalThermodynamics;
    d.kstrength = c->cfg.iceStrengthFormulation;
    d.krdgPartic = c->cfg.ridgingParticipation;
    d.krdgRedist = c->cfg.ridgingRedistribution;
    d.muRdg = c->cfg.ridgingEfoldingScale;
    d.Cf = c->cfg.ridgingWorkRatio;
    d.openWater = fd("openWaterArea");
    d.haloVertices = c->cfg.computeHaloVertices ? 1 : 0;
    d.symMetric = c->symMetric ? 1 : 0;
    d.deriveGrad = c->deriveGrad ? 1 : 0;
    d.deriveRow = c->deriveRow ? 1 : 0;
    {
        // Blocks from about 120 000 cells on move more per iteration than the caches hold: level 1 marks the stresses
        // and the partial sums (measured: 164 k cells 0.2 % faster, 655 k cells 6 %; 88 k cells 2.5 % slower, 47 k cells
        // 3 % slower).  Where even the basis arrays cannot survive from one iteration to the next, level 2 marks them
        // too (655 k cells: 12 % faster than without marks; at 164 k cells, where they do survive, 12 % slower).
        // SIH_STREAM_HINTS = 0 / 1 / 2 overrides.
        const char *e = getenv("SIH_STREAM_HINTS");
        d.streamHints = e ? atoi(e) : (c->nCells >= 400000 ? 2 : (c->nCells >= 120000 ? 1 : 0));
        // the cell kernel's blocks in contiguous eighths per XCD (xcdChunkedBlock): 164 k cells 5 % faster, 655 k cells
        // 4 % slower (the vertex kernel gains at every size and always has it); SIH_XCD_CHUNKS = 0 / 1 overrides
        const char *x = getenv("SIH_XCD_CHUNKS");
        d.chunkCells = x ? (atoi(x) != 0) : (c->nCells < 400000);
    }
}

__device__ inline double sign1(double x) { return copysign(1.0, x); }

// =========================================================================== pre-subcycle kernels

// stage 0: aggregate_mass_and_area, velocity_solver.F:685-752 (loops to nCells, halo included) and
// iceAreaCellInitial = iceAreaCell (:819), unless the column physics has set it earlier in the time step (:808)
__global__ void k_pre0_cells(SihDev d)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    double a = 0.0, vi = 0.0, vs = 0.0;
    for (int k = 0; k < d.nCat; k++) {
        a = a + d.iceAreaCat[(size_t)k * d.padC + c];
        vi = vi + d.iceVolCat[(size_t)k * d.padC + c];
        vs = vs + d.snowVolCat[(size_t)k * d.padC + c];
    }
    d.iceAreaCell[c] = a;
    d.iceVolumeCell[c] = vi;
    d.snowVolumeCell[c] = vs;
    d.totalMassCell[c] = vi * RHOI + vs * RHOS;
    if (!d.columnStrength) d.iceAreaCellInitial[c] = a;
}

// seaice_interpolate_cell_to_vertex, mesh.F:2835-2851 (the sentinel slot holds area 0, value 0)
__device__ inline double interpC2V(const SihDev &d, int v, const double *cellVar)
{
    double acc = 0.0, tot = 0.0;
    for (int k = 0; k < d.D; k++) {
        int c = d.cov[k * d.padV + v];
        acc = acc + d.areaCell[c] * cellVar[c];
        tot = tot + d.areaCell[c];
    }
    return acc / tot;
}

// stage 1a: area/mass to vertices (:883-891) and velocity_calculation_mask (:1073-1150)
__global__ void k_pre1_vertices(SihDev d, int calcMasks)
{
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= d.nVertices) return;
    if (d.vActive[v]) {
        double av = interpC2V(d, v, d.iceAreaCellInitial);
        double mv = interpC2V(d, v, d.totalMassCell);
        d.iceAreaVertex[v] = av;
        d.totalMassVertex[v] = mv;
        if (calcMasks) {
            int s = 0;
            if (d.interiorVertex[v] == 1 && d.landIceMaskVertex[v] == 0 && av > AREA_MIN && mv > MASS_MIN) s = 1;
            d.solveVelocity[v] = s;
        }
    } else if (calcMasks) {
        d.solveVelocity[v] = 0;
    }
}

// stage 1b: stress_calculation_mask, velocity_solver.F:961-1059
__global__ void k_pre1_cells(SihDev d)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    int s = 0;
    if (d.iceAreaCellInitial[c] > AREA_MIN && d.totalMassCell[c] > MASS_MIN && d.landIceMask[c] == 0) {
        s = 1;
    } else {
        int E = d.nEdgesOnCell[c];
        for (int i = 0; i < E; i++) {
            int nb = d.coc[i * d.padC + c];
            if (d.iceAreaCellInitial[nb] > AREA_MIN && d.totalMassCell[nb] > MASS_MIN && d.landIceMask[nb] == 0) {
                s = 1;
                break;
            }
        }
    }
    d.solveStress[c] = s;
}

// stage 2: new_ice_velocities, velocity_solver.F:1238-1277
__global__ void k_pre2_vertices(SihDev d)
{
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= d.nVertices) return;
    if (d.vActive[v]) {
        double uo = interpC2V(d, v, d.uOcn);
        double vo = interpC2V(d, v, d.vOcn);
        d.uOcnVert[v] = uo;
        d.vOcnVert[v] = vo;
        if (d.solveVelocity[v] == 1) {
            if (d.solveVelocityPrevious[v] == 0) {
                d.u[v] = uo;
                d.v[v] = vo;
            }
        } else {
            d.u[v] = 0.0;
            d.v[v] = 0.0;
            d.divU[v] = 0.0;
            d.divV[v] = 0.0;
            d.ocnStressU[v] = 0.0;
            d.ocnStressV[v] = 0.0;
        }
    }
    d.solveVelocityPrevious[v] = d.solveVelocity[v];
    d.uInit[v] = d.u[v];
    d.vInit[v] = d.v[v];
}

// colpkg_ice_strength (src/column/ice_colpkg.F90:3498-3610) with its ridge_itd (src/column/ice_mechred.F90:
// 741-1012): Rothrock's strength from the potential energy change of the ridging ice.  Once per cell and
// dynamics step; the per-category work arrays are small and dynamically indexed, which is fine here.
__device__ double columnIceStrength(const SihDev &d, int c)
{
    const double puny = 1.0e-11, c0 = 0.0, c1 = 1.0, c2 = 2.0, p5 = 0.5, p333 = 1.0 / 3.0;
    const double Gstar = 0.15, astar = 0.05, maxraft = 1.0, Hstar = 25.0;  // ice_mechred.F90:54-61
    const double Gstari = c1 / Gstar, astari = c1 / astar;
    const double Cp = 0.5 * 9.80616 * (1026.0 - 917.0) * 917.0 / 1026.0;  // ice_constants_colpkg.F90:60
    const int ncat = d.nCat;
    const size_t pc = d.padC;
    if (d.kstrength != 1) return PSTAR * d.iceVolumeCell[c] * exp(-CSTAR * (c1 - d.iceAreaCell[c]));
    const double aice0 = d.openWater[c];
    double GsumA[SIH_MAXCAT + 2], apartic[SIH_MAXCAT + 1], hrmin[SIH_MAXCAT + 1], hrmax[SIH_MAXCAT + 1],
        hrexp[SIH_MAXCAT + 1], krdg[SIH_MAXCAT + 1];
    double *Gsum = GsumA + 1;  // Gsum(-1:ncat)
    Gsum[-1] = c0;
    Gsum[0] = aice0 > puny ? aice0 : Gsum[-1];
    apartic[0] = c0;
    for (int n = 1; n <= ncat; n++) {
        const double a = d.iceAreaCat[(size_t)(n - 1) * pc + c];
        apartic[n] = c0;
        hrmin[n] = c0;
        hrmax[n] = c0;
        hrexp[n] = c0;
        krdg[n] = c1;
        Gsum[n] = a > puny ? Gsum[n - 1] + a : Gsum[n - 1];
    }
    const double work = c1 / Gsum[ncat];
    for (int n = 0; n <= ncat; n++) Gsum[n] = Gsum[n] * work;
    if (d.krdgPartic == 0) {  // Thorndike et al. 1975
        for (int n = 0; n <= ncat; n++) {
            if (Gsum[n] < Gstar)
                apartic[n] = Gstari * (Gsum[n] - Gsum[n - 1]) * (c2 - Gstari * (Gsum[n - 1] + Gsum[n]));
            else if (Gsum[n - 1] < Gstar)
                apartic[n] = Gstari * (Gstar - Gsum[n - 1]) * (c2 - Gstari * (Gstar + Gsum[n - 1]));
        }
    } else {  // exponential dependence on G(h)
        const double xtmp = c1 / (c1 - exp(-astari));
        Gsum[-1] = exp(-Gsum[-1] * astari) * xtmp;
        for (int n = 0; n <= ncat; n++) {
            Gsum[n] = exp(-Gsum[n] * astari) * xtmp;
            apartic[n] = Gsum[n - 1] - Gsum[n];
        }
    }
    for (int n = 1; n <= ncat; n++) {
        const double a = d.iceAreaCat[(size_t)(n - 1) * pc + c], v = d.iceVolCat[(size_t)(n - 1) * pc + c];
        if (a > puny) {
            double hi = v / a;
            if (d.krdgRedist == 0) {  // Hibler 1980
                hrmin[n] = fmin(c2 * hi, hi + maxraft);
                hrmax[n] = c2 * sqrt(Hstar * hi);
                hrmax[n] = fmax(hrmax[n], hrmin[n] + puny);
                const double hrmean = p5 * (hrmin[n] + hrmax[n]);
                krdg[n] = hrmean / hi;
            } else {  // exponential redistribution
                hi = fmax(hi, puny);
                hrmin[n] = fmin(c2 * hi, hi + maxraft);
                hrexp[n] = d.muRdg * sqrt(hi);
                krdg[n] = (hrmin[n] + hrexp[n]) / hi;
            }
        }
    }
    double aksum = apartic[0];
    for (int n = 1; n <= ncat; n++) aksum = aksum + apartic[n] * (c1 - c1 / krdg[n]);
    double strength = 0.0;
    for (int n = 1; n <= ncat; n++) {
        const double a = d.iceAreaCat[(size_t)(n - 1) * pc + c], v = d.iceVolCat[(size_t)(n - 1) * pc + c];
        if (a > puny && apartic[n] > c0) {
            const double hi = v / a;
            double h2rdg;
            if (d.krdgRedist == 0)
                h2rdg = p333 * (hrmax[n] * hrmax[n] * hrmax[n] - hrmin[n] * hrmin[n] * hrmin[n]) / (hrmax[n] - hrmin[n]);
            else
                h2rdg = hrmin[n] * hrmin[n] + c2 * hrmin[n] * hrexp[n] + c2 * hrexp[n] * hrexp[n];
            const double dh2rdg = -hi * hi + h2rdg / krdg[n];
            strength = strength + apartic[n] * dh2rdg;
        }
    }
    return d.Cf * Cp * strength / aksum;
}

// stage 3: ice_strength (:1419-1458) and constant_air_stress (:1665-1728)
__global__ void k_pre3_cells(SihDev d)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCellsSolve) return;
    if (d.solveStress[c] == 1)
        d.icePressure[c] = d.columnStrength ? columnIceStrength(d, c)
                                            : PSTAR * d.iceVolumeCell[c] * exp(-CSTAR * (1.0 - d.iceAreaCell[c]));
    else
        d.icePressure[c] = 0.0;
    if (!d.useAir) {
        d.airCellU[c] = 0.0;
        d.airCellV[c] = 0.0;
    } else if (!d.columnStrength) {
        double ua = d.uAir[c], va = d.vAir[c];
        double windSpeed = sqrt(ua * ua + va * va);
        d.airCellU[c] = d.airDensity[c] * windSpeed * 0.0012 * ua * d.iceAreaCell[c];
        d.airCellV[c] = d.airDensity[c] * windSpeed * 0.0012 * va * d.iceAreaCell[c];
    }  // else: airStressCellU / V come from the column physics' atmosphere fluxes (:1562-1565)
}

// stage 4 (vertices): air stress to vertices (:1638-1646), coriolis_force_coefficient (:1742-1788),
// ocean_stress (:1802-1883), surface_tilt (:1897-2213), vertex part of init_subcycle_variables (:2287-2310)
__global__ void k_pre4_vertices(SihDev d)
{
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= d.nVertices) return;
    d.divU[v] = 0.0;
    d.divV[v] = 0.0;
    if (!d.vActive[v]) {
        if (!d.useOcean) {
            d.ocnStressU[v] = 0.0;
            d.ocnStressV[v] = 0.0;
        }
        if (d.tiltMode == 0) {
            d.tiltForceU[v] = 0.0;
            d.tiltForceV[v] = 0.0;
        }
        return;
    }
    d.airVertU[v] = interpC2V(d, v, d.airCellU);
    d.airVertV[v] = interpC2V(d, v, d.airCellV);
    double m = d.totalMassVertex[v], f = d.fVertex[v];
    d.massF[v] = m * f;
    int solve = d.solveVelocity[v] == 1;
    double uo = d.uOcnVert[v], vo = d.vOcnVert[v];
    if (d.useOcean && solve) {
        d.ocnStressU[v] = uo * COS_TURN - vo * SIN_TURN * sign1(f);
        d.ocnStressV[v] = uo * SIN_TURN * sign1(f) + vo * COS_TURN;
    } else {
        d.ocnStressU[v] = 0.0;
        d.ocnStressV[v] = 0.0;
    }
    if (d.tiltMode == 2) {
        d.tiltVertU[v] = interpC2V(d, v, d.tiltU);
        d.tiltVertV[v] = interpC2V(d, v, d.tiltV);
    }
    if (d.tiltMode != 0 && solve) {
        if (d.tiltMode == 1) {
            d.tiltForceU[v] = -f * m * vo;
            d.tiltForceV[v] = f * m * uo;
        } else {
            d.tiltForceU[v] = -GRAVIT * m * d.tiltVertU[v];
            d.tiltForceV[v] = -GRAVIT * m * d.tiltVertV[v];
        }
    } else {
        d.tiltForceU[v] = 0.0;
        d.tiltForceV[v] = 0.0;
    }
    // what the momentum equation's right-hand side holds unchanged through the subcycles: solve_velocity adds the air
    // stress and the tilt force to the stress divergence one after the other (:3177-3199); the subcycle kernels add
    // their sum, formed here once (the difference is one rounding of the sum, far inside the parity bar, and two arrays
    // less to fetch per vertex and subcycle)
    d.airTiltU[v] = d.airVertU[v] + d.tiltForceU[v];
    d.airTiltV[v] = d.airVertV[v] + d.tiltForceV[v];
    if (!solve) {
        d.u[v] = 0.0;
        d.v[v] = 0.0;
        d.ocnCoeff[v] = 0.0;
    }
}

// stage 4 (cells): cell part of init_subcycle_variables (:2322-2345); also clears the per-corner
// divergence partial sums, which unsolved cells never write (their stresses are zero all step long)
__global__ void k_pre4_cells(SihDev d)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    bool unsolved = d.solveStress[c] != 1;
    for (int s = 0; s < d.M; s++) {
        size_t k = (size_t)s * d.padC + c;
        d.strain11[k] = 0.0;
        d.strain22[k] = 0.0;
        d.strain12[k] = 0.0;
        d.partU[k] = 0.0;
        d.partV[k] = 0.0;
        if (unsolved) {
            d.stress11[k] = 0.0;
            d.stress22[k] = 0.0;
            d.stress12[k] = 0.0;
        }
    }
}

// =========================================================================== subcycle kernels

// Fused cell kernel: seaice_strain_tensor_variational (variational.F:575-670), the constitutive
// relation (constitutive_relation.F:178-373 via variational.F:777-975) and the cell-local inner sums of
// seaice_stress_divergence_variational (variational.F:1151-1170).  The latter are the per-(cell, corner)
// partial sums stressDivergenceUCell / VCell; the vertex kernel adds the three of them in
// cellsOnVertex order exactly as the reference loop does (:1172-1173), so no rounding changes.
// MODE 0: all of it in one pass.  With config_average_variational_strain the strains must pass through
// memory and a vertex loop in between (seaice_average_strains_on_vertex): MODE 1 stops after storing the
// strains, MODE 2 starts from the stored (averaged) strains.  MODE 3 is the inner part of the divergence
// operator on its own: the stresses are read from memory as they are, for every cell of the block.
template <int M, bool SPARSE, bool LAST, int MODE>
__global__ void __launch_bounds__(256) k_sub_cells(SihDev d, const int *__restrict__ list, int nList)
{
    // list == nullptr: every cell; else the nList cells it names (ascending, so loads stay mostly coalesced)
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (list) {
        if (c >= nList) return;
        c = list[c];
    } else if (c >= d.nCells) {
        return;
    }
    const int P = d.padC;
    if (MODE != 3 && d.solveStress[c] != 1) {
        if (LAST && MODE != 1 && d.relation == 1) {
#pragma unroll
            for (int s = 0; s < M; s++) d.Prep[(size_t)s * P + c] = 0.0;
        }
        return;
    }
    const int E = d.nEdgesOnCell[c];
    double u[M], v[M], t[M];
#pragma unroll
    for (int s = 0; s < M; s++) {
        u[s] = 0.0;
        v[s] = 0.0;
        t[s] = 0.0;
        if (s < E) {
            int iv = d.voc[(size_t)s * P + c];
            u[s] = d.u[iv];
            v[s] = d.v[iv];
            t[s] = d.tanLat[iv];
        }
    }
    double uLast = 0.0, vLast = 0.0;
#pragma unroll
    for (int s = 0; s < M; s++)
        if (s == E - 1) {
            uLast = u[s];
            vLast = v[s];
        }
    const double P_ice = d.icePressure[c];
    const double dtE = d.dtE, T = d.dampT;
    double s11[M], s22[M], s12[M];
#pragma unroll
    for (int g = 0; g < M; g++) {
        s11[g] = 0.0;
        s22[g] = 0.0;
        s12[g] = 0.0;
        if (MODE == 3) {
            if (g < E) {
                s11[g] = d.stress11[(size_t)g * P + c];
                s22[g] = d.stress22[(size_t)g * P + c];
                s12[g] = d.stress12[(size_t)g * P + c];
            }
            continue;
        }
        if (g < E) {
            double e11 = 0.0, e22 = 0.0, e12 = 0.0;
            const size_t k = (size_t)g * P + c;
            if (MODE == 2) {
                e11 = d.strain11[k];
                e22 = d.strain22[k];
                e12 = d.strain12[k];
            } else if (SPARSE) {
                // the three non-zero basis gradients of corner g, met in ascending basis-vertex order
                double x0u, x1u, x2u, x0v, x1v, x2v;
                const bool isLast = (g == E - 1);
                if (g == 0) {
                    x0u = u[0];
                    x0v = v[0];
                    x1u = u[1];
                    x1v = v[1];
                    x2u = uLast;
                    x2v = vLast;
                } else {
                    const int gn = (g + 1 < M) ? g + 1 : g;
                    x0u = isLast ? u[0] : u[g - 1];
                    x0v = isLast ? v[0] : v[g - 1];
                    x1u = isLast ? u[g - 1] : u[g];
                    x1v = isLast ? v[g - 1] : v[g];
                    x2u = isLast ? u[g] : u[gn];
                    x2v = isLast ? v[g] : v[gn];
                }
                const double xu[3] = {x0u, x1u, x2u}, xv[3] = {x0v, x1v, x2v};
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    double gu = d.gradU3[(size_t)(q * M + g) * P + c];
                    double gv = d.gradV3[(size_t)(q * M + g) * P + c];
                    e11 = e11 + xu[q] * gu;
                    e22 = e22 + xv[q] * gv;
                    e12 = e12 + 0.5 * (xu[q] * gv + xv[q] * gu);
                }
            } else {
#pragma unroll
                for (int b = 0; b < M; b++) {
                    if (b < E) {
                        double gu = d.gradU[(size_t)(g * M + b) * P + c];
                        double gv = d.gradV[(size_t)(g * M + b) * P + c];
                        e11 = e11 + u[b] * gu;
                        e22 = e22 + v[b] * gv;
                        e12 = e12 + 0.5 * (u[b] * gv + v[b] * gu);
                    }
                }
            }
            if (MODE != 2) {
                e11 = e11 - v[g] * t[g];
                e12 = e12 + u[g] * t[g] * 0.5;
            }
            if ((LAST && MODE == 0) || MODE == 1) {
                d.strain11[k] = e11;
                d.strain22[k] = e22;
                d.strain12[k] = e12;
            }
            if (MODE == 1) continue;
            double a11, a22, a12;
            if (d.relation == 3) {
                a11 = 1.0 * e11;
                a22 = 1.0 * e22;
                a12 = 1.0 * e12;
            } else {
                a11 = d.stress11[k];
                a22 = d.stress22[k];
                a12 = d.stress12[k];
                double eD = e11 + e22, eT = e11 - e22, eS = e12 * 2.0;
                double st1 = a11 + a22, st2 = a11 - a22;
                double Delta = sqrt(eD * eD + (eT * eT + eS * eS) / ECC2);
                double pc = P_ice / fmax(Delta, PUNY);
                if (LAST) d.Prep[k] = pc * Delta;
                double den;
                if (d.relation == 1) {
                    pc = (pc * dtE) / (2.0 * T);
                    den = 1.0 + (0.5 * dtE) / T;
                } else {
                    pc = (pc * 2.0 * 5.5e-3) / 0.86;
                    den = 1.0 + (2.0 * 5.5e-3) / 0.86;
                }
                st1 = (st1 + pc * (eD - Delta)) / den;
                st2 = (st2 + (pc / ECC2) * eT) / den;
                a12 = (a12 + (pc / ECC2) * eS * 0.5) / den;
                a11 = 0.5 * (st1 + st2);
                a22 = 0.5 * (st1 - st2);
            }
            d.stress11[k] = a11;
            d.stress22[k] = a22;
            d.stress12[k] = a12;
            s11[g] = a11;
            s22[g] = a22;
            s12[g] = a12;
        } else if (LAST && MODE != 1 && d.relation == 1) {
            d.Prep[(size_t)g * P + c] = 0.0;
        }
    }
    if (MODE == 1) return;
    // per-corner partial sums of the stress divergence
    if (!d.deriveRow) {
#pragma unroll
        for (int j = 0; j < M; j++) {
            if (j < E) {
                double cu = 0.0, cv = 0.0;
                const double tj = t[j];
#pragma unroll
                for (int s = 0; s < M; s++) {
                    if (s < E) {
                        const size_t kb = (size_t)(j * M + s) * P + c;
                        const double bu = d.SU[kb], bv = d.SV[kb], bm = d.SM[kb];
                        cu = cu - s11[s] * bu - s12[s] * bv - s12[s] * bm * tj;
                        cv = cv - s22[s] * bv - s12[s] * bu + s11[s] * bm * tj;
                    }
                }
                d.partU[(size_t)j * P + c] = cu;
                d.partV[(size_t)j * P + c] = cv;
            }
        }
        return;
    }
    // the arithmetic of k_sub_cells_split with deriveRow: the terms with basisIntegralsU / V (nu, nv) and the metric
    // terms (mu, mv) are added up separately, and in a marked cell corner 0 takes minus the other corners' nu, nv
    const bool rowCell = (d.nEdgesDerive[c] & 512) != 0;
    double du = 0.0, dv = 0.0;
#pragma unroll
    for (int jj = 1; jj <= M; jj++) {
        const int j = jj == M ? 0 : jj;  // corner 0 last
        if (j < E) {
            double nu = 0.0, nv = 0.0, mu = 0.0, mv = 0.0;
            const double tj = t[j];
#pragma unroll
            for (int s = 0; s < M; s++) {
                if (s < E) {
                    const size_t kb = (size_t)(j * M + s) * P + c;
                    const double bm = d.SM[kb];
                    mu = mu - s12[s] * bm * tj;
                    mv = mv + s11[s] * bm * tj;
                    if (!(rowCell && j == 0)) {
                        const double bu = d.SU[kb], bv = d.SV[kb];
                        nu = nu - s11[s] * bu - s12[s] * bv;
                        nv = nv - s22[s] * bv - s12[s] * bu;
                    }
                }
            }
            if (rowCell && j == 0) {
                d.partU[c] = mu - du;
                d.partV[c] = mv - dv;
            } else {
                d.partU[(size_t)j * P + c] = nu + mu;
                d.partV[(size_t)j * P + c] = nv + mv;
                // (in the order the last wave of the split kernel adds them: corner 1 first)
                du = j == 1 ? nu : du + nu;
                dv = j == 1 ? nv : dv + nv;
            }
        }
    }
}

// Workgroups are dealt round-robin over the eight XCDs, each with an L2 of its own: with this permutation of the block
// index an XCD works on one contiguous eighth of the range, so that neighbouring blocks, which gather from the same
// cache lines along their common border, mostly share an L2.
__device__ __forceinline__ int xcdChunkedBlock()
{
    const int b = blockIdx.x, nb = gridDim.x, q = nb / 8, r = nb % 8, x = b % 8;
    return x * q + (x < r ? x : r) + b / 8;
}

// Loads and stores of data that passes through once an iteration (stresses, partial sums), non-temporal on request:
// on a block whose working set does not fit the caches that leaves more of them (the 256 MB Infinity Cache above all)
// to the arrays the next iteration reads again.  On a block that does fit, the same marks cost time.
template <bool NT>
__device__ __forceinline__ double ldStreamT(const double *p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <bool NT>
__device__ __forceinline__ void stStreamT(double *p, double v)
{
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}
#define ldStream(p, nt) ldStreamT<nt>(p)
#define stStream(p, v, nt) stStreamT<nt>(p, v)

// The same arithmetic as k_sub_cells<M, SPARSE, LAST, 0> with one thread per (cell, corner): a block is 64
// consecutive cells (threadIdx.x) times M corners (threadIdx.y), so every wave still reads 64 consecutive
// entries of one plane.  Corner velocities and the new corner stresses are exchanged through LDS.  Each
// thread issues 1/M of the loads of the one-thread-per-cell kernel, which shortens the dependent chain
// that dominates on small blocks (many-GPU runs), at the price of two block barriers.
template <int M, bool SPARSE, bool LAST, int NT>
__global__ void __launch_bounds__(64 * M) k_sub_cells_split(SihDev d, const int *__restrict__ list, int nList)
{
    __shared__ double su[M][64], sv[M][64], s11[M][64], s22[M][64], s12[M][64];
    __shared__ int arrived;  // waves of the block that have left their part of corner 0's sums (deriveRow)
    const int lane = threadIdx.x, g = threadIdx.y;
    int c = (d.chunkCells ? xcdChunkedBlock() : (int)blockIdx.x) * 64 + lane;
    bool inRange;
    if (list) {
        inRange = c < nList;
        c = inRange ? list[c] : 0;
    } else {
        inRange = c < d.nCells;
        if (!inRange) c = 0;
    }
    const int P = d.padC;
    constexpr bool nt = NT >= 1, ntAll = NT >= 2;  // (a template parameter: behind a run-time condition the compiler merges the two loads into a plain one)
    const bool solve = inRange && d.solveStress[c] == 1;
    // (with derived gradients or integral rows: the number of edges with the cell's marks in bits 8 and 9, from the
    // set-up)
    const int Ev = inRange ? ((SPARSE && d.deriveGrad) || d.deriveRow ? d.nEdgesDerive[c] : d.nEdgesOnCell[c]) : 0;
    const int E = Ev & 255;
    // this cell's first row of basisIntegralsU / V is minus the sum of its other rows, bit for bit
    const bool rowCell = d.deriveRow && (Ev & 512);
    const bool corner = solve && g < E;
    const size_t k = (size_t)g * P + c;
    double ug = 0.0, vg = 0.0, tg = 0.0;
    if (corner) {
        const int iv = d.voc[k];
        ug = d.u[iv];
        vg = d.v[iv];
        tg = d.tanLat[iv];
    }
    su[g][lane] = ug;
    sv[g][lane] = vg;
    if (g == 0 && lane == 0) arrived = 0;
    __syncthreads();
    double a11 = 0.0, a22 = 0.0, a12 = 0.0;
    if (corner) {
        double e11 = 0.0, e22 = 0.0, e12 = 0.0;
        if (SPARSE) {
            // the three non-zero basis gradients of corner g, met in ascending basis-vertex order
            int i0, i1, i2;
            if (g == 0) {
                i0 = 0;
                i1 = 1;
                i2 = E - 1;
            } else if (g == E - 1) {
                i0 = 0;
                i1 = g - 1;
                i2 = g;
            } else {
                i0 = g - 1;
                i1 = g;
                i2 = g + 1;
            }
            const double xu[3] = {su[i0][lane], su[i1][lane], su[i2][lane]};
            const double xv[3] = {sv[i0][lane], sv[i1][lane], sv[i2][lane]};
            // the corner's own gradient (slot `own`) is minus the sum of the other two in the cells where the set-up found
            // the partition of unity to hold to rounding: 32 instead of 48 bytes of gradients per corner from memory
            const int own = (d.deriveGrad && (Ev & 256)) ? (g == 0 ? 0 : (g == E - 1 ? 2 : 1)) : -1;
            double gu3[3], gv3[3];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                gu3[q] = 0.0;
                gv3[q] = 0.0;
                if (q != own) {
                    gu3[q] = ldStream(&d.gradU3[(size_t)(q * M + g) * P + c], ntAll);
                    gv3[q] = ldStream(&d.gradV3[(size_t)(q * M + g) * P + c], ntAll);
                }
            }
            if (own >= 0) {
                const double du = -((own == 0 ? gu3[1] : gu3[0]) + (own == 2 ? gu3[1] : gu3[2]));
                const double dv = -((own == 0 ? gv3[1] : gv3[0]) + (own == 2 ? gv3[1] : gv3[2]));
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    gu3[q] = q == own ? du : gu3[q];
                    gv3[q] = q == own ? dv : gv3[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const double gu = gu3[q], gv = gv3[q];
                e11 = e11 + xu[q] * gu;
                e22 = e22 + xv[q] * gv;
                e12 = e12 + 0.5 * (xu[q] * gv + xv[q] * gu);
            }
        } else {
#pragma unroll
            for (int b = 0; b < M; b++) {
                if (b < E) {
                    const double gu = d.gradU[(size_t)(g * M + b) * P + c];
                    const double gv = d.gradV[(size_t)(g * M + b) * P + c];
                    e11 = e11 + su[b][lane] * gu;
                    e22 = e22 + sv[b][lane] * gv;
                    e12 = e12 + 0.5 * (su[b][lane] * gv + sv[b][lane] * gu);
                }
            }
        }
        e11 = e11 - vg * tg;
        e12 = e12 + ug * tg * 0.5;
        if (LAST) {
            d.strain11[k] = e11;
            d.strain22[k] = e22;
            d.strain12[k] = e12;
        }
        if (d.relation == 3) {
            a11 = 1.0 * e11;
            a22 = 1.0 * e22;
            a12 = 1.0 * e12;
        } else {
            const double P_ice = d.icePressure[c];
            const double dtE = d.dtE, T = d.dampT;
            // (the stresses and the sums below pass through once an iteration: on a block too large for the caches
            // they are marked non-temporal and leave more room to the basis arrays, which the next iteration reads again)
            a11 = ldStream(&d.stress11[k], nt);
            a22 = ldStream(&d.stress22[k], nt);
            a12 = ldStream(&d.stress12[k], nt);
            double eD = e11 + e22, eT = e11 - e22, eS = e12 * 2.0;
            double st1 = a11 + a22, st2 = a11 - a22;
            double Delta = sqrt(eD * eD + (eT * eT + eS * eS) / ECC2);
            double pc = P_ice / fmax(Delta, PUNY);
            if (LAST) d.Prep[k] = pc * Delta;
            double den;
            if (d.relation == 1) {
                pc = (pc * dtE) / (2.0 * T);
                den = 1.0 + (0.5 * dtE) / T;
            } else {
                pc = (pc * 2.0 * 5.5e-3) / 0.86;
                den = 1.0 + (2.0 * 5.5e-3) / 0.86;
            }
            st1 = (st1 + pc * (eD - Delta)) / den;
            st2 = (st2 + (pc / ECC2) * eT) / den;
            a12 = (a12 + (pc / ECC2) * eS * 0.5) / den;
            a11 = 0.5 * (st1 + st2);
            a22 = 0.5 * (st1 - st2);
        }
        stStream(&d.stress11[k], a11, nt);
        stStream(&d.stress22[k], a22, nt);
        stStream(&d.stress12[k], a12, nt);
    } else if (LAST && inRange && d.relation == 1) {
        d.Prep[k] = 0.0;  // unsolved cells and the unused corners of solved ones
    }
    s11[g][lane] = a11;
    s22[g][lane] = a22;
    s12[g][lane] = a12;
    __syncthreads();
    // Partial sums of the stress divergence for corner j = g.
    if (!d.deriveRow) {  // (the same for every thread of the launch)
        if (corner) {
            // (all loads first: requested inside the loop of the sums, every pass waited for its own three)
            double bu[M], bv[M], bm[M];
#pragma unroll
            for (int s = 0; s < M; s++) {
                bu[s] = 0.0;
                bv[s] = 0.0;
                bm[s] = 0.0;
                if (s < E) {
                    const size_t kb = (size_t)(g * M + s) * P + c;
                    // a symmetric metric matrix is read from one side of its diagonal only: 21 planes of HBM traffic
                    // instead of 36 for M = 6 (the corners of a cell sit in different waves of this block)
                    const size_t km = d.symMetric ? (size_t)((g > s ? g : s) * M + (g > s ? s : g)) * P + c : kb;
                    bu[s] = d.SU[kb];
                    bv[s] = d.SV[kb];
                    bm[s] = d.SM[km];
                }
            }
            double cu = 0.0, cv = 0.0;
#pragma unroll
            for (int s = 0; s < M; s++) {
                if (s < E) {
                    const double x11 = s11[s][lane], x22 = s22[s][lane], x12 = s12[s][lane];
                    cu = cu - x11 * bu[s] - x12 * bv[s] - x12 * bm[s] * tg;
                    cv = cv - x22 * bv[s] - x12 * bu[s] + x11 * bm[s] * tg;
                }
            }
            d.partU[k] = cu;
            d.partV[k] = cv;
        }
        return;
    }
    // The basis functions sum to one, so summed over the velocity corners of a cell the terms with basisIntegralsU / V
    // cancel.  Every corner adds up those terms (nu, nv) and the metric terms (mu, mv) separately; in a marked cell
    // (bit 9, set at set-up where the integrals obey the cancellation to rounding) corner 0 does not fetch its rows of
    // U and V: its sums are its metric terms minus the other corners' nu, nv, added in ascending corner order.  The
    // corners of a cell sit in the block's M waves: each leaves its part in LDS and counts itself in, and the wave that
    // comes last finishes corner 0 for the block's 64 cells -- no wave waits for another.
    const bool cellDerives = solve && rowCell;
    double nu = 0.0, nv = 0.0, mu = 0.0, mv = 0.0;
    if (corner) {
        const bool skipsRows = cellDerives && g == 0;
        // (all loads first, the skipped rows in one predicated region: requested inside the loop below, every pass
        // waited for its own loads)
        double bu[M], bv[M], bm[M];
#pragma unroll
        for (int s = 0; s < M; s++) {
            bu[s] = 0.0;
            bv[s] = 0.0;
            bm[s] = 0.0;
            if (s < E) {
                const size_t kb = (size_t)(g * M + s) * P + c;
                bm[s] = ldStream(&d.SM[d.symMetric ? (size_t)((g > s ? g : s) * M + (g > s ? s : g)) * P + c : kb], ntAll);
            }
        }
        if (!skipsRows) {
#pragma unroll
            for (int s = 0; s < M; s++) {
                if (s < E) {
                    const size_t kb = (size_t)(g * M + s) * P + c;
                    bu[s] = ldStream(&d.SU[kb], ntAll);
                    bv[s] = ldStream(&d.SV[kb], ntAll);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < M; s++) {
            if (s < E) {
                const double x11 = s11[s][lane], x22 = s22[s][lane], x12 = s12[s][lane];
                mu = mu - x12 * bm[s] * tg;
                mv = mv + x11 * bm[s] * tg;
                nu = nu - x11 * bu[s] - x12 * bv[s];
                nv = nv - x22 * bv[s] - x12 * bu[s];
            }
        }
        if (cellDerives) {  // (su / sv: every corner has taken its velocities from them before the second barrier)
            su[g][lane] = g == 0 ? mu : nu;
            sv[g][lane] = g == 0 ? mv : nv;
        }
    }
    int lastWave = 0;
    if (lane == 0)
        lastWave = __hip_atomic_fetch_add(&arrived, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == M - 1 ? 1 : 0;
    lastWave = __builtin_amdgcn_readfirstlane(lastWave);
    // (stores to memory after the count: a release waits for the stores before it)
    if (corner && !(cellDerives && g == 0)) {
        stStream(&d.partU[k], nu + mu, nt);
        stStream(&d.partV[k], nv + mv, nt);
    }
    if (lastWave && cellDerives) {
        double du = su[1][lane], dv = sv[1][lane];
#pragma unroll
        for (int j = 2; j < M; j++) {
            if (j < E) {
                du = du + su[j][lane];
                dv = dv + sv[j][lane];
            }
        }
        stStream(&d.partU[c], su[0][lane] - du, nt);
        stStream(&d.partV[c], sv[0][lane] - dv, nt);
    }
}

// seaice_average_strains_on_vertex (variational.F:684-763): area-weighted mean of the (up to) three corner
// strains that meet at a vertex, written back to those corners.  A (corner, cell) entry belongs to exactly
// one vertex, so threads do not collide; sums run in cellsOnVertex order like the reference's.
__global__ void __launch_bounds__(256) k_avg_strain_vertices(SihDev d)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= d.nVerticesSolve) return;
    const int PV = d.padV;
    double a11 = 0.0, a22 = 0.0, a12 = 0.0, den = 0.0;
    int g[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ok[k] = false;
        g[k] = 0;
        if (k >= d.D) continue;
        const int c = d.cov[k * PV + v];
        ok[k] = c < d.nCells;
        g[k] = d.gidx[k * PV + v];
        if (ok[k]) {
            const double area = d.areaCell[c];
            a11 = a11 + d.strain11[g[k]] * area;
            a22 = a22 + d.strain22[g[k]] * area;
            a12 = a12 + d.strain12[g[k]] * area;
            den = den + area;
        }
    }
    a11 = a11 / den;
    a22 = a22 / den;
    a12 = a12 / den;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (ok[k]) {
            d.strain11[g[k]] = a11;
            d.strain22[g[k]] = a22;
            d.strain12[g[k]] = a12;
        }
}

// seaice_set_special_boundaries_velocity (special_boundaries.F:253-331) in two steps, so that every
// marked vertex sees the value its (resolved) source had before the call
__global__ void k_sb_gather(int n, const int *__restrict__ origin, const int *__restrict__ sign,
                            const double *__restrict__ u, const double *__restrict__ v, double *__restrict__ tmp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int sg = sign[i];
    double uu = 0.0, vv = 0.0;
    if (sg != 0) {
        uu = u[origin[i]];
        vv = v[origin[i]];
        if (sg < 0) {
            uu = -uu;
            vv = -vv;
        }
    }
    tmp[i] = uu;
    tmp[n + i] = vv;
}

__global__ void k_sb_scatter(int n, const int *__restrict__ idx, const double *__restrict__ tmp, double *__restrict__ u,
                             double *__restrict__ v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u[idx[i]] = tmp[i];
    v[idx[i]] = tmp[n + i];
}

// seaice_set_special_boundaries_velocity_masks (special_boundaries.F:345-401)
__global__ void k_sb_masks(SihDev d, const int *__restrict__ maskV, const int *__restrict__ maskC)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.nVertices) d.solveVelocity[i] = maskV[i];
    if (i < d.nCells) d.solveStress[i] = maskC[i];
}

// Fused vertex kernel: outer sum and division of seaice_stress_divergence_variational
// (variational.F:1139-1178), ocean_stress_coefficient (velocity_solver.F:2986-3082) and solve_velocity
// (:3096-3208) or solve_velocity_revised (:3222-3342).
// PREDIV: the stress divergence was computed by another kernel (weak scheme) and is read from divU / divV
template <bool LAST, bool PREDIV>
__global__ void __launch_bounds__(256) k_sub_vertices(SihDev d, const int *__restrict__ list, int nList)
{
    int v = xcdChunkedBlock() * blockDim.x + threadIdx.x;
    if (list) {
        if (v >= nList) return;
        v = list[v];
    } else if (v >= d.nVertices) {
        return;
    }
    // (the places of the vertex's partial sums are requested with the flags that decide whether it is solved -- the
    // planes are padded, every address is valid --: the sums themselves then travel together with everything else the
    // solve reads, two memory round trips a thread instead of three)
    const int PV = d.padV;
    int g3[3] = {0, 0, 0};
    if (!PREDIV && d.D == 3) {
#pragma unroll
        for (int k = 0; k < 3; k++) g3[k] = d.gidx[k * PV + v];
    }
    const int solvedFlag = d.solveVelocity[v];
    if (d.haloVertices) {  // halo vertices with all three cells in the block are updated redundantly
        if (!d.vActive[v]) return;
    } else if (v >= d.nVerticesSolve) {
        return;
    }
    if (solvedFlag != 1) {
        if (LAST && !d.useOcean) d.ocnCoeff[v] = 0.0;
        return;
    }
    double dU = 0.0, dV = 0.0;
    if (PREDIV) {
        dU = d.divU[v];
        dV = d.divV[v];
    } else {
        // the partial sums of the vertex's cells in cellsOnVertex order: three, or four on a quadrilateral mesh (a
        // uniform branch, so that the common case keeps its three unconditional loads)
        if (d.D == 3) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                dU = dU + d.partU[g3[k]];
                dV = dV + d.partV[g3[k]];
            }
        } else {
            for (int k = 0; k < d.D; k++) {
                int g = d.gidx[k * PV + v];
                dU = dU + d.partU[g];
                dV = dV + d.partV[g];
            }
        }
        const double den0 = d.varDenom[v];
        dU = dU / den0;
        dV = dV / den0;
    }
    double uu = d.u[v], vv = d.v[v];
    const double uo = d.uOcnVert[v], vo = d.vOcnVert[v];
    double coeff = 0.0;
    if (d.useOcean) {
        if (d.quadratic) {
            double du = uo - uu, dv = vo - vv;
            coeff = DRAGIO * RHOW * d.iceAreaVertex[v] * sqrt(du * du + dv * dv);
        } else {
            coeff = DRAGIO * RHOW * d.iceAreaVertex[v];
        }
    }
    const double m = d.totalMassVertex[v], mf = d.massF[v];
    // oceanStressU / V of a solved vertex are the ocean's velocity itself: u cos(0) - v sin(0) sign(f) (:1878-1881 with
    // the turning angle of :61-63), so `uo`, `vo` stand for them; without ocean stress the coefficient is zero
    double l11, l22, r1, r2;
    const double l12 = -mf - coeff * SIN_TURN * sign1(mf);
    const double l21 = mf + coeff * SIN_TURN * sign1(mf);
    if (d.relation == 2) {
        const double brlx = d.brlx, dtD = d.dtDyn;
        l11 = (brlx + 1.0) * (m / dtD) + coeff * COS_TURN;
        l22 = (brlx + 1.0) * (m / dtD) + coeff * COS_TURN;
        r1 = dU + d.airTiltU[v] + coeff * uo + (m * (brlx * uu + d.uInit[v])) / dtD;
        r2 = dV + d.airTiltV[v] + coeff * vo + (m * (brlx * vv + d.vInit[v])) / dtD;
    } else {
        const double dtE = d.dtE;
        l11 = m / dtE + coeff * COS_TURN;
        l22 = m / dtE + coeff * COS_TURN;
        r1 = dU + d.airTiltU[v] + coeff * uo + (m * uu) / dtE;
        r2 = dV + d.airTiltV[v] + coeff * vo + (m * vv) / dtE;
    }
    if (d.relation != 3) {  // the linear relation has no velocity solve (velocity_solver.F:2529-2541)
        const double den = l11 * l22 - l12 * l21;
        d.u[v] = (l22 * r1 - l12 * r2) / den;
        d.v[v] = (l11 * r2 - l21 * r1) / den;
    }
    if (LAST) {
        d.divU[v] = dU;
        d.divV[v] = dV;
        d.ocnCoeff[v] = coeff;
    }
}

// The outer part of seaice_stress_divergence_variational (variational.F:1135-1178) on its own: owned vertices
// with solveVelocity = 1, the three partial sums in cellsOnVertex order, the division by the denominator.
__global__ void __launch_bounds__(256) k_div_vertices(SihDev d)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= d.nVerticesSolve || d.solveVelocity[v] != 1) return;
    const int PV = d.padV;
    double dU = 0.0, dV = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (k < d.D) {
            const int g = d.gidx[k * PV + v];
            dU = dU + d.partU[g];
            dV = dV + d.partV[g];
        }
    }
    const double den0 = d.varDenom[v];
    d.divU[v] = dU / den0;
    d.divV[v] = dV / den0;
}

// =========================================================================== post-subcycle kernels

// seaice_final_divergence_shear_variational (variational.F:1198-1330) and principal_stresses
// (velocity_solver.F:3534-3610)
__global__ void k_post0_cells(SihDev d)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    const int E = d.nEdgesOnCell[c];
    const int P = d.padC;
    double div = 0.0, shr = 0.0, dAvg = 0.0;
    const bool solve = d.solveStress[c] == 1;
    if (solve) {
        double dS = 0.0, tS = 0.0, sS = 0.0, dA = 0.0;
        for (int i = 0; i < E; i++) {
            size_t k = (size_t)i * P + c;
            double e11 = d.strain11[k], e22 = d.strain22[k], e12 = d.strain12[k];
            double sd = e11 + e22, st = e11 - e22, ss = e12 * 2.0;
            double Delta = sqrt(sd * sd + (st * st + ss * ss) / ECC2);
            dS = dS + sd;
            tS = tS + st;
            sS = sS + ss;
            dA = dA + Delta;
        }
        double en = (double)E;
        div = dS / en;
        shr = sqrt(tS * tS + sS * sS) / en;
        dAvg = dA / en;
    }
    if (d.useColumn) {
        if (solve) {
            d.ridgeConv[c] = -fmin(div, 0.0);
            d.ridgeShear[c] = 0.5 * (dAvg - fabs(div));
        } else {
            d.ridgeConv[c] = 0.0;
            d.ridgeShear[c] = 0.0;
        }
    }
    d.divergence[c] = div * 100.0 * 86400.0;
    d.shear[c] = shr * 100.0 * 86400.0;
    if (c < d.nCellsSolve) {
        for (int i = 0; i < E; i++) {
            size_t k = (size_t)i * P + c;
            double rp = d.Prep[k];
            if (rp > PUNY) {
                double a11 = d.stress11[k], a22 = d.stress22[k], a12 = d.stress12[k];
                double sq = (a11 + a22) * (a11 + a22) - 4.0 * a11 * a22 + 4.0 * (a12 * a12);
                double p1 = 0.5 * (a11 + a22) + 0.5 * sqrt(sq);
                double p2 = 0.5 * (a11 + a22) - 0.5 * sqrt(sq);
                d.prin1[k] = p1 / rp;
                d.prin2[k] = p2 / rp;
            } else {
                d.prin1[k] = 1.0e30;
                d.prin2[k] = 1.0e30;
            }
        }
    }
}

// ocean_stress_final, vertex part: coefficient refresh (:3674) and :3709-3732
__global__ void k_post0_vertices(SihDev d)
{
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= d.nVertices) return;
    if (!d.useOcean) {
        d.ocnCoeff[v] = 0.0;
        d.ocnStressU[v] = 0.0;
        d.ocnStressV[v] = 0.0;
        return;
    }
    if (!d.vActive[v]) return;
    if (d.solveVelocity[v] == 1) {
        double uo = d.uOcnVert[v], vo = d.vOcnVert[v], uu = d.u[v], vv = d.v[v], f = d.fVertex[v];
        double coeff;
        if (d.quadratic) {
            double du = uo - uu, dv = vo - vv;
            coeff = DRAGIO * RHOW * d.iceAreaVertex[v] * sqrt(du * du + dv * dv);
        } else {
            coeff = DRAGIO * RHOW * d.iceAreaVertex[v];
        }
        d.ocnCoeff[v] = coeff;
        double su = coeff * ((uo - uu) * COS_TURN - (vo - vv) * SIN_TURN * sign1(f));
        double sv = coeff * ((vo - vv) * COS_TURN + (uo - uu) * SIN_TURN * sign1(f));
        d.ocnStressU[v] = su / d.iceAreaVertex[v];
        d.ocnStressV[v] = sv / d.iceAreaVertex[v];
    } else {
        d.ocnStressU[v] = 0.0;
        d.ocnStressV[v] = 0.0;
    }
}

// seaice_interpolate_vertex_to_cell, mesh.F:2906-2976
__global__ void k_post1_cells(SihDev d)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    if (!d.useOcean) {
        d.ocnStressCellU[c] = 0.0;
        d.ocnStressCellV[c] = 0.0;
        return;
    }
    if (c >= d.nCellsSolve) return;
    const int E = d.nEdgesOnCell[c];
    double aU = 0.0, aV = 0.0, tot = 0.0;
    for (int i = 0; i < E; i++) {
        int v = d.voc[(size_t)i * d.padC + c];
        double w = (double)(float)d.interiorVertex[v];
        aU = aU + d.areaTriangle[v] * d.ocnStressU[v] * w;
        aV = aV + d.areaTriangle[v] * d.ocnStressV[v] * w;
        tot = tot + d.areaTriangle[v] * w;
    }
    if (tot > 0.0) {
        aU = aU / tot;
        aV = aV / tot;
    }
    d.ocnStressCellU[c] = aU;
    d.ocnStressCellV[c] = aV;
}

// velocity_solver.F:3816-3825
__global__ void k_post1_vertices(SihDev d)
{
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= d.nVertices || !d.useOcean || !d.vActive[v]) return;
    if (d.solveVelocity[v] == 1) {
        d.ocnStressU[v] = d.ocnStressU[v] * d.iceAreaVertex[v];
        d.ocnStressV[v] = d.ocnStressV[v] * d.iceAreaVertex[v];
    }
}

// =========================================================================== launchers
static inline dim3 gridFor(int n) { return dim3((unsigned)((n + 255) / 256)); }

static int checkInit(sih_ctx *c, const char *who)
{
    if (!c) {
        sih_set_error(std::string(who) + ": null context");
        return SIH_ERR_ARG;
    }
    if (!c->inited) {
        sih_set_error(std::string(who) + ": sih_evp_init has not been called");
        return SIH_ERR_STATE;
    }
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return sih_hip_fail(e, "hipSetDevice", __FILE__, __LINE__);
    return SIH_OK;
}

extern "C" int sih_evp_pre_subcycle(sih_ctx *c, int stage)
{
    int rc = checkInit(c, "sih_evp_pre_subcycle");
    if (rc) return rc;
    if (stage < -1 || stage > 4) {
        sih_set_error("sih_evp_pre_subcycle: stage must be -1..4");
        return SIH_ERR_ARG;
    }
    const SihDev &d = c->dev;
    dim3 b(256), gc = gridFor(c->nCells), gv = gridFor(c->nVertices);
    hipStream_t s = c->stream;
    if (stage == -1 || stage == 0) hipLaunchKernelGGL(k_pre0_cells, gc, b, 0, s, d);
    if (stage == -1 || stage == 1) {
        hipLaunchKernelGGL(k_pre1_vertices, gv, b, 0, s, d, c->cfg.calcVelocityMasks);
        if (c->cfg.calcVelocityMasks) hipLaunchKernelGGL(k_pre1_cells, gc, b, 0, s, d);
    }
    if (stage == -1 || stage == 2) hipLaunchKernelGGL(k_pre2_vertices, gv, b, 0, s, d);
    if (stage == -1 || stage == 3) hipLaunchKernelGGL(k_pre3_cells, gc, b, 0, s, d);
    if (stage == -1 || stage == 4) {
        hipLaunchKernelGGL(k_pre4_vertices, gv, b, 0, s, d);
        hipLaunchKernelGGL(k_pre4_cells, gc, b, 0, s, d);
        if (c->cfg.stressDivergenceScheme == 1) sih_weak_launch_pre(c);
    }
    if (stage == -1 || stage == 4) c->sbPending = c->sbVelocity || c->sbMasks;
    HIPCHK(hipGetLastError());
    return SIH_OK;
}

template <int M>
static void launchCells(sih_ctx *c, bool last, const int *list, int nList, int mode)
{
    dim3 b(256), g = gridFor(list ? nList : c->nCells);
    const SihDev &d = c->dev;
    if (mode == 0 && c->splitCells) {
        const int n = list ? nList : c->nCells;
        dim3 bs(64, M), gs((unsigned)((n + 63) / 64));
        const bool nt = d.streamHints != 0;
#define SIH_LAUNCH_SPLIT(SP, LA)                                                                                        \
    do {                                                                                                                \
        if (d.streamHints >= 2)                                                                                         \
            hipLaunchKernelGGL((k_sub_cells_split<M, SP, LA, 2>), gs, bs, 0, c->stream, d, list, nList);                \
        else if (nt)                                                                                                    \
            hipLaunchKernelGGL((k_sub_cells_split<M, SP, LA, 1>), gs, bs, 0, c->stream, d, list, nList);                \
        else                                                                                                            \
            hipLaunchKernelGGL((k_sub_cells_split<M, SP, LA, 0>), gs, bs, 0, c->stream, d, list, nList);                \
    } while (0)
        if (c->sparseGrad) {
            if (last)
                SIH_LAUNCH_SPLIT(true, true);
            else
                SIH_LAUNCH_SPLIT(true, false);
        } else {
            if (last)
                SIH_LAUNCH_SPLIT(false, true);
            else
                SIH_LAUNCH_SPLIT(false, false);
        }
#undef SIH_LAUNCH_SPLIT
        return;
    }
    if (mode == 1) {
        hipLaunchKernelGGL((k_sub_cells<M, false, false, 1>), g, b, 0, c->stream, d, list, nList);
    } else if (mode == 3) {
        hipLaunchKernelGGL((k_sub_cells<M, false, false, 3>), g, b, 0, c->stream, d, list, nList);
    } else if (mode == 2) {
        if (last)
            hipLaunchKernelGGL((k_sub_cells<M, false, true, 2>), g, b, 0, c->stream, d, list, nList);
        else
            hipLaunchKernelGGL((k_sub_cells<M, false, false, 2>), g, b, 0, c->stream, d, list, nList);
    } else if (c->sparseGrad) {
        if (last)
            hipLaunchKernelGGL((k_sub_cells<M, true, true, 0>), g, b, 0, c->stream, d, list, nList);
        else
            hipLaunchKernelGGL((k_sub_cells<M, true, false, 0>), g, b, 0, c->stream, d, list, nList);
    } else {
        if (last)
            hipLaunchKernelGGL((k_sub_cells<M, false, true, 0>), g, b, 0, c->stream, d, list, nList);
        else
            hipLaunchKernelGGL((k_sub_cells<M, false, false, 0>), g, b, 0, c->stream, d, list, nList);
    }
}

static void launchCellsM(sih_ctx *c, bool last, const int *list, int nList, int mode)
{
    switch (c->M) {
        case 3: launchCells<3>(c, last, list, nList, mode); break;
        case 4: launchCells<4>(c, last, list, nList, mode); break;
        case 5: launchCells<5>(c, last, list, nList, mode); break;
        case 6: launchCells<6>(c, last, list, nList, mode); break;
        case 7: launchCells<7>(c, last, list, nList, mode); break;
        case 8: launchCells<8>(c, last, list, nList, mode); break;
        case 9: launchCells<9>(c, last, list, nList, mode); break;
        default: launchCells<10>(c, last, list, nList, mode); break;
    }
}

// subset >= 0: only the entities of c->d_part[subset] (nothing to do when that subset is empty)
static void launchCellKernel(sih_ctx *c, bool last, int subset = -1)
{
    const int *list = subset >= 0 ? c->d_part[subset] : nullptr;
    const int nList = subset >= 0 ? c->nPart[subset] : 0;
    if (subset >= 0 && nList == 0) return;
    if (c->cfg.strainScheme == 1 && c->cfg.stressDivergenceScheme == 1) {
        sih_weak_launch_cells(c);  // weak scheme: strain + stress per cell, then the divergence at every vertex
        return;
    }
    if (c->cfg.strainScheme == 1) {  // weak strain interpolated to the corners, variational from there on
        sih_weak_launch_hybrid_strain(c);
        launchCellsM(c, last, nullptr, 0, 2);
        return;
    }
    if (c->cfg.averageVariationalStrain) {  // whole block only (checked where subsets are set up)
        launchCellsM(c, last, nullptr, 0, 1);
        hipLaunchKernelGGL(k_avg_strain_vertices, gridFor(c->nVerticesSolve), dim3(256), 0, c->stream, c->dev);
        launchCellsM(c, last, nullptr, 0, 2);
        return;
    }
    launchCellsM(c, last, list, nList, 0);
}

static void applySpecialBoundaryVelocity(sih_ctx *c)
{
    if (!c->sbVelocity || c->nSb == 0) return;
    dim3 g = gridFor(c->nSb), b(256);
    hipLaunchKernelGGL(k_sb_gather, g, b, 0, c->stream, c->nSb, c->d_sbOrigin, c->d_sbSign, c->dev.u, c->dev.v, c->d_sbTmp);
    hipLaunchKernelGGL(k_sb_scatter, g, b, 0, c->stream, c->nSb, c->d_sbIdx, c->d_sbTmp, c->dev.u, c->dev.v);
}

// what subcycle_velocity_solver does before its loop (velocity_solver.F:2441-2443); the masks do not change
// during the loop, so setting them once equals the reference's repetition after every iteration
static void beginSubcycles(sih_ctx *c)
{
    if (!c->sbPending) return;
    c->sbPending = false;
    applySpecialBoundaryVelocity(c);
    if (c->sbMasks)
        hipLaunchKernelGGL(k_sb_masks, gridFor(std::max(c->nCells, c->nVertices)), dim3(256), 0, c->stream, c->dev,
                           c->d_sbMaskV, c->d_sbMaskC);
}

static void launchVertexKernel(sih_ctx *c, bool last, int subset = -1)
{
    const int *list = subset >= 0 ? c->d_part[subset] : nullptr;
    const int nList = subset >= 0 ? c->nPart[subset] : 0;
    if (subset >= 0 && nList == 0) return;
    dim3 b(256), g = gridFor(list ? nList : (c->cfg.computeHaloVertices ? c->nVertices : c->nVerticesSolve));
    if (c->cfg.stressDivergenceScheme == 1) {
        if (last)
            hipLaunchKernelGGL((k_sub_vertices<true, true>), g, b, 0, c->stream, c->dev, list, nList);
        else
            hipLaunchKernelGGL((k_sub_vertices<false, true>), g, b, 0, c->stream, c->dev, list, nList);
    } else if (last)
        hipLaunchKernelGGL((k_sub_vertices<true, false>), g, b, 0, c->stream, c->dev, list, nList);
    else
        hipLaunchKernelGGL((k_sub_vertices<false, false>), g, b, 0, c->stream, c->dev, list, nList);
    if (!c->sbDeferred) applySpecialBoundaryVelocity(c);  // velocity_solver.F:2454-2456
}

// Several blocks: the reference applies the special boundaries after the halo update that ends an iteration
// (velocity_solver.F:2449-2456), so the host switches the application inside the iteration off and asks for it
// after its exchange.
extern "C" int sih_evp_defer_special_boundaries(sih_ctx *c, int on)
{
    int rc = checkInit(c, "sih_evp_defer_special_boundaries");
    if (rc) return rc;
    c->sbDeferred = on != 0;
    if (c->graphExec) {  // a captured graph holds the launches of the other mode
        (void)hipGraphExecDestroy(c->graphExec);
        c->graphExec = nullptr;
        c->graphN = 0;
    }
    return SIH_OK;
}

extern "C" int sih_evp_apply_special_boundaries(sih_ctx *c)
{
    int rc = checkInit(c, "sih_evp_apply_special_boundaries");
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    applySpecialBoundaryVelocity(c);
    HIPCHK(hipGetLastError());
    return SIH_OK;
}

// Blocks that qualify run loops of two or more iterations in the persistent kernel (evp_persist.hip): a launch of
// it costs about as much as one iteration of the two-kernel form before the first iteration starts.
// Only while this context is the only one of the process on its device: the kernel needs all its workgroups resident
// together, and the launches of other blocks (each on its own stream) would take their places.
// sih_evp_set_persistent chooses per context; without it the environment does: SIH_PERSISTENT=0 keeps the two-kernel
// form everywhere.  The halo exchange inside the kernel (mode 2) is never chosen by default: the blocks that
// exchange with each other must all run it or none, which only their host can know (SIH_PERSIST_EXCHANGE=1 asks
// for it from the environment, for runs whose blocks are known to qualify alike).
static int persistentMode(sih_ctx *c)
{
    static const int fromEnv = [] {
        const char *e = getenv("SIH_PERSISTENT");
        if (e && atoi(e) == 0) return 0;
        const char *x = getenv("SIH_PERSIST_EXCHANGE");
        return (x && atoi(x) != 0) ? 2 : 1;
    }();
    return c->persistMode >= 0 ? c->persistMode : fromEnv;
}

static bool fusedAllowed()
{
    static const bool on = [] {
        const char *e = getenv("SIH_FUSED");
        return !(e && atoi(e) == 0);
    }();
    return on;
}

static bool usePersistent(sih_ctx *c, int iterations)
{
    // a launch costs 14 to 25 us before its first iteration (396 to 1 010 workgroups check in): against the two-kernel
    // form that is won back with the second iteration, against the fused kernel with the fourth
    const int least = (fusedAllowed() && sih_fused_usable(c)) ? 4 : 2;
    return persistentMode(c) >= 1 && iterations >= least && sih_persist_usable(c) && sih_live_contexts(c->device) == 1;
}

// Where the persistent kernel does not apply (block too large to be resident, or not alone on its device), loops of
// two or more iterations run as one launch per iteration over the same tiles (k_sub_fused, evp_persist.hip) up to
// the size where the two-kernel form, bound by memory bandwidth there, is as fast: SIH_FUSED_MAX_CELLS (default
// below), SIH_FUSED=0 never, SIH_FUSED=1 at any size.
static bool useFused(sih_ctx *c, int iterations)
{
    static const int fromEnv = [] {
        const char *e = getenv("SIH_FUSED");
        return e ? atoi(e) : -1;
    }();
    static const int maxCells = [] {
        const char *e = getenv("SIH_FUSED_MAX_CELLS");
        return e ? atoi(e) : SIH_FUSED_MAX_CELLS_DEFAULT;
    }();
    if (fromEnv == 0 || persistentMode(c) == 0 || iterations < 2 || usePersistent(c, iterations) || !sih_fused_usable(c))
        return false;
    return fromEnv == 1 || c->nCells <= maxCells;
}

// which of the three forms queueIterations takes: part of the key of a captured graph
static int iterationForm(sih_ctx *c, int m)
{
    c->lastForm = usePersistent(c, m) ? 1 : useFused(c, m) ? 2 : 0;  // sih_get_info "iterationForm"
    return c->lastForm;
}

// m iterations (none the last of the dynamics step) queued on the context's stream in the form that suits the block
static int queueIterations(sih_ctx *c, int m)
{
    const int form = iterationForm(c, m);
    if (form == 1) return sih_persist_launch(c, m);
    if (form == 2) return sih_fused_launch(c, m);
    for (int it = 0; it < m; it++) {
        launchCellKernel(c, false);
        launchVertexKernel(c, false);
    }
    return SIH_OK;
}

extern "C" int sih_evp_set_persistent(sih_ctx *c, int mode)
{
    if (!c || mode < -1 || mode > 2) {
        sih_set_error("sih_evp_set_persistent: mode must be -1 (environment), 0, 1 or 2");
        return SIH_ERR_ARG;
    }
    c->persistMode = mode;
    return SIH_OK;
}

extern "C" int sih_evp_subcycle(sih_ctx *c, int n, int last)
{
    int rc = checkInit(c, "sih_evp_subcycle");
    if (rc) return rc;
    if (n < 0) {
        sih_set_error("sih_evp_subcycle: n must be >= 0");
        return SIH_ERR_ARG;
    }
    beginSubcycles(c);
    for (int it = 0; it < n; it++) {
        bool l = last && it == n - 1;
        launchCellKernel(c, l);
        launchVertexKernel(c, l);
    }
    HIPCHK(hipGetLastError());
    return SIH_OK;
}

// The reference's two public operators on their own, as its operator test cases drive them.
extern "C" int sih_evp_strain_tensor(sih_ctx *c)
{
    int rc = checkInit(c, "sih_evp_strain_tensor");
    if (rc) return rc;
    if (c->cfg.strainScheme != 0) {
        sih_set_error("sih_evp_strain_tensor: variational scheme only");
        return SIH_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(c->device));
    launchCellsM(c, false, nullptr, 0, 1);
    HIPCHK(hipGetLastError());
    return SIH_OK;
}

extern "C" int sih_evp_stress_divergence(sih_ctx *c)
{
    int rc = checkInit(c, "sih_evp_stress_divergence");
    if (rc) return rc;
    if (c->cfg.stressDivergenceScheme != 0) {
        sih_set_error("sih_evp_stress_divergence: variational scheme only");
        return SIH_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(c->device));
    launchCellsM(c, false, nullptr, 0, 3);
    hipLaunchKernelGGL(k_div_vertices, gridFor(c->nVerticesSolve), dim3(256), 0, c->stream, c->dev);
    HIPCHK(hipGetLastError());
    return SIH_OK;
}

extern "C" int sih_evp_post_subcycle(sih_ctx *c, int stage)
{
    int rc = checkInit(c, "sih_evp_post_subcycle");
    if (rc) return rc;
    if (stage < -1 || stage > 1) {
        sih_set_error("sih_evp_post_subcycle: stage must be -1..1");
        return SIH_ERR_ARG;
    }
    const SihDev &d = c->dev;
    dim3 b(256), gc = gridFor(c->nCells), gv = gridFor(c->nVertices);
    if (stage == -1 || stage == 0) {
        if (c->cfg.stressDivergenceScheme == 1)
            sih_weak_launch_post(c);  // velocity_solver.F:3416-3418, 3497-3516
        else
            hipLaunchKernelGGL(k_post0_cells, gc, b, 0, c->stream, d);
        hipLaunchKernelGGL(k_post0_vertices, gv, b, 0, c->stream, d);
    }
    if (stage == -1 || stage == 1) {
        hipLaunchKernelGGL(k_post1_cells, gc, b, 0, c->stream, d);
        hipLaunchKernelGGL(k_post1_vertices, gv, b, 0, c->stream, d);
    }
    HIPCHK(hipGetLastError());
    return SIH_OK;
}

extern "C" int sih_fields_get_start(sih_ctx *c, int nf, const char *const *names, void *const *hosts, const size_t *counts);
extern "C" int sih_fields_get_finish(sih_ctx *c);

static int evpRun(sih_ctx *c, int nEarly, const char *const *names, void *const *hosts, const size_t *counts)
{
    int rc = checkInit(c, "sih_evp_run");
    if (rc) return rc;
    if ((rc = sih_evp_pre_subcycle(c, -1))) return rc;
    // what the pre-cycle leaves for good starts its way to the host now and travels while the iterations run
    if (nEarly > 0 && (rc = sih_fields_get_start(c, nEarly, names, hosts, counts))) return rc;
    // All iterations but the last are the same pair (or few) of launches: replayed from a graph captured at the
    // first call, which takes the launch gaps out (2 % of the loop at 164 k cells, 25 % at 10 k).  The last
    // iteration stores more and runs as it is.  SIH_EVP_GRAPH=0 keeps every launch separate.
    static const bool useGraph = [] {
        const char *e = getenv("SIH_EVP_GRAPH");
        return !(e && atoi(e) == 0);
    }();
    const int n = c->cfg.elasticSubcycleNumber;
    if (useGraph && n > 2) {
        if ((rc = sih_evp_subcycle_graph(c, n - 1))) return rc;
        if ((rc = sih_evp_subcycle(c, 1, 1))) return rc;
    } else if ((rc = sih_evp_subcycle(c, n, 1)))
        return rc;
    if ((rc = sih_evp_post_subcycle(c, -1))) return rc;
    // everything is queued: the early outputs are finished (their last hop into ordinary host memory is made by this
    // thread) while the device still runs the iterations
    if (nEarly > 0 && (rc = sih_fields_get_finish(c))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return sih_persist_status(c);
}

extern "C" int sih_evp_run(sih_ctx *c) { return evpRun(c, 0, nullptr, nullptr, nullptr); }

// sih_evp_run with a download inside it: the listed fields, which must be among those the pre-cycle computes and the
// iterations and the post-cycle leave alone (masks, ice strength, forcing terms, aggregated mass and area, the
// velocities at the start of the step), are read right after the pre-cycle and copied to the host while the
// iterations run (sih_fields_get_start; complete at sih_fields_get_finish or at the next sih_fields_get_start).
extern "C" int sih_evp_run_early_outputs(sih_ctx *c, int nFields, const char *const *names, void *const *hosts,
                                         const size_t *counts)
{
    return evpRun(c, nFields, names, hosts, counts);
}

static int ensureEvents(sih_ctx *c, size_t n)
{
    while (c->events.size() < n) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        c->events.push_back(e);
    }
    return SIH_OK;
}

// n iterations (none of them the last of the dynamics step) replayed from a captured HIP graph; asynchronous
extern "C" int sih_evp_subcycle_graph(sih_ctx *c, int n)
{
    int rc = checkInit(c, "sih_evp_subcycle_graph");
    if (rc) return rc;
    if (n < 1) {
        sih_set_error("sih_evp_subcycle_graph: n must be >= 1");
        return SIH_ERR_ARG;
    }
    beginSubcycles(c);
    const int form = iterationForm(c, n);
    if (form == 1) return sih_persist_launch(c, n);  // small block: the whole loop as one resident launch
    if (!c->graphExec || c->graphN != n || c->graphForm != form) {
        if (c->graphExec) {
            (void)hipGraphExecDestroy(c->graphExec);
            c->graphExec = nullptr;
        }
        hipGraph_t graph;
        HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        rc = queueIterations(c, n);
        hipError_t e = hipStreamEndCapture(c->stream, &graph);
        if (rc) return rc;
        HIPCHK(e);
        HIPCHK(hipGraphInstantiate(&c->graphExec, graph, nullptr, nullptr, 0));
        HIPCHK(hipGraphDestroy(graph));
        c->graphN = n;
        c->graphForm = form;
    }
    HIPCHK(hipGraphLaunch(c->graphExec, c->stream));
    return SIH_OK;
}

// n iterations (none the last of the dynamics step) in the persistent kernel; an error if the block does not
// qualify for it (sih_get_info "persistentKernel")
extern "C" int sih_evp_subcycle_persistent(sih_ctx *c, int n)
{
    int rc = checkInit(c, "sih_evp_subcycle_persistent");
    if (rc) return rc;
    if (n < 1) {
        sih_set_error("sih_evp_subcycle_persistent: n must be >= 1");
        return SIH_ERR_ARG;
    }
    if (!sih_persist_usable(c)) {
        sih_set_error("sih_evp_subcycle_persistent: this block does not qualify (too many cells to stay resident, "
                      "or an option that needs the two-kernel form)");
        return SIH_ERR_UNSUPPORTED;
    }
    beginSubcycles(c);
    return sih_persist_launch(c, n);
}

// n iterations (none the last of the dynamics step) as n + 1 launches of the fused kernel; an error if the block
// does not qualify for it (sih_get_info "fusedKernel")
extern "C" int sih_evp_subcycle_fused(sih_ctx *c, int n)
{
    int rc = checkInit(c, "sih_evp_subcycle_fused");
    if (rc) return rc;
    if (n < 1) {
        sih_set_error("sih_evp_subcycle_fused: n must be >= 1");
        return SIH_ERR_ARG;
    }
    if (!sih_fused_usable(c)) {
        sih_set_error("sih_evp_subcycle_fused: this block does not qualify (an option that needs the two-kernel form)");
        return SIH_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(c->device));
    beginSubcycles(c);
    return sih_fused_launch(c, n);
}

// ---------------------------------------------------------------------------- special boundaries
static int uploadIntVec(int **dst, const std::vector<int> &v)
{
    if (*dst) {
        HIPCHK(hipFree(*dst));
        *dst = nullptr;
    }
    if (!v.empty()) {
        HIPCHK(hipMalloc((void **)dst, sizeof(int) * v.size()));
        HIPCHK(hipMemcpy(*dst, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice));
    }
    return SIH_OK;
}

extern "C" int sih_evp_set_special_boundaries(sih_ctx *c, const int32_t *type, const int32_t *source,
                                              const int32_t *maskV, const int32_t *maskC)
{
    int rc = checkInit(c, "sih_evp_set_special_boundaries");
    if (rc) return rc;
    const bool wantVel = c->cfg.useSpecialBoundariesVelocity != 0, wantMask = c->cfg.useSpecialBoundariesVelocityMasks != 0;
    if ((wantVel && (!type || !source)) || (wantMask && (!maskV || !maskC))) {
        sih_set_error("sih_evp_set_special_boundaries: an array the configuration asks for is missing");
        return SIH_ERR_ARG;
    }
    if (wantVel && c->partsReady) {
        sih_set_error("sih_evp_set_special_boundaries: not together with the exchange-overlap subsets");
        return SIH_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    const int nV = c->nVertices, nC = c->nCells;
    c->sbVelocity = false;
    c->sbMasks = false;
    if (wantVel) {
        // the reference loop in its own (host) order, carried out on (sign, origin) pairs instead of values
        std::vector<int> org(nV), sg(nV, 1);
        for (int h = 0; h < nV; h++) org[h] = h;
        std::vector<int> marked;
        for (int h = 0; h < nV; h++) {
            const int t = type[h];
            if (t == 1 || t == 2) {
                const int s = source[h];
                if (s == 0) continue;  // source outside this block: the vertex keeps what the exchange brought
                if (s < 1 || s > nV) {
                    sih_set_error("sih_evp_set_special_boundaries: vertexBoundarySourceLocal out of range");
                    return SIH_ERR_ARG;
                }
                org[h] = org[s - 1];
                sg[h] = (t == 1) ? sg[s - 1] : -sg[s - 1];
                marked.push_back(h);
            } else if (t == 3) {
                sg[h] = 0;
                marked.push_back(h);
            } else if (t != 0) {
                sih_set_error("sih_evp_set_special_boundaries: unknown vertexBoundaryType");
                return SIH_ERR_ARG;
            }
        }
        std::vector<int> idx, origin, sign;
        for (int h : marked) {
            idx.push_back(c->vertInv[h]);
            origin.push_back(c->vertInv[org[h]]);
            sign.push_back(sg[h]);
        }
        if ((rc = uploadIntVec(&c->d_sbIdx, idx))) return rc;
        if ((rc = uploadIntVec(&c->d_sbOrigin, origin))) return rc;
        if ((rc = uploadIntVec(&c->d_sbSign, sign))) return rc;
        if (c->d_sbTmp) {
            HIPCHK(hipFree(c->d_sbTmp));
            c->d_sbTmp = nullptr;
        }
        c->nSb = (int)idx.size();
        if (c->nSb) HIPCHK(hipMalloc((void **)&c->d_sbTmp, sizeof(double) * 2 * c->nSb));
        c->sbVelocity = true;
    }
    if (wantMask) {
        std::vector<int> mv(nV), mc(nC);
        for (int i = 0; i < nV; i++) mv[i] = maskV[c->vertPerm[i]];
        for (int i = 0; i < nC; i++) mc[i] = maskC[c->cellPerm[i]];
        if ((rc = uploadIntVec(&c->d_sbMaskV, mv))) return rc;
        if ((rc = uploadIntVec(&c->d_sbMaskC, mc))) return rc;
        c->sbMasks = true;
    }
    // a captured graph would not hold the extra launches
    if (c->graphExec) {
        (void)hipGraphExecDestroy(c->graphExec);
        c->graphExec = nullptr;
        c->graphN = 0;
    }
    return SIH_OK;
}

// ---------------------------------------------------------------------------- exchange overlap
// Subsets of one block's entities, so that a host can start the halo exchange of a subcycle as soon as
// the vertices its neighbours need are done and keep the GPU busy with everything that needs no halo
// value until the message has arrived.  The subsets partition what the full kernels cover, and every
// entity is computed once per subcycle from the same inputs, so results do not change by a bit.
extern "C" int sih_evp_overlap_setup(sih_ctx *c)
{
    int rc = checkInit(c, "sih_evp_overlap_setup");
    if (rc) return rc;
    if (c->cfg.averageVariationalStrain || c->sbVelocity || c->cfg.strainScheme == 1) {
        sih_set_error("sih_evp_overlap_setup: strain averaging, special boundaries and the weak scheme need "
                      "whole-block loops");
        return SIH_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(c->device));
    const int nC = c->nCells, nV = c->nVertices;
    std::vector<int> active(nV, 0);
    if (c->cfg.computeHaloVertices) {
        SihField *f = sih_find_field(c, "_vertexActive");
        if (!f) {
            sih_set_error("sih_evp_overlap_setup: internal field missing");
            return SIH_ERR_STATE;
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpy(active.data(), f->d, sizeof(int) * nV, hipMemcpyDeviceToHost));
    } else {
        for (int v = 0; v < c->nVerticesSolve; v++) active[v] = 1;
    }
    std::vector<char> sendV(nV, 0), recvV(nV, 0), recvC(nC, 0);
    for (int v : c->hSend[1]) sendV[v] = 1;
    for (int v : c->hRecv[1]) recvV[v] = 1;
    for (int x : c->hRecv[0]) recvC[x] = 1;
    std::vector<int> part[4];
    for (int i = 0; i < nC; i++) {
        bool rim = recvC[i];
        for (int s = 0; s < c->hNe[i] && !rim; s++) {
            int v = c->hVoc[(size_t)s * c->padC + i];
            if (v < nV && recvV[v]) rim = true;
        }
        part[rim ? 1 : 0].push_back(i);
    }
    for (int v = 0; v < nV; v++)
        if (active[v]) part[sendV[v] ? 2 : 3].push_back(v);
    for (int k = 0; k < 4; k++) {
        if (c->d_part[k]) {
            HIPCHK(hipFree(c->d_part[k]));
            c->d_part[k] = nullptr;
        }
        c->nPart[k] = (int)part[k].size();
        if (!part[k].empty()) {
            HIPCHK(hipMalloc((void **)&c->d_part[k], sizeof(int) * part[k].size()));
            HIPCHK(hipMemcpy(c->d_part[k], part[k].data(), sizeof(int) * part[k].size(), hipMemcpyHostToDevice));
        }
    }
    for (auto &kv : c->partGraphs) (void)hipGraphExecDestroy(kv.second);
    c->partGraphs.clear();
    c->partsReady = true;
    return SIH_OK;
}

static int launchPart(sih_ctx *c, int part)
{
    switch (part) {
        case SIH_PART_CELLS_ALL: launchCellKernel(c, false); break;
        case SIH_PART_CELLS_INTERIOR: launchCellKernel(c, false, 0); break;
        case SIH_PART_CELLS_RIM: launchCellKernel(c, false, 1); break;
        case SIH_PART_VERTICES_ALL: launchVertexKernel(c, false); break;
        case SIH_PART_VERTICES_SEND: launchVertexKernel(c, false, 2); break;
        case SIH_PART_VERTICES_REST: launchVertexKernel(c, false, 3); break;
        default: sih_set_error("sih_evp_subcycle_parts: unknown part"); return SIH_ERR_ARG;
    }
    return SIH_OK;
}

extern "C" int sih_evp_subcycle_parts(sih_ctx *c, int nParts, const int32_t *parts, int useGraph)
{
    int rc = checkInit(c, "sih_evp_subcycle_parts");
    if (rc) return rc;
    if (nParts < 0 || (nParts && !parts)) {
        sih_set_error("sih_evp_subcycle_parts: bad argument");
        return SIH_ERR_ARG;
    }
    bool needSets = false;
    for (int k = 0; k < nParts; k++) {
        if (parts[k] < 0 || parts[k] > SIH_PART_VERTICES_REST) {
            sih_set_error("sih_evp_subcycle_parts: unknown part");
            return SIH_ERR_ARG;
        }
        if (parts[k] != SIH_PART_CELLS_ALL && parts[k] != SIH_PART_VERTICES_ALL) needSets = true;
    }
    if (needSets && !c->partsReady) {
        sih_set_error("sih_evp_subcycle_parts: call sih_evp_overlap_setup after sih_halo_set_lists first");
        return SIH_ERR_STATE;
    }
    if (nParts == 0) return SIH_OK;
    HIPCHK(hipSetDevice(c->device));
    beginSubcycles(c);
    if (!useGraph || nParts < 2) {
        for (int k = 0; k < nParts; k++)
            if ((rc = launchPart(c, parts[k]))) return rc;
        HIPCHK(hipGetLastError());
        return SIH_OK;
    }
    std::vector<int> key(parts, parts + nParts);
    key.push_back(c->sparseGrad ? 1 : 0);
    auto it = c->partGraphs.find(key);
    if (it == c->partGraphs.end()) {
        hipGraph_t graph;
        hipGraphExec_t exec;
        HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < nParts; k++) launchPart(c, parts[k]);
        HIPCHK(hipStreamEndCapture(c->stream, &graph));
        HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        HIPCHK(hipGraphDestroy(graph));
        it = c->partGraphs.emplace(key, exec).first;
    }
    HIPCHK(hipGraphLaunch(it->second, c->stream));
    return SIH_OK;
}

// ---------------------------------------------------------------------------- subcycles with the halo update inside
// One chunk of `m` iterations (none the last of the dynamics step) followed by the halo update of `names`, as a
// sequence of launches on the context's stream.  Plain order: (cells, vertices) x m, push, pull.  Overlapped order
// (the subsets of sih_evp_overlap_setup): the vertices the neighbours need are updated first and pushed from a
// second stream while this stream finishes the other vertices and, if more iterations follow, runs the next
// iteration's cell kernel on the cells that touch no halo vertex; the pull and the rim cells open the next chunk.
static int queueChunk(sih_ctx *c, int m, bool overlap, bool pendingIn, bool more, int nNames, const char *const *names)
{
    int rc;
    if (!overlap) {
        if ((rc = queueIterations(c, m))) return rc;
        if ((rc = sih_comm_push(c, nNames, names, c->stream))) return rc;
        if ((rc = sih_comm_pull(c, nNames, names, c->stream))) return rc;
        if (c->sbDeferred) applySpecialBoundaryVelocity(c);  // velocity_solver.F:2449-2456
        return SIH_OK;
    }
    int first = 0;
    if (pendingIn) {  // the first iteration of this chunk already has its interior cells
        if ((rc = sih_comm_pull(c, nNames, names, c->stream))) return rc;
        launchCellKernel(c, false, 1);
        if (m == 1) {
            launchVertexKernel(c, false, 2);
        } else {
            launchVertexKernel(c, false);
        }
        first = 1;
    }
    for (int it = first; it < m; it++) {
        if (pendingIn && m == 1) break;
        launchCellKernel(c, false);
        if (it == m - 1)
            launchVertexKernel(c, false, 2);
        else
            launchVertexKernel(c, false);
    }
    hipStream_t side = sih_comm_stream(c);
    hipEvent_t evFork, evJoin;
    sih_comm_events(c, &evFork, &evJoin);
    HIPCHK(hipEventRecord(evFork, c->stream));
    HIPCHK(hipStreamWaitEvent(side, evFork, 0));
    if ((rc = sih_comm_push(c, nNames, names, side))) return rc;
    HIPCHK(hipEventRecord(evJoin, side));
    launchVertexKernel(c, false, 3);
    // the interior cells of the next iteration rewrite their stresses: when the stresses travel with the
    // velocities (more than two fields), the push must have read them first
    const bool pushReadsStress = nNames > 2;
    if (pushReadsStress) HIPCHK(hipStreamWaitEvent(c->stream, evJoin, 0));
    if (more) launchCellKernel(c, false, 0);
    if (!pushReadsStress) HIPCHK(hipStreamWaitEvent(c->stream, evJoin, 0));
    return SIH_OK;
}

extern "C" int sih_evp_subcycle_exchange(sih_ctx *c, int n, int last, int interval, int overlap, int useGraph)
{
    int rc = checkInit(c, "sih_evp_subcycle_exchange");
    if (rc) return rc;
    if (n < 0 || interval < 1) {
        sih_set_error("sih_evp_subcycle_exchange: n must be >= 0 and interval >= 1");
        return SIH_ERR_ARG;
    }
    if (!sih_comm_ready(c)) {
        sih_set_error("sih_evp_subcycle_exchange: call sih_comm_create and sih_comm_connect first");
        return SIH_ERR_STATE;
    }
    if (interval > 1 && !c->cfg.computeHaloVertices) {
        sih_set_error("sih_evp_subcycle_exchange: an interval above 1 needs computeHaloVertices (and as many halo layers)");
        return SIH_ERR_ARG;
    }
    if (overlap && !c->partsReady) {
        sih_set_error("sih_evp_subcycle_exchange: call sih_evp_overlap_setup first");
        return SIH_ERR_STATE;
    }
    if (c->sbVelocity && (!c->sbDeferred || interval > 1 || overlap)) {
        sih_set_error("sih_evp_subcycle_exchange: with special velocity boundaries: deferred application, interval 1, no overlap");
        return SIH_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(c->device));
    // a block small enough for the persistent kernel gains more from it than from hiding the exchange behind the
    // pieces of the two-kernel form (8-way block of QU60: 8.4 us an iteration against 17.8)
    if (overlap && usePersistent(c, std::min(interval, last ? n - 1 : n))) overlap = 0;
    static const char *uv[2] = {"uVelocity", "vVelocity"};
    static const char *uvs[5] = {"uVelocity", "vVelocity", "stress11", "stress22", "stress12"};
    const char *const *names = interval == 1 ? uv : uvs;
    const int nNames = interval == 1 ? 2 : 5;
    beginSubcycles(c);
    const int nMain = last ? n - 1 : n;
    int done = 0;
    bool pending = false;
    // a small block whose transport allows it: all the iterations and the exchanges between them as one launch
    // (interval 1, the reference's order: the velocities after every iteration; deeper halos: velocities and
    // stresses after every interval-th)
    if (nMain >= 2 && persistentMode(c) == 2 && !c->sbVelocity && usePersistent(c, nMain) && sih_persist_exchange_usable(c)) {
        if ((rc = sih_persist_launch_exchange(c, nMain, interval))) return rc;
        c->lastForm = 1;
        done = nMain;
    }
    while (done < nMain) {
        const int m = std::min(interval, nMain - done);
        const bool more = done + m < nMain;
        if (useGraph) {
            std::vector<int> key = {-1000, m, overlap ? 1 : 0, pending ? 1 : 0, more ? 1 : 0, c->sparseGrad ? 1 : 0,
                                    overlap ? 0 : iterationForm(c, m)};
            auto it = c->partGraphs.find(key);
            if (it == c->partGraphs.end()) {
                hipGraph_t graph;
                hipGraphExec_t exec;
                // the tables of the field group are built (and copied to the device) before the capture begins
                if ((rc = sih_comm_prepare(c, nNames, names))) return rc;
                HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                rc = queueChunk(c, m, overlap != 0, pending, more, nNames, names);
                hipError_t e = hipStreamEndCapture(c->stream, &graph);
                if (rc) return rc;
                HIPCHK(e);
                HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                HIPCHK(hipGraphDestroy(graph));
                it = c->partGraphs.emplace(key, exec).first;
            }
            HIPCHK(hipGraphLaunch(it->second, c->stream));
        } else if ((rc = queueChunk(c, m, overlap != 0, pending, more, nNames, names))) {
            return rc;
        }
        pending = overlap != 0;
        done += m;
    }
    if (pending && (rc = sih_comm_pull(c, nNames, names, c->stream))) return rc;
    if (last && n > 0) {
        launchCellKernel(c, true);
        launchVertexKernel(c, true);
        if ((rc = sih_comm_push(c, nNames, names, c->stream))) return rc;
        if ((rc = sih_comm_pull(c, nNames, names, c->stream))) return rc;
        if (c->sbDeferred) applySpecialBoundaryVelocity(c);
    }
    HIPCHK(hipGetLastError());
    return SIH_OK;
}

extern "C" int sih_evp_subcycle_timed(sih_ctx *c, int n, int useGraph, int sampleEvery, double *totalMs, double *cellMs,
                                      double *vertexMs)
{
    int rc = checkInit(c, "sih_evp_subcycle_timed");
    if (rc) return rc;
    if (n < 1 || !totalMs) {
        sih_set_error("sih_evp_subcycle_timed: bad argument");
        return SIH_ERR_ARG;
    }
    if (cellMs) *cellMs = 0.0;
    if (vertexMs) *vertexMs = 0.0;
    beginSubcycles(c);
    if (useGraph) {
        if (!c->graphExec || c->graphN != n || c->graphForm != 0) {  // the two kernels, whatever the block's own form
            if (c->graphExec) {
                (void)hipGraphExecDestroy(c->graphExec);
                c->graphExec = nullptr;
            }
            hipGraph_t graph;
            HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            for (int it = 0; it < n; it++) {
                launchCellKernel(c, false);
                launchVertexKernel(c, false);
            }
            HIPCHK(hipStreamEndCapture(c->stream, &graph));
            HIPCHK(hipGraphInstantiate(&c->graphExec, graph, nullptr, nullptr, 0));
            HIPCHK(hipGraphDestroy(graph));
            c->graphN = n;
            c->graphForm = 0;
        }
        if ((rc = ensureEvents(c, 2))) return rc;
        HIPCHK(hipEventRecord(c->events[0], c->stream));
        HIPCHK(hipGraphLaunch(c->graphExec, c->stream));
        HIPCHK(hipEventRecord(c->events[1], c->stream));
        HIPCHK(hipEventSynchronize(c->events[1]));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->events[0], c->events[1]));
        *totalMs = ms;
        return SIH_OK;
    }
    int nSamples = (sampleEvery > 0) ? (n + sampleEvery - 1) / sampleEvery : 0;
    if ((rc = ensureEvents(c, 2 + 3 * (size_t)nSamples))) return rc;
    HIPCHK(hipEventRecord(c->events[0], c->stream));
    int si = 0;
    for (int it = 0; it < n; it++) {
        bool sample = sampleEvery > 0 && (it % sampleEvery) == 0;
        if (sample) HIPCHK(hipEventRecord(c->events[2 + 3 * si], c->stream));
        launchCellKernel(c, false);
        if (sample) HIPCHK(hipEventRecord(c->events[2 + 3 * si + 1], c->stream));
        launchVertexKernel(c, false);
        if (sample) {
            HIPCHK(hipEventRecord(c->events[2 + 3 * si + 2], c->stream));
            si++;
        }
    }
    HIPCHK(hipEventRecord(c->events[1], c->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventSynchronize(c->events[1]));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->events[0], c->events[1]));
    *totalMs = ms;
    double sc = 0.0, sv = 0.0;
    for (int k = 0; k < si; k++) {
        float a = 0.f, b = 0.f;
        HIPCHK(hipEventElapsedTime(&a, c->events[2 + 3 * k], c->events[2 + 3 * k + 1]));
        HIPCHK(hipEventElapsedTime(&b, c->events[2 + 3 * k + 1], c->events[2 + 3 * k + 2]));
        sc += a;
        sv += b;
    }
    if (si > 0) {
        if (cellMs) *cellMs = sc / si;
        if (vertexMs) *vertexMs = sv / si;
    }
    return SIH_OK;
}

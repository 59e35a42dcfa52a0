// First-order upwind advection of the reference (/root/reference/src/shared/mpas_seaice_advection_upwind.F,
// selected by config_advection_type = 'upwind', mpas_seaice_advection.F:64, :115).
//
// The reference advects a chain of four (ONE, nCategories, nCells) tracers, each carried by the edge flux of
// the one before (define_tracer_connectivities :145-189):
//     iceAreaCategory ('none') -> surfaceTemperature -> iceVolumeCategory -> snowVolumeCategory
// The first link's parent is a stand-in that is 1 next to ice (prepare_none_parent_tracer :819-915).  For each
// link: upwind flux of the child through every interior edge (upwind_tendencies :1242-1389), new value = old *
// parent + tendency * dt, and the old value is overwritten with the product old * parent, which is the next
// link's parent (run_advection_subvariable :1198-1226).  Afterwards the two time levels trade places and the
// products are divided back link by link (finalize_tracers :1989-2086).
//
// Everything here is a plain streaming kernel over cells or edges (a few bytes of arithmetic per value): the
// cost of the whole step is the HBM traffic of about 30 planes of nCategories values.  Planes are SoA,
// d[k * pad + i]; edges use the internal numbering shared with the incremental remap.
#include <cmath>
#include <cstring>

#include "sih_internal.h"
#include "sih_normals.h"

#define ICE_AREA_MINIMUM 1.0e-11  // seaice_constants: iceAreaMinimum = seaicePuny
#define UW_NCHAIN 4

static const char *const kChain[UW_NCHAIN] = {"iceAreaCategory", "surfaceTemperature", "iceVolumeCategory",
                                              "snowVolumeCategory"};

struct UwDev {
    int nCells, nCellsSolve, nEdges, M, nCat, padC, padE;
    const int *nEdgesOnCell, *coc, *eoc, *coe, *voe, *slot;
    int *interiorEdge;
    const double *areaCell, *dvEdge, *u, *v;
    double *nve, *edgeVelocity;
    double *lev1[UW_NCHAIN], *lev2[UW_NCHAIN], *tend[UW_NCHAIN], *flux[UW_NCHAIN];
    double *pOld, *pNew, *pFlux, *fluxUp;
};

struct sih_upwind_state {
    UwDev d{};
    int nCat = 0;
    bool prepared = false;
    double *d_sums = nullptr;
    std::vector<double> sumInit[UW_NCHAIN], sumFinal[UW_NCHAIN];
};

void sih_upwind_destroy(sih_ctx *c)
{
    if (!c->upwind) return;
    if (c->upwind->d_sums) (void)hipFree(c->upwind->d_sums);
    delete c->upwind;
    c->upwind = nullptr;
}

// ------------------------------------------------------------------------------------------ init
struct UwInit {
    int onSphere, rotate;
    double R;
    const double *xCell, *yCell, *zCell, *xVertex, *yVertex, *zVertex, *xE, *yE, *zE;
};

// seaice_normal_vectors_polygon(mesh, normalVectorEdge, rotate, removeMetricTerms = .false.), mesh.F:760-793:
// the great-circle normal of each edge seen from the cell, without the rotation onto the equator
__global__ void k_uw_init_cells(UwDev d, UwInit a)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    const size_t pc = d.padC, pe = d.padE;
    const int n = d.nEdgesOnCell[c];
    for (int s = 0; s < d.M; s++) {
        double n1 = 0.0, n2 = 0.0;
        if (s < n) {
            const int e = d.eoc[(size_t)s * pc + c];
            const int v1 = d.voe[e], v2 = d.voe[pe + e];
            if (a.onSphere) {
                double er[3], p[3], q[3];
                rotFwd(er, a.xE[e], a.yE[e], a.zE[e], a.rotate);
                rotFwd(p, a.xVertex[v1], a.yVertex[v1], a.zVertex[v1], a.rotate);
                rotFwd(q, a.xVertex[v2], a.yVertex[v2], a.zVertex[v2], a.rotate);
                greatCircleNormal(n1, n2, p, q, er, c == d.coe[pe + e]);
            } else {  // normal_vectors_planar_polygon, mesh.F:858-943
                double tx = a.xVertex[v2] - a.xVertex[v1], ty = a.yVertex[v2] - a.yVertex[v1];
                const double tmag = sqrt(tx * tx + ty * ty);
                tx = tx / tmag;
                ty = ty / tmag;
                const double nx = a.xE[e] - a.xCell[c], ny = a.yE[e] - a.yCell[c];
                if ((nx * ty - ny * tx) < 0.0) {
                    tx = -tx;
                    ty = -ty;
                }
                n1 = ty;
                n2 = -tx;
            }
        }
        d.nve[(size_t)(s * 2) * pc + c] = n1;
        d.nve[(size_t)(s * 2 + 1) * pc + c] = n2;
    }
}

// interior_edges (mesh.F:567-614) for every edge of the block, and the position of each edge among the edges
// of its first cell (edge_from_vertex_velocity takes the normal from there, :1464-1485)
__global__ void k_uw_init_edges(UwDev d, int *slot)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d.nEdges) return;
    const int c1 = d.coe[e], c2 = d.coe[(size_t)d.padE + e];
    d.interiorEdge[e] = (c1 < d.nCells && c2 < d.nCells) ? 1 : 0;
    int sl = -1;
    if (c1 < d.nCells) {
        const int n = d.nEdgesOnCell[c1];
        for (int s = 0; s < n; s++)
            if (d.eoc[(size_t)s * d.padC + c1] == e) sl = s;  // the cell loop's last match wins, as in the reference
    }
    slot[e] = sl;
}

// ------------------------------------------------------------------------------------------ prepare
// initialize_timelevel_variables (:1633-1772) and the volume -> thickness conversion of prepare_tracers
// (:1949-1967, all cells of the block)
__global__ void k_uw_prepare_cells(UwDev d)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    for (int k = 0; k < d.nCat; k++) {
        const size_t i = (size_t)k * d.padC + c;
        for (int t = 0; t < UW_NCHAIN; t++) {
            d.lev2[t][i] = 0.0;
            d.tend[t][i] = 0.0;
        }
        const double a = d.lev1[0][i];
        if (a > ICE_AREA_MINIMUM) {
            d.lev1[2][i] = d.lev1[2][i] / a;
            d.lev1[3][i] = d.lev1[3][i] / a;
        }
    }
}

// zeroed edge fluxes (:1764-1770) and edge_from_vertex_velocity (:1403-1535)
__global__ void k_uw_prepare_edges(UwDev d)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d.nEdges) return;
    for (int k = 0; k < d.nCat; k++)
        for (int t = 0; t < UW_NCHAIN; t++) d.flux[t][(size_t)k * d.padE + e] = 0.0;
    double ev = 0.0;
    const int s = d.slot[e];
    if (s >= 0) {
        const int c1 = d.coe[e];
        double ue = 0.0, ve = 0.0;
        for (int k = 0; k < 2; k++) {
            const int iv = d.voe[(size_t)k * d.padE + e];
            ue = ue + d.u[iv];
            ve = ve + d.v[iv];
        }
        ue = ue / 2.0;
        ve = ve / 2.0;
        ev = ue * d.nve[(size_t)(s * 2) * d.padC + c1] + ve * d.nve[(size_t)(s * 2 + 1) * d.padC + c1];
    }
    d.edgeVelocity[e] = ev;
}

// one tracer of initial_conservation (:2492-2542) / final_conservation (:2645-2656): out[k] = sum over the
// owned cells of child (* product) * areaCell.  With `prodOut` the product parent * child is left there for the
// next tracer.  One block per category; the order of additions differs from the reference's serial loop.
__global__ void k_uw_sum(UwDev d, const double *child, const double *prod, double *prodOut, double *out)
{
    __shared__ double sh[256];
    const int k = blockIdx.x;
    double s = 0.0;
    for (int c = threadIdx.x; c < d.nCellsSolve; c += blockDim.x) {
        const size_t i = (size_t)k * d.padC + c;
        if (prod) {
            s = s + child[i] * prod[i] * d.areaCell[c];
            if (prodOut) prodOut[i] = prod[i] * child[i];
        } else {
            s = s + child[i] * d.areaCell[c];
            if (prodOut) prodOut[i] = child[i];
        }
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = blockDim.x / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = sh[0];
}

__global__ void k_uw_zero_level2(UwDev d)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCellsSolve) return;
    for (int k = 0; k < d.nCat; k++)
        for (int t = 0; t < UW_NCHAIN; t++) d.lev2[t][(size_t)k * d.padC + c] = 0.0;
}

// ------------------------------------------------------------------------------------------ one link
// prepare_none_parent_tracer (:819-915), cell part
__global__ void k_uw_none_cells(UwDev d)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    const int n = d.nEdgesOnCell[c];
    const double *A = d.lev1[0];
    for (int k = 0; k < d.nCat; k++) {
        const size_t pl = (size_t)k * d.padC;
        double p = 0.0;
        if (A[pl + c] > ICE_AREA_MINIMUM) p = 1.0;
        for (int s = 0; s < n; s++)
            if (A[pl + d.coc[(size_t)s * d.padC + c]] > ICE_AREA_MINIMUM) {
                p = 1.0;
                break;
            }
        d.pOld[pl + c] = p;
        d.pNew[pl + c] = p;
    }
}

// prepare_none_parent_tracer, edge part
__global__ void k_uw_none_edges(UwDev d)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d.nEdges) return;
    const int c1 = d.coe[e], c2 = d.coe[(size_t)d.padE + e];
    const double ev = d.edgeVelocity[e];
    const double *A = d.lev1[0];
    for (int k = 0; k < d.nCat; k++) {
        const size_t pl = (size_t)k * d.padC;
        d.pFlux[(size_t)k * d.padE + e] = (A[pl + c1] > ICE_AREA_MINIMUM || A[pl + c2] > ICE_AREA_MINIMUM) ? ev : 0.0;
    }
}

// the edge part of upwind_tendencies (:1346-1383): the upwind flux of the child through each edge.  fluxUp
// keeps dvEdge * (...) for the cell kernel; childEdgeFlux is only set from an owned cell in the reference.
__global__ void k_uw_flux(UwDev d, const double *child, const double *pFlux, const double *pOld, double pMin,
                          double *childEdgeFlux)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d.nEdges) return;
    const int c1 = d.coe[e], c2 = d.coe[(size_t)d.padE + e];
    const bool interior = d.interiorEdge[e] == 1;
    const bool visited = c1 < d.nCellsSolve || c2 < d.nCellsSolve;
    const double dv = d.dvEdge[e];
    for (int k = 0; k < d.nCat; k++) {
        const size_t pl = (size_t)k * d.padC, ie = (size_t)k * d.padE + e;
        double fu = 0.0, ef = 0.0;
        if (interior && (pOld[pl + c1] > pMin || pOld[pl + c2] > pMin)) {
            const double pf = pFlux[ie];
            fu = dv * (fmax(0.0, pf) * child[pl + c1] + fmin(0.0, pf) * child[pl + c2]);
            ef = fu / dv;
        }
        d.fluxUp[ie] = fu;
        childEdgeFlux[ie] = visited ? ef : 0.0;
    }
}

// the cell part of upwind_tendencies (owned cells) and the new tracer (:1198-1226, all cells of the block)
__global__ void k_uw_update(UwDev d, double *childOld, double *childNew, double *childTend, const double *pOld,
                            const double *pNew, double pMin, double dt)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    const int n = d.nEdgesOnCell[c];
    const bool owned = c < d.nCellsSolve;
    const double invArea = owned ? 1.0 / d.areaCell[c] : 0.0;
    for (int k = 0; k < d.nCat; k++) {
        const size_t i = (size_t)k * d.padC + c;
        double tend = 0.0;
        if (owned)
            for (int s = 0; s < n; s++) {
                const int e = d.eoc[(size_t)s * d.padC + c];
                const int sign = (d.coe[e] == c) ? -1 : 1;
                tend = tend + sign * d.fluxUp[(size_t)k * d.padE + e] * invArea;
            }
        childTend[i] = tend;
        double nw = 0.0;
        if (pNew[i] > pMin) {
            const double o = childOld[i], p = pOld[i];
            nw = o * p + tend * dt;
            childOld[i] = o * p;
        }
        childNew[i] = nw;
    }
}

// finalize_tracers (:1989-2086) after the conservation sums: the time levels trade places, the products are
// divided back from the last link to the first (scale_tracers_back :2141-2253, owned cells) and thickness
// becomes volume again (owned cells)
__global__ void k_uw_finalize(UwDev d)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    const bool owned = c < d.nCellsSolve;
    for (int k = 0; k < d.nCat; k++) {
        const size_t i = (size_t)k * d.padC + c;
        double nw[UW_NCHAIN], od[UW_NCHAIN];
        for (int t = 0; t < UW_NCHAIN; t++) {
            od[t] = d.lev1[t][i];
            nw[t] = d.lev2[t][i];
        }
        if (owned) {
            for (int t = UW_NCHAIN - 1; t >= 1; t--) nw[t] = (nw[t - 1] > 0.0) ? nw[t] / nw[t - 1] : 0.0;
            if (nw[0] > ICE_AREA_MINIMUM) {
                nw[2] = nw[2] * nw[0];
                nw[3] = nw[3] * nw[0];
            }
        }
        for (int t = 0; t < UW_NCHAIN; t++) {
            d.lev1[t][i] = nw[t];
            d.lev2[t][i] = od[t];
        }
    }
}

// ------------------------------------------------------------------------------------------ host side
extern "C" int sih_upwind_init(sih_ctx *c, const sih_ir_desc *m)
{
    if (!c || !m) {
        sih_set_error("sih_upwind_init: null argument");
        return SIH_ERR_ARG;
    }
    if (!m->edgesOnCell || !m->cellsOnEdge || !m->verticesOnEdge || !m->edgesOnVertex || !m->xEdge || !m->yEdge ||
        !m->zEdge || !m->dcEdge || !m->dvEdge || m->nCategories < 1 || c->nEdges < 1) {
        sih_set_error("sih_upwind_init: incomplete edge description of the mesh");
        return SIH_ERR_ARG;
    }
    if (c->D != 3) {
        sih_set_error("upwind advection: only vertexDegree = 3 (hexagonal meshes) is implemented");
        return SIH_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(c->device));
    int rc;
    // the incremental remap may already have put the edge tables on the device; they are the same ones
    if (!c->ir && (rc = sih_setup_edges(c, m, "sih_upwind_init"))) return rc;
    sih_upwind_destroy(c);
    sih_upwind_state *u = new sih_upwind_state();
    c->upwind = u;
    u->nCat = m->nCategories;
    const int nCat = u->nCat, M = c->M;
    if ((rc = sih_add_field(c, "normalVectorEdge", LOC_CELL, 2 * M, false))) return rc;
    if ((rc = sih_add_field(c, "edgeVelocity", LOC_EDGE, 1, false))) return rc;
    if ((rc = sih_add_field(c, "interiorEdge", LOC_EDGE, 1, true))) return rc;
    if ((rc = sih_add_field(c, "_uwSlot", LOC_EDGE, 1, true))) return rc;
    if ((rc = sih_add_field(c, "_uwParentOld", LOC_CELL, nCat, false))) return rc;
    if ((rc = sih_add_field(c, "_uwParentNew", LOC_CELL, nCat, false))) return rc;
    if ((rc = sih_add_field(c, "_uwParentFlux", LOC_EDGE, nCat, false))) return rc;
    if ((rc = sih_add_field(c, "_uwFluxUp", LOC_EDGE, nCat, false))) return rc;
    auto fi = [&](const char *n) { return (int *)sih_find_field(c, n)->d; };
    auto fd = [&](const std::string &n) { return (double *)sih_find_field(c, n.c_str())->d; };
    UwDev &d = u->d;
    for (int t = 0; t < UW_NCHAIN; t++) {
        const std::string n = kChain[t];
        if ((rc = sih_add_field(c, n.c_str(), LOC_CELL, nCat, false))) return rc;
        if ((rc = sih_add_field(c, (n + "Level2").c_str(), LOC_CELL, nCat, false))) return rc;
        if ((rc = sih_add_field(c, (n + "Tend").c_str(), LOC_CELL, nCat, false))) return rc;
        if ((rc = sih_add_field(c, (n + "EdgeFlux").c_str(), LOC_EDGE, nCat, false))) return rc;
        d.lev1[t] = fd(n);
        d.lev2[t] = fd(n + "Level2");
        d.tend[t] = fd(n + "Tend");
        d.flux[t] = fd(n + "EdgeFlux");
        u->sumInit[t].assign(nCat, 0.0);
        u->sumFinal[t].assign(nCat, 0.0);
    }
    d.nCells = c->nCells;
    d.nCellsSolve = c->nCellsSolve;
    d.nEdges = c->nEdges;
    d.M = M;
    d.nCat = nCat;
    d.padC = c->padC;
    d.padE = c->padE;
    d.nEdgesOnCell = fi("_nEdgesOnCell");
    d.coc = fi("_coc");
    d.eoc = fi("_eoc");
    d.coe = fi("_coe");
    d.voe = fi("_voe");
    d.slot = fi("_uwSlot");
    d.interiorEdge = fi("interiorEdge");
    d.areaCell = fd("areaCell");
    d.dvEdge = fd("dvEdge");
    d.u = fd("uVelocity");
    d.v = fd("vVelocity");
    d.nve = fd("normalVectorEdge");
    d.edgeVelocity = fd("edgeVelocity");
    d.pOld = fd("_uwParentOld");
    d.pNew = fd("_uwParentNew");
    d.pFlux = fd("_uwParentFlux");
    d.fluxUp = fd("_uwFluxUp");
    HIPCHK(hipMalloc((void **)&u->d_sums, sizeof(double) * nCat));

    UwInit a{c->onASphere, m->rotateCartesianGrid, c->R, fd("xCell"), fd("yCell"), fd("zCell"), fd("xVertex"),
             fd("yVertex"), fd("zVertex"), fd("xEdge"), fd("yEdge"), fd("zEdge")};
    const dim3 blk(256);
    hipLaunchKernelGGL(k_uw_init_cells, dim3((d.nCells + 255) / 256), blk, 0, c->stream, d, a);
    hipLaunchKernelGGL(k_uw_init_edges, dim3((d.nEdges + 255) / 256), blk, 0, c->stream, d, fi("_uwSlot"));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return SIH_OK;
}

static int uwSums(sih_ctx *c, bool init)
{
    sih_upwind_state *u = c->upwind;
    UwDev &d = u->d;
    for (int t = 0; t < UW_NCHAIN; t++) {
        if (init)  // the products of the chain are built in the second time level, which is zeroed afterwards
            hipLaunchKernelGGL(k_uw_sum, dim3(d.nCat), dim3(256), 0, c->stream, d, d.lev1[t], t ? d.lev2[t - 1] : nullptr,
                               d.lev2[t], u->d_sums);
        else  // after the update the second level holds the new values (the levels have not traded places yet)
            hipLaunchKernelGGL(k_uw_sum, dim3(d.nCat), dim3(256), 0, c->stream, d, d.lev2[t], nullptr, nullptr, u->d_sums);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync((init ? u->sumInit : u->sumFinal)[t].data(), u->d_sums, sizeof(double) * d.nCat,
                              hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (init) {
        hipLaunchKernelGGL(k_uw_zero_level2, dim3((d.nCellsSolve + 255) / 256), dim3(256), 0, c->stream, d);
        HIPCHK(hipGetLastError());
    }
    return SIH_OK;
}

static int uwReady(sih_ctx *c, const char *who)
{
    if (!c || !c->upwind) {
        sih_set_error(std::string(who) + ": sih_upwind_init has not run");
        return SIH_ERR_STATE;
    }
    return SIH_OK;
}

extern "C" int sih_upwind_prepare(sih_ctx *c, int checks)
{
    int rc;
    if ((rc = uwReady(c, "sih_upwind_prepare"))) return rc;
    HIPCHK(hipSetDevice(c->device));
    UwDev &d = c->upwind->d;
    const dim3 blk(256);
    hipLaunchKernelGGL(k_uw_prepare_cells, dim3((d.nCells + 255) / 256), blk, 0, c->stream, d);
    hipLaunchKernelGGL(k_uw_prepare_edges, dim3((d.nEdges + 255) / 256), blk, 0, c->stream, d);
    HIPCHK(hipGetLastError());
    if ((checks & SIH_IR_CHECK_CONSERVATION) && (rc = uwSums(c, true))) return rc;
    c->upwind->prepared = true;
    return SIH_OK;
}

extern "C" int sih_upwind_advect(sih_ctx *c, double dt, int checks)
{
    int rc;
    if ((rc = uwReady(c, "sih_upwind_advect"))) return rc;
    sih_upwind_state *u = c->upwind;
    if (!u->prepared) {
        sih_set_error("sih_upwind_advect: sih_upwind_prepare must run first");
        return SIH_ERR_STATE;
    }
    HIPCHK(hipSetDevice(c->device));
    UwDev &d = u->d;
    const dim3 blk(256), gc((d.nCells + 255) / 256), ge((d.nEdges + 255) / 256);
    const double pMin = 0.0;  // every parentTracerMinimum of the reference's table is zero (:342-367)
    for (int t = 0; t < UW_NCHAIN; t++) {
        const double *pOld, *pNew, *pFlux;
        if (t == 0) {
            hipLaunchKernelGGL(k_uw_none_cells, gc, blk, 0, c->stream, d);
            hipLaunchKernelGGL(k_uw_none_edges, ge, blk, 0, c->stream, d);
            pOld = d.pOld;
            pNew = d.pNew;
            pFlux = d.pFlux;
        } else {
            pOld = d.lev1[t - 1];
            pNew = d.lev2[t - 1];
            pFlux = d.flux[t - 1];
        }
        hipLaunchKernelGGL(k_uw_flux, ge, blk, 0, c->stream, d, (const double *)d.lev1[t], pFlux, pOld, pMin, d.flux[t]);
        hipLaunchKernelGGL(k_uw_update, gc, blk, 0, c->stream, d, d.lev1[t], d.lev2[t], d.tend[t], pOld, pNew, pMin, dt);
    }
    HIPCHK(hipGetLastError());
    if ((checks & SIH_IR_CHECK_CONSERVATION) && (rc = uwSums(c, false))) return rc;
    hipLaunchKernelGGL(k_uw_finalize, gc, blk, 0, c->stream, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    u->prepared = false;
    return SIH_OK;
}

extern "C" int sih_upwind_run(sih_ctx *c, double dt, int checks)
{
    int rc;
    if ((rc = sih_upwind_prepare(c, checks))) return rc;
    return sih_upwind_advect(c, dt, checks);
}

extern "C" int sih_upwind_get_sums(sih_ctx *c, const char *tracer, double *initSums, double *finalSums)
{
    int rc;
    if ((rc = uwReady(c, "sih_upwind_get_sums"))) return rc;
    for (int t = 0; t < UW_NCHAIN; t++)
        if (tracer && !strcmp(tracer, kChain[t])) {
            const size_t nb = sizeof(double) * c->upwind->nCat;
            if (initSums) memcpy(initSums, c->upwind->sumInit[t].data(), nb);
            if (finalSums) memcpy(finalSums, c->upwind->sumFinal[t].data(), nb);
            return SIH_OK;
        }
    sih_set_error("sih_upwind_get_sums: not a tracer of the upwind chain");
    return SIH_ERR_ARG;
}

// The "weak" (cell-centred) strain / stress / stress-divergence scheme of the reference
// (/root/reference/src/shared/mpas_seaice_velocity_solver_weak.F, selected by config_strain_scheme =
// config_stress_divergence_scheme = 'weak', velocity_solver.F:169-187) and the normal vectors it is built
// on (mpas_seaice_mesh.F:703-1606).  One strain / stress value per cell instead of one per cell corner:
//   strain  = (1/A) sum over the cell's edges of the edge-midpoint velocity times the outward normal
//   stress  = the same constitutive relations, once per cell
//   div     = (1/A_triangle) sum over the vertex's three edges of the edge-mean stress times the normal of
//             the dual triangle's side
// The default scheme is the variational one; this one is kept simple: two extra launches per subcycle
// (cells, divergence) in front of the shared vertex solve.
#include <cmath>
#include <cstring>

#include "sih_internal.h"
#include "sih_normals.h"

#define ECC2 4.0
#define PUNY 1.0e-11

struct WeakDev {
    int nCells, nCellsSolve, nVertices, nVerticesSolve, M, padC, padV, lastOwnedCell;
    const int *nEdgesOnCell, *voc, *cov, *solveStress, *solveVelocity, *interiorVertex;
    const int *ev1, *ev2;  // (M, cell): the two vertices of edgesOnCell(s, c)
    const int *ec1, *ec2;  // (3, vertex): the two cells of edgesOnVertex(k, v)
    const double *dvOnCell, *dcOnVertex, *areaCell, *areaTriangle, *icePressure;
    double *nPoly, *nTri, *latCell, *latVertex;  // normalVectorPolygon (2, M, cell), normalVectorTriangle (2, 3, vertex)
    double *strain11, *strain22, *strain12, *stress11, *stress22, *stress12, *Prep, *prin1, *prin2;
    const double *u, *v;
    double *divU, *divV, *divergence, *shear, *ridgeConv, *ridgeShear;
    double sphereRadius, dtE, dampT;
    int relation, useColumn;
    // weak strain + variational divergence: strains at vertices and the variational pool's corner strains
    double *sv11, *sv22, *sv12, *var11, *var22, *var12;
    const int *gidx;
};

struct sih_weak_state {
    // edge description in host numbering, kept from sih_evp_set_edge_mesh until sih_evp_init uses it
    std::vector<int> edgesOnCell, cellsOnEdge, verticesOnEdge, edgesOnVertex;
    std::vector<double> xEdge, yEdge, zEdge, dcEdge, dvEdge;
    bool haveEdges = false, ready = false;
    WeakDev d{};
};

void sih_weak_destroy(sih_ctx *c)
{
    delete c->weak;
    c->weak = nullptr;
}

extern "C" int sih_evp_set_edge_mesh(sih_ctx *c, const sih_ir_desc *m)
{
    if (!c || !m) {
        sih_set_error("sih_evp_set_edge_mesh: null argument");
        return SIH_ERR_ARG;
    }
    if (!m->edgesOnCell || !m->cellsOnEdge || !m->verticesOnEdge || !m->edgesOnVertex || !m->xEdge || !m->yEdge ||
        !m->zEdge || !m->dcEdge || !m->dvEdge || c->nEdges < 1) {
        sih_set_error("sih_evp_set_edge_mesh: incomplete edge description of the mesh");
        return SIH_ERR_ARG;
    }
    const int nC = c->nCells, nV = c->nVertices, nE = c->nEdges, M = c->M;
    for (int e = 0; e < nE; e++)
        for (int k = 0; k < 2; k++) {
            const int cc = m->cellsOnEdge[(size_t)e * 2 + k], vv = m->verticesOnEdge[(size_t)e * 2 + k];
            if (cc < 0 || cc > nC + 1 || vv < 1 || vv > nV) {
                sih_set_error("sih_evp_set_edge_mesh: cellsOnEdge / verticesOnEdge index out of range");
                return SIH_ERR_ARG;
            }
        }
    for (size_t k = 0; k < (size_t)nC * M; k++)
        if (m->edgesOnCell[k] < 0 || m->edgesOnCell[k] > nE + 1) {
            sih_set_error("sih_evp_set_edge_mesh: edgesOnCell index out of range");
            return SIH_ERR_ARG;
        }
    for (size_t k = 0; k < (size_t)nV * 3; k++)
        if (m->edgesOnVertex[k] < 0 || m->edgesOnVertex[k] > nE + 1) {
            sih_set_error("sih_evp_set_edge_mesh: edgesOnVertex index out of range");
            return SIH_ERR_ARG;
        }
    if (!c->weak) c->weak = new sih_weak_state();
    sih_weak_state *w = c->weak;
    w->edgesOnCell.assign(m->edgesOnCell, m->edgesOnCell + (size_t)nC * M);
    w->cellsOnEdge.assign(m->cellsOnEdge, m->cellsOnEdge + (size_t)nE * 2);
    w->verticesOnEdge.assign(m->verticesOnEdge, m->verticesOnEdge + (size_t)nE * 2);
    w->edgesOnVertex.assign(m->edgesOnVertex, m->edgesOnVertex + (size_t)nV * 3);
    w->xEdge.assign(m->xEdge, m->xEdge + nE);
    w->yEdge.assign(m->yEdge, m->yEdge + nE);
    w->zEdge.assign(m->zEdge, m->zEdge + nE);
    w->dcEdge.assign(m->dcEdge, m->dcEdge + nE);
    w->dvEdge.assign(m->dvEdge, m->dvEdge + nE);
    w->haveEdges = true;
    w->ready = false;
    return SIH_OK;
}

// ------------------------------------------------------------------------------------------ normals
struct WeakInit {
    int nCells, nVertices, nVerticesSolve, nEdges, M, padC, padV, onSphere, rotate;
    double R;
    const int *nEdgesOnCell, *cellPerm, *vertPerm, *cellInv, *vertInv, *interiorVertex;
    // host-numbered edge description (temporary device copies)
    const int *eoc, *coe, *voe, *eov;
    const double *xE, *yE, *zE;
    const double *xCell, *yCell, *zCell, *xVertex, *yVertex, *zVertex;  // library fields (internal numbering)
    double *nPoly, *nTri, *latCell, *latVertex;
};

// seaice_normal_vectors_polygon: normal_vectors_spherical_polygon_metric (:1038-1241) / _planar_polygon (:858-943)
__global__ void k_weak_init_cells(WeakInit a)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nCells) return;
    const int h = a.cellPerm[c], M = a.M, n = a.nEdgesOnCell[c];
    const size_t pc = a.padC;
    double cr[3], lon = 0.0, lat = 0.0;
    if (a.onSphere) {
        rotFwd(cr, a.xCell[c], a.yCell[c], a.zCell[c], a.rotate);
        lon = atan2(cr[1], cr[0]);
        lat = asin(cr[2] / a.R);
    }
    for (int s = 0; s < M; s++) {
        double n1 = 0.0, n2 = 0.0;
        if (s < n) {
            const int e = a.eoc[(size_t)h * M + s] - 1;
            const int v1 = a.vertInv[a.voe[(size_t)e * 2] - 1], v2 = a.vertInv[a.voe[(size_t)e * 2 + 1] - 1];
            if (a.onSphere) {
                double er[3], p[3], q[3], eq[3], pq[3], qq[3];
                rotFwd(er, a.xE[e], a.yE[e], a.zE[e], a.rotate);
                rotFwd(p, a.xVertex[v1], a.yVertex[v1], a.zVertex[v1], a.rotate);
                rotFwd(q, a.xVertex[v2], a.yVertex[v2], a.zVertex[v2], a.rotate);
                toEquator(eq, er, lat, lon);
                toEquator(pq, p, lat, lon);
                toEquator(qq, q, lat, lon);
                greatCircleNormal(n1, n2, pq, qq, eq, h + 1 == a.coe[(size_t)e * 2 + 1]);
            } else {
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
        a.nPoly[(size_t)(s * 2) * pc + c] = n1;
        a.nPoly[(size_t)(s * 2 + 1) * pc + c] = n2;
    }
    a.latCell[c] = lat;
}

// seaice_normal_vectors_triangle: normal_vectors_spherical_triangle_metric (:1393-1606, owned vertices) /
// _planar_triangle (:957-1024, all vertices)
__global__ void k_weak_init_vertices(WeakInit a)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= a.nVertices) return;
    const size_t pv = a.padV;
    const int h = a.vertPerm[v];
    double n1[3] = {0.0, 0.0, 0.0}, n2[3] = {0.0, 0.0, 0.0}, lat = 0.0;
    const bool looped = a.onSphere ? (h < a.nVerticesSolve) : true;
    if (looped && a.interiorVertex[v] == 1) {
        double vr[3], lon = 0.0;
        if (a.onSphere) {
            rotFwd(vr, a.xVertex[v], a.yVertex[v], a.zVertex[v], a.rotate);
            lon = atan2(vr[1], vr[0]);
            lat = asin(vr[2] / a.R);
        }
        for (int k = 0; k < 3; k++) {
            const int e = a.eov[(size_t)h * 3 + k] - 1;
            if (e < 0 || e >= a.nEdges) continue;  // cannot happen at an interior vertex
            if (a.onSphere) {
                const int c1 = a.cellInv[a.coe[(size_t)e * 2] - 1], c2 = a.cellInv[a.coe[(size_t)e * 2 + 1] - 1];
                double er[3], p[3], q[3], eq[3], pq[3], qq[3];
                rotFwd(er, a.xE[e], a.yE[e], a.zE[e], a.rotate);
                rotFwd(p, a.xCell[c1], a.yCell[c1], a.zCell[c1], a.rotate);
                rotFwd(q, a.xCell[c2], a.yCell[c2], a.zCell[c2], a.rotate);
                toEquator(eq, er, lat, lon);
                toEquator(pq, p, lat, lon);
                toEquator(qq, q, lat, lon);
                greatCircleNormal(n1[k], n2[k], pq, qq, eq, h + 1 == a.voe[(size_t)e * 2]);
            } else {
                const double dx = a.xE[e] - a.xVertex[v], dy = a.yE[e] - a.yVertex[v];
                n1[k] = dx / sqrt(dx * dx + dy * dy);
                n2[k] = dy / sqrt(dx * dx + dy * dy);
            }
        }
    }
    for (int k = 0; k < 3; k++) {
        a.nTri[(size_t)(k * 2) * pv + v] = n1[k];
        a.nTri[(size_t)(k * 2 + 1) * pv + v] = n2[k];
    }
    a.latVertex[v] = lat;
}

// ------------------------------------------------------------------------------------------ time step
// init_subcycle_variables, weak branch (velocity_solver.F:2347-2380)
__global__ void k_weak_pre(WeakDev d)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    d.strain11[c] = 0.0;
    d.strain22[c] = 0.0;
    d.strain12[c] = 0.0;
    if (d.solveStress[c] != 1) {
        d.stress11[c] = 0.0;
        d.stress22[c] = 0.0;
        d.stress12[c] = 0.0;
    }
}

// seaice_strain_tensor_weak (velocity_solver_weak.F:112-253) and seaice_stress_tensor_weak (:267-405)
// strainOnly: the mixed scheme uses the weak strain but the variational stresses
__global__ void __launch_bounds__(256) k_weak_cells(WeakDev d, int strainOnly)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    const size_t pc = d.padC;
    double s11 = 0.0, s22 = 0.0, s12 = 0.0;
    if (d.solveStress[c] != 1) {
        d.strain11[c] = 0.0;
        d.strain22[c] = 0.0;
        d.strain12[c] = 0.0;
        if (!strainOnly) {
            d.stress11[c] = 0.0;
            d.stress22[c] = 0.0;
            d.stress12[c] = 0.0;
        }
        return;
    }
    const int n = d.nEdgesOnCell[c];
    double uc = 0.0, vc = 0.0;
    for (int s = 0; s < n; s++) {
        const int iv = d.voc[(size_t)s * pc + c];
        uc = uc + d.u[iv];
        vc = vc + d.v[iv];
        const int w1 = d.ev1[(size_t)s * pc + c], w2 = d.ev2[(size_t)s * pc + c];
        double ue = 0.0, ve = 0.0;
        ue = ue + d.u[w1];
        ve = ve + d.v[w1];
        ue = ue + d.u[w2];
        ve = ve + d.v[w2];
        ue = ue / 2.0;
        ve = ve / 2.0;
        const double n1 = d.nPoly[(size_t)(s * 2) * pc + c], n2 = d.nPoly[(size_t)(s * 2 + 1) * pc + c];
        const double dv = d.dvOnCell[(size_t)s * pc + c];
        s11 = s11 + ue * n1 * dv;
        s22 = s22 + ve * n2 * dv;
        s12 = s12 + 0.5 * (ue * n2 + ve * n1) * dv;
    }
    uc = uc / (double)n;
    vc = vc / (double)n;
    const double area = d.areaCell[c];
    s11 = s11 / area;
    s22 = s22 / area;
    s12 = s12 / area;
    const double tl = tan(d.latCell[c]);
    s11 = s11 - (vc * tl) / d.sphereRadius;
    s12 = s12 + (uc * tl * 0.5) / d.sphereRadius;
    d.strain11[c] = s11;
    d.strain22[c] = s22;
    d.strain12[c] = s12;
    if (strainOnly) return;
    double a11, a22, a12;
    if (d.relation == 3) {
        a11 = 1.0 * s11;
        a22 = 1.0 * s22;
        a12 = 1.0 * s12;
    } else {
        a11 = d.stress11[c];
        a22 = d.stress22[c];
        a12 = d.stress12[c];
        const double eD = s11 + s22, eT = s11 - s22, eS = s12 * 2.0;
        double st1 = a11 + a22, st2 = a11 - a22;
        const double Delta = sqrt(eD * eD + (eT * eT + eS * eS) / ECC2);
        double pcf = d.icePressure[c] / fmax(Delta, PUNY);
        d.Prep[c] = pcf * Delta;
        double den;
        if (d.relation == 1) {
            pcf = (pcf * d.dtE) / (2.0 * d.dampT);
            den = 1.0 + (0.5 * d.dtE) / d.dampT;
        } else {
            pcf = (pcf * 2.0 * 5.5e-3) / 0.86;
            den = 1.0 + (2.0 * 5.5e-3) / 0.86;
        }
        st1 = (st1 + pcf * (eD - Delta)) / den;
        st2 = (st2 + (pcf / ECC2) * eT) / den;
        a12 = (a12 + (pcf / ECC2) * eS * 0.5) / den;
        a11 = 0.5 * (st1 + st2);
        a22 = 0.5 * (st1 - st2);
    }
    d.stress11[c] = a11;
    d.stress22[c] = a22;
    d.stress12[c] = a12;
}

// seaice_stress_divergence_weak (velocity_solver_weak.F:493-640): all vertices of the block; a missing cell
// points to the zero-filled slot behind the last cell, as the reference's arrays have it
__global__ void __launch_bounds__(256) k_weak_divergence(WeakDev d)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= d.nVertices) return;
    const size_t pv = d.padV;
    double dU = 0.0, dV = 0.0;
    if (d.solveVelocity[v] == 1) {
        double s11v = 0.0, s22v = 0.0, s12v = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int c = d.cov[k * pv + v];
            s11v = s11v + d.stress11[c];
            s22v = s22v + d.stress22[c];
            s12v = s12v + d.stress12[c];
            const int c1 = d.ec1[k * pv + v], c2 = d.ec2[k * pv + v];
            double s11e = 0.0, s22e = 0.0, s12e = 0.0;
            s11e = s11e + d.stress11[c1];
            s22e = s22e + d.stress22[c1];
            s12e = s12e + d.stress12[c1];
            s11e = s11e + d.stress11[c2];
            s22e = s22e + d.stress22[c2];
            s12e = s12e + d.stress12[c2];
            s11e = s11e / 2.0;
            s22e = s22e / 2.0;
            s12e = s12e / 2.0;
            const double n1 = d.nTri[(size_t)(k * 2) * pv + v], n2 = d.nTri[(size_t)(k * 2 + 1) * pv + v];
            const double dc = d.dcOnVertex[k * pv + v];
            dU = dU + (s11e * n1 + s12e * n2) * dc;
            dV = dV + (s22e * n2 + s12e * n1) * dc;
        }
        s11v = s11v / 3.0;
        s22v = s22v / 3.0;
        s12v = s12v / 3.0;
        const double at = d.areaTriangle[v];
        dU = dU / at;
        dV = dV / at;
        const double tl = tan(d.latVertex[v]);
        dU = dU - (tl * s12v * 2.0) / d.sphereRadius;
        dV = dV + (tl * (s11v - s22v)) / d.sphereRadius;
    }
    d.divU[v] = dU;
    d.divV[v] = dV;
}

// interpolate_strains_weak_to_variational (velocity_solver.F:2877-2960), first loop: area-weighted mean of
// the cell strains at every owned vertex
__global__ void k_hybrid_vertex_strain(WeakDev d)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= d.nVerticesSolve) return;
    const size_t pv = d.padV;
    double a = 0.0, b = 0.0, c12 = 0.0, denom = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int c = d.cov[k * pv + v];
        if (c < d.nCells) {
            const double area = d.areaCell[c];
            a = a + area * d.strain11[c];
            b = b + area * d.strain22[c];
            c12 = c12 + area * d.strain12[c];
            denom = denom + area;
        }
    }
    d.sv11[v] = a / denom;
    d.sv22[v] = b / denom;
    d.sv12[v] = c12 / denom;
}

// second loop: every corner of every cell takes its vertex's value
__global__ void k_hybrid_fill_corners(WeakDev d)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    const size_t pc = d.padC;
    const int n = d.nEdgesOnCell[c];
    for (int s = 0; s < n; s++) {
        const int v = d.voc[(size_t)s * pc + c];
        d.var11[(size_t)s * pc + c] = d.sv11[v];
        d.var22[(size_t)s * pc + c] = d.sv22[v];
        d.var12[(size_t)s * pc + c] = d.sv12[v];
    }
}

// seaice_final_divergence_shear_weak (velocity_solver_weak.F:654-749) and the weak branch of
// principal_stresses_driver (velocity_solver.F:3497-3516).  At :720 the reference assigns the scalar Delta of
// the current cell to its whole work array, so the ridging diagnostics of every cell are computed from the
// last owned cell's Delta; the same value is used here.
__global__ void k_weak_post(WeakDev d)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    // owned cells in the reference's numbering are the first nCellsSolve ones; internally they come first too
    if (c >= d.nCellsSolve) return;
    const double sd = d.strain11[c] + d.strain22[c], st = d.strain11[c] - d.strain22[c], ss = d.strain12[c] * 2.0;
    d.divergence[c] = sd;
    d.shear[c] = sqrt(st * st + ss * ss);
    if (d.useColumn) {
        const int l = d.lastOwnedCell;
        const double ld = d.strain11[l] + d.strain22[l], lt = d.strain11[l] - d.strain22[l], ls = d.strain12[l] * 2.0;
        const double DeltaAll = sqrt(ld * ld + (lt * lt + ls * ls) / ECC2);
        d.ridgeConv[c] = -fmin(sd, 0.0);
        d.ridgeShear[c] = 0.5 * (DeltaAll - fabs(sd));
    }
    const double rp = d.Prep[c];
    if (rp > PUNY) {
        const double a11 = d.stress11[c], a22 = d.stress22[c], a12 = d.stress12[c];
        const double sq = (a11 + a22) * (a11 + a22) - 4.0 * a11 * a22 + 4.0 * (a12 * a12);
        const double p1 = 0.5 * (a11 + a22) + 0.5 * sqrt(sq), p2 = 0.5 * (a11 + a22) - 0.5 * sqrt(sq);
        d.prin1[c] = p1 / rp;
        d.prin2[c] = p2 / rp;
    } else {
        d.prin1[c] = 1.0e30;
        d.prin2[c] = 1.0e30;
    }
}

// ------------------------------------------------------------------------------------------ host side
template <typename T>
static int tmpUpload(T **dst, const std::vector<T> &v)
{
    *dst = nullptr;
    if (v.empty()) return SIH_OK;
    HIPCHK(hipMalloc((void **)dst, sizeof(T) * v.size()));
    HIPCHK(hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return SIH_OK;
}

// fields of the velocity_weak pool under their Registry.xml names (:3730-3744) plus the per-corner /
// per-vertex-edge lookups the kernels use; called from sih_evp_init when the weak scheme is selected,
// after interiorVertex has been computed
int sih_weak_init(sih_ctx *c)
{
    sih_weak_state *w = c->weak;
    if (!w || !w->haveEdges) {
        sih_set_error("the weak scheme needs the edge description of the mesh: call sih_evp_set_edge_mesh first");
        return SIH_ERR_STATE;
    }
    const int nC = c->nCells, nV = c->nVertices, nE = c->nEdges, M = c->M;
    int rc;
    struct F {
        const char *n;
        SihLoc loc;
        int lead;
        bool isInt;
    };
    const F fields[] = {{"normalVectorPolygon", LOC_CELL, 2 * M, false}, {"normalVectorTriangle", LOC_VERTEX, 6, false},
                        {"latCellRotated", LOC_CELL, 1, false},          {"latVertexRotated", LOC_VERTEX, 1, false},
                        {"strain11weak", LOC_CELL, 1, false},            {"strain22weak", LOC_CELL, 1, false},
                        {"strain12weak", LOC_CELL, 1, false},            {"stress11weak", LOC_CELL, 1, false},
                        {"stress22weak", LOC_CELL, 1, false},            {"stress12weak", LOC_CELL, 1, false},
                        {"principalStress1Weak", LOC_CELL, 1, false},    {"principalStress2Weak", LOC_CELL, 1, false},
                        {"replacementPressureWeak", LOC_CELL, 1, false}, {"_weakEv1", LOC_CELL, M, true},
                        {"_weakEv2", LOC_CELL, M, true},                 {"_weakEc1", LOC_VERTEX, 3, true},
                        {"_weakEc2", LOC_VERTEX, 3, true},               {"_weakDv", LOC_CELL, M, false},
                        {"_weakDc", LOC_VERTEX, 3, false},               {"strain11Vertex", LOC_VERTEX, 1, false},
                        {"strain22Vertex", LOC_VERTEX, 1, false},        {"strain12Vertex", LOC_VERTEX, 1, false}};
    for (const F &f : fields)
        if ((rc = sih_add_field(c, f.n, f.loc, f.lead, f.isInt))) return rc;
    auto fi = [&](const char *n) { return (int *)sih_find_field(c, n)->d; };
    auto fd = [&](const char *n) { return (double *)sih_find_field(c, n)->d; };
    // lookups in internal numbering, built on the host
    {
        std::vector<int> ev1((size_t)M * c->padC, nV), ev2((size_t)M * c->padC, nV);
        std::vector<double> dv((size_t)M * c->padC, 0.0);
        for (int i = 0; i < nC; i++) {
            const int h = c->cellPerm[i];
            for (int s = 0; s < c->hNe[i]; s++) {
                const int e = w->edgesOnCell[(size_t)h * M + s];
                if (e < 1 || e > nE) continue;
                ev1[(size_t)s * c->padC + i] = c->vertInv[w->verticesOnEdge[(size_t)(e - 1) * 2] - 1];
                ev2[(size_t)s * c->padC + i] = c->vertInv[w->verticesOnEdge[(size_t)(e - 1) * 2 + 1] - 1];
                dv[(size_t)s * c->padC + i] = w->dvEdge[e - 1];
            }
        }
        std::vector<int> ec1((size_t)3 * c->padV, nC), ec2((size_t)3 * c->padV, nC);
        std::vector<double> dc((size_t)3 * c->padV, 0.0);
        for (int i = 0; i < nV; i++) {
            const int h = c->vertPerm[i];
            for (int k = 0; k < 3; k++) {
                const int e = w->edgesOnVertex[(size_t)h * 3 + k];
                if (e < 1 || e > nE) continue;
                const int c1 = w->cellsOnEdge[(size_t)(e - 1) * 2], c2 = w->cellsOnEdge[(size_t)(e - 1) * 2 + 1];
                ec1[(size_t)k * c->padV + i] = (c1 >= 1 && c1 <= nC) ? c->cellInv[c1 - 1] : nC;
                ec2[(size_t)k * c->padV + i] = (c2 >= 1 && c2 <= nC) ? c->cellInv[c2 - 1] : nC;
                dc[(size_t)k * c->padV + i] = w->dcEdge[e - 1];
            }
        }
        HIPCHK(hipMemcpyAsync(fi("_weakEv1"), ev1.data(), ev1.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(fi("_weakEv2"), ev2.data(), ev2.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(fi("_weakEc1"), ec1.data(), ec1.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(fi("_weakEc2"), ec2.data(), ec2.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(fd("_weakDv"), dv.data(), dv.size() * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(fd("_weakDc"), dc.data(), dc.size() * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    // normal vectors and rotated latitudes: device kernels over temporary copies of the edge description
    WeakInit a{};
    a.nCells = nC;
    a.nVertices = nV;
    a.nVerticesSolve = c->nVerticesSolve;
    a.nEdges = nE;
    a.M = M;
    a.padC = c->padC;
    a.padV = c->padV;
    a.onSphere = c->onASphere;
    a.rotate = c->cfg.rotateCartesianGrid;
    a.R = c->R;
    a.nEdgesOnCell = fi("_nEdgesOnCell");
    a.cellPerm = c->d_cellPerm;
    a.vertPerm = c->d_vertPerm;
    a.interiorVertex = fi("interiorVertex");
    a.xCell = fd("xCell");
    a.yCell = fd("yCell");
    a.zCell = fd("zCell");
    a.xVertex = fd("xVertex");
    a.yVertex = fd("yVertex");
    a.zVertex = fd("zVertex");
    a.nPoly = fd("normalVectorPolygon");
    a.nTri = fd("normalVectorTriangle");
    a.latCell = fd("latCellRotated");
    a.latVertex = fd("latVertexRotated");
    int *t_eoc, *t_coe, *t_voe, *t_eov, *t_cinv, *t_vinv;
    double *t_x, *t_y, *t_z;
    if ((rc = tmpUpload(&t_eoc, w->edgesOnCell)) || (rc = tmpUpload(&t_coe, w->cellsOnEdge)) ||
        (rc = tmpUpload(&t_voe, w->verticesOnEdge)) || (rc = tmpUpload(&t_eov, w->edgesOnVertex)) ||
        (rc = tmpUpload(&t_cinv, c->cellInv)) || (rc = tmpUpload(&t_vinv, c->vertInv)) ||
        (rc = tmpUpload(&t_x, w->xEdge)) || (rc = tmpUpload(&t_y, w->yEdge)) || (rc = tmpUpload(&t_z, w->zEdge)))
        return rc;
    a.eoc = t_eoc;
    a.coe = t_coe;
    a.voe = t_voe;
    a.eov = t_eov;
    a.cellInv = t_cinv;
    a.vertInv = t_vinv;
    a.xE = t_x;
    a.yE = t_y;
    a.zE = t_z;
    hipLaunchKernelGGL(k_weak_init_cells, dim3((nC + 255) / 256), dim3(256), 0, c->stream, a);
    hipLaunchKernelGGL(k_weak_init_vertices, dim3((nV + 255) / 256), dim3(256), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    for (void *p : {(void *)t_eoc, (void *)t_coe, (void *)t_voe, (void *)t_eov, (void *)t_cinv, (void *)t_vinv,
                    (void *)t_x, (void *)t_y, (void *)t_z})
        if (p) (void)hipFree(p);
    w->ready = true;
    return SIH_OK;
}

// pointers and scalars of the time-step kernels; called after sih_bind_dev
void sih_weak_bind(sih_ctx *c)
{
    sih_weak_state *w = c->weak;
    auto fi = [&](const char *n) { return (int *)sih_find_field(c, n)->d; };
    auto fd = [&](const char *n) { return (double *)sih_find_field(c, n)->d; };
    WeakDev &d = w->d;
    d.nCells = c->nCells;
    d.nCellsSolve = c->nCellsSolve;
    d.nVertices = c->nVertices;
    d.nVerticesSolve = c->nVerticesSolve;
    d.M = c->M;
    d.padC = c->padC;
    d.padV = c->padV;
    d.lastOwnedCell = c->nCellsSolve > 0 ? c->cellInv[c->nCellsSolve - 1] : 0;
    d.nEdgesOnCell = fi("_nEdgesOnCell");
    d.voc = fi("_voc");
    d.cov = fi("_cov");
    d.solveStress = fi("solveStress");
    d.solveVelocity = fi("solveVelocity");
    d.interiorVertex = fi("interiorVertex");
    d.ev1 = fi("_weakEv1");
    d.ev2 = fi("_weakEv2");
    d.ec1 = fi("_weakEc1");
    d.ec2 = fi("_weakEc2");
    d.dvOnCell = fd("_weakDv");
    d.dcOnVertex = fd("_weakDc");
    d.areaCell = fd("areaCell");
    d.areaTriangle = fd("areaTriangle");
    d.icePressure = fd("icePressure");
    d.nPoly = fd("normalVectorPolygon");
    d.nTri = fd("normalVectorTriangle");
    d.latCell = fd("latCellRotated");
    d.latVertex = fd("latVertexRotated");
    d.strain11 = fd("strain11weak");
    d.strain22 = fd("strain22weak");
    d.strain12 = fd("strain12weak");
    d.stress11 = fd("stress11weak");
    d.stress22 = fd("stress22weak");
    d.stress12 = fd("stress12weak");
    d.Prep = fd("replacementPressureWeak");
    d.prin1 = fd("principalStress1Weak");
    d.prin2 = fd("principalStress2Weak");
    d.u = c->dev.u;
    d.v = c->dev.v;
    d.divU = c->dev.divU;
    d.divV = c->dev.divV;
    d.divergence = c->dev.divergence;
    d.shear = c->dev.shear;
    d.ridgeConv = c->dev.ridgeConv;
    d.ridgeShear = c->dev.ridgeShear;
    // "planar cases with zero sphere radius" (velocity_solver_weak.F:186-188)
    d.sphereRadius = (c->onASphere && c->R != 0.0) ? c->R : 1.0;
    d.dtE = c->dev.dtE;
    d.dampT = c->dev.dampT;
    d.relation = c->dev.relation;
    d.useColumn = c->dev.useColumn;
    d.sv11 = fd("strain11Vertex");
    d.sv22 = fd("strain22Vertex");
    d.sv12 = fd("strain12Vertex");
    d.var11 = c->dev.strain11;
    d.var22 = c->dev.strain22;
    d.var12 = c->dev.strain12;
    d.gidx = c->dev.gidx;
}

void sih_weak_launch_pre(sih_ctx *c)
{
    hipLaunchKernelGGL(k_weak_pre, dim3((c->nCells + 255) / 256), dim3(256), 0, c->stream, c->weak->d);
}

// weak strain, then the variational pool's corner strains from it; the caller follows with the variational
// stress / partial-sum pass that starts from stored strains
void sih_weak_launch_hybrid_strain(sih_ctx *c)
{
    const dim3 b(256), gc((c->nCells + 255) / 256), gv((c->nVerticesSolve + 255) / 256);
    hipLaunchKernelGGL(k_weak_cells, gc, b, 0, c->stream, c->weak->d, 1);
    hipLaunchKernelGGL(k_hybrid_vertex_strain, gv, b, 0, c->stream, c->weak->d);
    hipLaunchKernelGGL(k_hybrid_fill_corners, gc, b, 0, c->stream, c->weak->d);
}

void sih_weak_launch_cells(sih_ctx *c)
{
    hipLaunchKernelGGL(k_weak_cells, dim3((c->nCells + 255) / 256), dim3(256), 0, c->stream, c->weak->d, 0);
    hipLaunchKernelGGL(k_weak_divergence, dim3((c->nVertices + 255) / 256), dim3(256), 0, c->stream, c->weak->d);
}

void sih_weak_launch_post(sih_ctx *c)
{
    hipLaunchKernelGGL(k_weak_post, dim3((c->nCells + 255) / 256), dim3(256), 0, c->stream, c->weak->d);
}

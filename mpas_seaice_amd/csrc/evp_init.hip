// One-time initialisation of the variational EVP operators on the device:
// metric terms, cellVerticesAtVertex, local tangent-plane coordinates, Wachspress basis
// gradients and integrals, variational denominator, interior vertices.
//
// Each kernel evaluates the same expressions, in the same order, as the reference routine named
// above it; the file is compiled with -ffp-contract=off so every operation rounds once.
#include <cmath>
#include <cstring>

#include "sih_internal.h"

SihField *sih_find_field(sih_ctx *c, const char *name);

// the Dunavant and Fekete integration rules as a table of numbers (get_integration_factors_dunavant / _fekete,
// src/shared/mpas_seaice_velocity_solver_wachspress.F:1401-1941); normalisation factor 2 for all of them (:1426, :1751)
#include "wachspress_rules.h"

static const sih_rule_entry *findRule(int kind, int order)
{
    for (int r = 0; r < SIH_RULE_COUNT; r++)
        if (sih_rule_table[r].kind == kind && sih_rule_table[r].order == order) return &sih_rule_table[r];
    return nullptr;
}

struct InitArgs {
    int nCells, nVertices, nVerticesSolve, M, D, padC, padV, rotate, onASphere, includeMetric, alternateDenominator;
    int computeHalo;
    // integration rule on the unit triangle: points (qU, qV), weights qW, sum divided by qNorm
    int nQ;
    double qNorm;
    const double *qU, *qV, *qW;
    double R;
    const int *nEdgesOnCell, *voc, *cov;
    const double *xCell, *yCell, *zCell, *xVertex, *yVertex, *zVertex, *areaTriangle;
    const double *areaCell, *dvOnCell;  // dvOnCell: dvEdge(edgesOnCell(s, c)), null if the mesh came without edges
    int *cvav, *gidx, *interiorVertex, *vActive;
    double *tanLat, *varDenom;
    double *gradU, *gradV, *SU, *SV, *SM, *gradU3, *gradV3;
    int *flag;
    int *nEdgesDerive;  // nEdgesOnCell | 256 where every corner's own gradient is minus the sum of the other two
    int unmarkEvery;    // tests only (SIH_TEST_UNMARK_EVERY = n > 0): cells whose local index is a multiple of n get no mark
};

// mesh.F:2350-2381
__device__ inline void rotateForward(double &xp, double &yp, double &zp, double x, double y, double z, int rotate)
{
    if (rotate) {
        xp = -z;
        yp = y;
        zp = x;
    } else {
        xp = x;
        yp = y;
        zp = z;
    }
}

// variational_shared.F:293-358; interior_vertices mesh.F:423-488; seaice_cell_vertices_at_vertex mesh.F:632-685
__global__ void k_init_vertices(InitArgs a)
{
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= a.nVertices) return;
    if (a.includeMetric) {
        double xr, yr, zr;
        rotateForward(xr, yr, zr, a.xVertex[v], a.yVertex[v], a.zVertex[v], a.rotate);
        double lat = asin(zr / a.R);
        a.tanLat[v] = tan(lat) / a.R;
    } else {
        a.tanLat[v] = 0.0;
    }
    int nInterior = 0;
    for (int k = 0; k < a.D; k++) {
        int c = a.cov[k * a.padV + v];
        int j = -1;
        if (c < a.nCells) {
            nInterior++;
            int E = a.nEdgesOnCell[c];
            for (int s = 0; s < E; s++)
                if (a.voc[s * a.padC + c] == v) j = s;
        }
        a.cvav[k * a.padV + v] = j + 1;  // kept 1-based like the reference's field (0 = not found)
        // where the divergence partial sum of this (cell, corner) lives; the zero slot otherwise
        a.gidx[k * a.padV + v] = (j >= 0) ? j * a.padC + c : a.nCells;
    }
    // vertices the vertex loops cover: the owned ones (the reference's 1..nVerticesSolve) and, on request,
    // halo vertices that see all of their cells inside the block
    const int active = (v < a.nVerticesSolve) || (a.computeHalo && nInterior == a.D);
    a.vActive[v] = active ? 1 : 0;
    if (active) a.interiorVertex[v] = (nInterior == a.D) ? 1 : 0;
}

// ---- Wachspress basis on one polygon (wachspress.F:535-1069) --------------------------------
struct WachCell {
    int n;
    double x[SIH_MAXE], y[SIH_MAXE], A[SIH_MAXE], B[SIH_MAXE], kappa[SIH_MAXE];
};

__device__ inline int wrapIdx(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

// The functions below take the polygon's number of vertices as a template parameter N (4 to 7: every loop bound and
// every index into the polygon's arrays is then a constant, and the arrays are registers; with run-time bounds they
// lived in scratch memory and the integrals of QU60 took 29 ms) or N = 0 for any other polygon (bounds from w.n).
#define WACH_N(w) (N ? N : (w).n)

// variational_shared.F:191-279 with mesh.F:2021-2061, 2272-2332 (sphere) or :125-177 (plane);
// then calc_wachspress_coefficients wachspress.F:535-614
template <int N>
__device__ __forceinline__ void wachSetup(WachCell &w, const InitArgs &a, int c)
{
    const int n = N ? N : a.nEdgesOnCell[c];
    w.n = n;
    if (a.onASphere) {
        double xc, yc, zc;
        rotateForward(xc, yc, zc, a.xCell[c], a.yCell[c], a.zCell[c], a.rotate);
        double e0 = -yc, e1 = xc, e2 = 0.0;
        double mag = sqrt(e0 * e0 + e1 * e1 + e2 * e2);
        e0 = e0 / mag;
        e1 = e1 / mag;
        e2 = e2 / mag;
        double n0, n1, n2;
        if (zc != 0.0) {
            n0 = -xc;
            n1 = -yc;
            n2 = (xc * xc + yc * yc) / zc;
            mag = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
            n0 = n0 / mag;
            n1 = n1 / mag;
            n2 = n2 / mag;
            if (zc < 0.0) {
                n0 = -n0;
                n1 = -n1;
                n2 = -n2;
            }
        } else {
            n0 = 0.0;
            n1 = 0.0;
            n2 = 1.0;
        }
#pragma unroll
        for (int i = 0; i < n; i++) {
            int v = a.voc[i * a.padC + c];
            double px, py, pz;
            rotateForward(px, py, pz, a.xVertex[v], a.yVertex[v], a.zVertex[v], a.rotate);
            w.x[i] = px * e0 + py * e1 + pz * e2;
            w.y[i] = px * n0 + py * n1 + pz * n2;
        }
    } else {
#pragma unroll
        for (int i = 0; i < n; i++) {
            int v = a.voc[i * a.padC + c];
            w.x[i] = a.xVertex[v] - a.xCell[c];
            w.y[i] = a.yVertex[v] - a.yCell[c];
        }
    }
#pragma unroll
    for (int i = 0; i < n; i++) {
        int i1 = wrapIdx(i - 1, n), i2 = i;
        double den = w.x[i1] * w.y[i2] - w.x[i2] * w.y[i1];
        w.A[i] = (w.y[i2] - w.y[i1]) / den;
        w.B[i] = (w.x[i1] - w.x[i2]) / (w.x[i1] * w.y[i2] - w.x[i2] * w.y[i1]);
    }
    w.kappa[0] = 1.0;
#pragma unroll
    for (int j = 1; j < n; j++) {
        int i0 = j - 1, i1 = j, i2 = wrapIdx(j + 1, n);
        w.kappa[j] = w.kappa[j - 1] * (w.A[i2] * (w.x[i0] - w.x[i1]) + w.B[i2] * (w.y[i0] - w.y[i1])) /
                     (w.A[i0] * (w.x[i1] - w.x[i0]) + w.B[i0] * (w.y[i1] - w.y[i0]));
    }
}

__device__ __forceinline__ double wachEdge(const WachCell &w, int k, double x, double y) { return 1.0 - w.A[k] * x - w.B[k] * y; }

// numerator j = kappa_j * prod over edges k not in {j, j+1}, ascending k (wachspress.F:628-668, 864-925); j is a
// loop constant of the caller
template <int N>
__device__ __forceinline__ double wachNumerator(const WachCell &w, int j, double x, double y)
{
    const int n = WACH_N(w);
    int skip2 = wrapIdx(j + 1, n);
    double num = 1.0;
#pragma unroll
    for (int k = 0; k < n; k++)
        if (k != j && k != skip2) num = num * wachEdge(w, k, x, y);
    return num * w.kappa[j];
}

// wachspress.F:939-1035
template <int N>
__device__ __forceinline__ void wachNumeratorDerivative(const WachCell &w, int j, double x, double y, double &dU, double &dV)
{
    const int n = WACH_N(w);
    int skip2 = wrapIdx(j + 1, n);
    double sumU = 0.0, sumV = 0.0;
#pragma unroll
    for (int k = 0; k < n; k++) {
        if (k == j || k == skip2) continue;
        double pU = 1.0, pV = 1.0;
#pragma unroll
        for (int l = 0; l < k; l++) {
            if (l == j || l == skip2) continue;
            double e = wachEdge(w, l, x, y);
            pU = pU * e;
            pV = pV * e;
        }
        pU = pU * (-w.A[k]);
        pV = pV * (-w.B[k]);
#pragma unroll
        for (int l = k + 1; l < n; l++) {
            if (l == j || l == skip2) continue;
            double e = wachEdge(w, l, x, y);
            pU = pU * e;
            pV = pV * e;
        }
        sumU = sumU + pU;
        sumV = sumV + pV;
    }
    dU = sumU * w.kappa[j];
    dV = sumV * w.kappa[j];
}

// wachspress.F:763-850; i is a run-time index
template <int N>
__device__ __forceinline__ void wachBasisDerivative(const WachCell &w, int i, double x, double y, double &dU, double &dV)
{
    const int n = WACH_N(w);
    double den = 0.0, sU = 0.0, sV = 0.0, numi = 0.0, dUi = 0.0, dVi = 0.0;
#pragma unroll
    for (int j = 0; j < n; j++) {
        double num = wachNumerator<N>(w, j, x, y);
        den = den + num;
        double du, dv;
        wachNumeratorDerivative<N>(w, j, x, y, du, dv);
        sU = sU + du;
        sV = sV + dv;
        if (j == i) {
            numi = num;
            dUi = du;
            dVi = dv;
        }
    }
    dU = dUi / den - (numi / (den * den)) * sU;
    dV = dVi / den - (numi / (den * den)) * sV;
}

// calculate_wachspress_derivatives, wachspress.F:1083-1206, for basis vertex ib of cell c
template <int N>
__device__ __forceinline__ void wachGradientsOf(const InitArgs &a, int c, int ib)
{
    const int M = a.M;
    WachCell w;
    wachSetup<N>(w, a, c);
    const int n = WACH_N(w);
    if (ib >= n) return;
    // the gradient of basis function ib at its own vertex and at the two beside it (it vanishes at the others); the
    // loop runs over all vertices with a constant index, so that no array of the polygon is indexed by a variable
    const int before = wrapIdx(ib - 1, n), after = wrapIdx(ib + 1, n);
#pragma unroll
    for (int g = 0; g < n; g++) {
        if (g != ib && g != before && g != after) continue;
        double du, dv;
        wachBasisDerivative<N>(w, ib, w.x[g], w.y[g], du, dv);
        a.gradU[(size_t)(g * M + ib) * a.padC + c] = du;
        a.gradV[(size_t)(g * M + ib) * a.padC + c] = dv;
    }
}

// one thread per (cell, basis vertex)
__global__ void __launch_bounds__(64) k_wach_gradients(InitArgs a)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int c = t % a.nCells, ib = t / a.nCells;
    if (ib >= a.M) return;
    const int M = a.M;
    for (int g = 0; g < M; g++) {  // host index (iBasis, iGradient) -> lead g*M + ib
        a.gradU[(size_t)(g * M + ib) * a.padC + c] = 0.0;
        a.gradV[(size_t)(g * M + ib) * a.padC + c] = 0.0;
    }
    switch (a.nEdgesOnCell[c]) {
    case 4: wachGradientsOf<4>(a, c, ib); break;
    case 5: wachGradientsOf<5>(a, c, ib); break;
    case 6: wachGradientsOf<6>(a, c, ib); break;
    case 7: wachGradientsOf<7>(a, c, ib); break;
    default: break;  // k_wach_gradients_any
    }
}

// The same for the polygons the kernel above leaves out (triangles, octagons and beyond), with run-time loop bounds.
// A kernel of its own, launched only on a mesh that has such cells: its arrays live in scratch memory, and a kernel
// that declares scratch memory is given some whether or not a wave takes that path.
__global__ void __launch_bounds__(64) k_wach_gradients_any(InitArgs a)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int c = t % a.nCells, ib = t / a.nCells;
    if (ib >= a.M) return;
    const int n = a.nEdgesOnCell[c];
    if (n >= 4 && n <= 7) return;
    wachGradientsOf<0>(a, c, ib);
}

// integrate_wachspress / integrate_wachspress_polygon, wachspress.F:179-467, for velocity vertex j of cell c:
// accumulates the column (:, j) of the three integrals
template <int N>
__device__ __forceinline__ void wachIntegralsOf(const InitArgs &a, int c, int j)
{
    constexpr int NA = N ? N : SIH_MAXE;
    const int M = a.M;
    double SU[NA], SV[NA], SM[NA];
#pragma unroll
    for (int s = 0; s < NA; s++) SU[s] = SV[s] = SM[s] = 0.0;
    WachCell w;
    wachSetup<N>(w, a, c);
    const int n = WACH_N(w);
    if (j < n) {
#pragma unroll
        for (int tri = 0; tri < n; tri++) {
            int i1 = tri, i2 = wrapIdx(tri + 1, n);
            // get_triangle_mapping (:485-517) from the unit triangle (1,0),(0,1)
            double x1 = 1.0, y1 = 0.0, x2 = 0.0, y2 = 1.0;
            double u1 = w.x[i1], v1 = w.y[i1], u2 = w.x[i2], v2 = w.y[i2];
            double m11 = (u2 * y1 - u1 * y2) / (x2 * y1 - x1 * y2);
            double m12 = (u1 * x2 - u2 * x1) / (y1 * x2 - y2 * x1);
            double m21 = (v2 * y1 - v1 * y2) / (x2 * y1 - x1 * y2);
            double m22 = (v1 * x2 - v2 * x1) / (y1 * x2 - y2 * x1);
            double jac = m11 * m22 - m12 * m21;
            double sU[NA], sV[NA], sM[NA];
#pragma unroll
            for (int s = 0; s < NA; s++) sU[s] = sV[s] = sM[s] = 0.0;
            for (int p = 0; p < a.nQ; p++) {
                double px = m11 * a.qU[p] + m12 * a.qV[p];
                double py = m21 * a.qU[p] + m22 * a.qV[p];
                double num[NA], den = 0.0, numj = 0.0;
#pragma unroll
                for (int s = 0; s < n; s++) {
                    num[s] = wachNumerator<N>(w, s, px, py);
                    den = den + num[s];
                    numj = s == j ? num[s] : numj;
                }
                double dNu, dNv;
                wachBasisDerivative<N>(w, j, px, py, dNu, dNv);
                double Nj = numj / den;
#pragma unroll
                for (int s = 0; s < n; s++) {
                    double tmp = jac * a.qW[p] * (num[s] / den);
                    sU[s] = sU[s] + tmp * dNu;
                    sV[s] = sV[s] + tmp * dNv;
                    sM[s] = sM[s] + tmp * Nj;
                }
            }
#pragma unroll
            for (int s = 0; s < n; s++) {
                SU[s] = SU[s] + sU[s] / a.qNorm;
                SV[s] = SV[s] + sV[s] / a.qNorm;
                SM[s] = SM[s] + sM[s] / a.qNorm;
            }
        }
    }
    // host index (iStress s, iVelocity j) -> lead j*M + s
#pragma unroll
    for (int s = 0; s < NA; s++)
        if (s < M) {
            size_t k = (size_t)(j * M + s) * a.padC + c;
            a.SU[k] = SU[s];
            a.SV[k] = SV[s];
            a.SM[k] = SM[s];
        }
    for (int s = NA; s < M; s++) {
        size_t k = (size_t)(j * M + s) * a.padC + c;
        a.SU[k] = 0.0;
        a.SV[k] = 0.0;
        a.SM[k] = 0.0;
    }
}

// one thread per (cell, velocity vertex j)
__global__ void __launch_bounds__(64) k_wach_integrals(InitArgs a)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int c = t % a.nCells, j = t / a.nCells;
    if (j >= a.M) return;
    switch (a.nEdgesOnCell[c]) {
    case 4: wachIntegralsOf<4>(a, c, j); break;
    case 5: wachIntegralsOf<5>(a, c, j); break;
    case 6: wachIntegralsOf<6>(a, c, j); break;
    case 7: wachIntegralsOf<7>(a, c, j); break;
    default: break;  // k_wach_integrals_any
    }
}

__global__ void __launch_bounds__(64) k_wach_integrals_any(InitArgs a)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int c = t % a.nCells, j = t / a.nCells;
    if (j >= a.M) return;
    const int n = a.nEdgesOnCell[c];
    if (n >= 4 && n <= 7) return;
    wachIntegralsOf<0>(a, c, j);
}

// ---- piecewise-linear basis (mpas_seaice_velocity_solver_pwl.F:44-373) ----------------------------
// seaice_solve_linear_basis_system (mpas_seaice_numerics.F:44-212) for the 3x3 systems of the sub-triangles:
// LU decomposition with implicitly scaled partial pivoting, then back substitution
__device__ void solveLinearBasisSystem(const double L[3][3], const double rhs[3], double sol[3])
{
    const int n = 3;
    double a[3][3], b[3], maxa[3];
    int indices[3];
    for (int i = 0; i < n; i++) {
        b[i] = rhs[i];
        for (int j = 0; j < n; j++) a[i][j] = L[i][j];
    }
    for (int i = 0; i < n; i++) {
        double m = 0.0;
        for (int j = 0; j < n; j++) m = fmax(m, fabs(a[i][j]));
        maxa[i] = 1.0 / m;
    }
    for (int j = 0; j < n; j++) {
        int jmax = j;
        double best = maxa[j] * fabs(a[j][j]);
        for (int i = j + 1; i < n; i++) {
            double v = maxa[i] * fabs(a[i][j]);
            if (v > best) {
                best = v;
                jmax = i;
            }
        }
        if (j != jmax) {
            for (int k = 0; k < n; k++) {
                double t = a[jmax][k];
                a[jmax][k] = a[j][k];
                a[j][k] = t;
            }
            maxa[jmax] = maxa[j];
        }
        indices[j] = jmax;
        if (a[j][j] == 0.0) a[j][j] = 1.0e-20;
        for (int i = j + 1; i < n; i++) a[i][j] = a[i][j] / a[j][j];
        for (int i = j + 1; i < n; i++)
            for (int k = j + 1; k < n; k++) a[i][k] = a[i][k] - a[i][j] * a[j][k];
    }
    int jj = -1;
    for (int i = 0; i < n; i++) {
        int k = indices[i];
        double sums = b[k];
        b[k] = b[i];
        if (jj >= 0) {
            double dp = 0.0;
            for (int q = jj; q <= i - 1; q++) dp = dp + a[i][q] * b[q];
            sums = sums - dp;
        } else if (sums != 0.0) {
            jj = i;
        }
        b[i] = sums;
    }
    for (int i = n - 1; i >= 0; i--) {
        double dp = 0.0;
        for (int q = i + 1; q < n; q++) dp = dp + a[i][q] * b[q];
        b[i] = (b[i] - dp) / a[i][i];
    }
    for (int i = 0; i < n; i++) sol[i] = b[i];
}

// one thread per cell: sub-triangle areas (Heron, rescaled to areaCell), the linear basis on every
// sub-triangle, the PWL gradients at the corners and the three integral matrices
__global__ void k_pwl_basis(InitArgs a)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nCells) return;
    const int M = a.M;
    const size_t pc = a.padC;
    for (int k = 0; k < M * M; k++) {
        a.gradU[(size_t)k * pc + c] = 0.0;
        a.gradV[(size_t)k * pc + c] = 0.0;
        a.SU[(size_t)k * pc + c] = 0.0;
        a.SV[(size_t)k * pc + c] = 0.0;
        a.SM[(size_t)k * pc + c] = 0.0;
    }
    WachCell w;  // only its local coordinates are used (seaice_calc_local_coords)
    wachSetup<0>(w, a, c);
    const int n = w.n;
    if (n < 3) return;
    const double alpha = 1.0 / (double)n;
    double xc = 0.0, yc = 0.0;
    for (int i = 0; i < n; i++) {
        xc = xc + alpha * w.x[i];
        yc = yc + alpha * w.y[i];
    }
    double subArea[SIH_MAXE], subGU[SIH_MAXE][3], subGV[SIH_MAXE][3];
    double areaSum = 0.0;
    for (int s = 0; s < n; s++) {
        const int i2 = wrapIdx(s + 1, n);
        const double cc = a.dvOnCell[(size_t)s * pc + c];
        const double dx1 = w.x[s] - xc, dy1 = w.y[s] - yc, dx2 = w.x[i2] - xc, dy2 = w.y[i2] - yc;
        const double aa = sqrt(dx1 * dx1 + dy1 * dy1), bb = sqrt(dx2 * dx2 + dy2 * dy2);
        const double sp = (aa + bb + cc) * 0.5;
        subArea[s] = sqrt(sp * (sp - aa) * (sp - bb) * (sp - cc));
        areaSum = areaSum + subArea[s];
    }
    const double scale = a.areaCell[c] / areaSum;
    for (int s = 0; s < n; s++) subArea[s] = subArea[s] * scale;
    for (int s = 0; s < n; s++) {
        const int i2 = wrapIdx(s + 1, n);
        const double L[3][3] = {{w.x[s] - xc, w.y[s] - yc, 1.0}, {w.x[i2] - xc, w.y[i2] - yc, 1.0}, {0.0, 0.0, 1.0}};
        const double r1[3] = {1.0, 0.0, 0.0}, r2[3] = {0.0, 1.0, 0.0};
        double sol[3];
        solveLinearBasisSystem(L, r1, sol);
        subGU[s][0] = sol[0];
        subGV[s][0] = sol[1];
        solveLinearBasisSystem(L, r2, sol);
        subGU[s][1] = sol[0];
        subGV[s][1] = sol[1];
        subGU[s][2] = -subGU[s][0] - subGU[s][1];
        subGV[s][2] = -subGV[s][0] - subGV[s][1];
    }
    // gradient of basis function ib on sub-triangle s (subCellGradientU/V(iBasisVertex, iSubCell))
    auto cellG = [&](const double (*g)[3], int ib, int s) {
        double v = g[s][2] * alpha;
        if (s == ib)
            v = v + g[s][0];
        else if (s == wrapIdx(ib - 1, n))
            v = v + g[s][1];
        return v;
    };
    for (int ib = 0; ib < n; ib++)
        for (int g = 0; g < n; g++) {
            const int s2 = wrapIdx(g - 1, n);
            a.gradU[(size_t)(g * M + ib) * pc + c] = 0.5 * (cellG(subGU, ib, g) + cellG(subGU, ib, s2));
            a.gradV[(size_t)(g * M + ib) * pc + c] = 0.5 * (cellG(subGV, ib, g) + cellG(subGV, ib, s2));
        }
    for (int is = 0; is < n; is++)
        for (int iv = 0; iv < n; iv++) {
            double su = 0.0, sv = 0.0, sm = 0.0;
            for (int s = 0; s < n; s++) {
                const bool nearS = (s == is) || (s == wrapIdx(is - 1, n));
                const double bi = nearS ? ((alpha + 1) * subArea[s]) / 3.0 : (alpha * subArea[s]) / 3.0;
                su = su + cellG(subGU, iv, s) * bi;
                sv = sv + cellG(subGV, iv, s) * bi;
                const int ts = (s == is) ? 1 : (s == wrapIdx(is - 1, n) ? 2 : 3);
                const int tv = (s == iv) ? 1 : (s == wrapIdx(iv - 1, n) ? 2 : 3);
                double m;
                if ((ts == 1 && tv == 1) || (ts == 2 && tv == 2))
                    m = 2.0 * (alpha * alpha) + 2.0 * alpha + 2.0;
                else if ((ts == 1 && tv == 2) || (ts == 2 && tv == 1))
                    m = 2.0 * (alpha * alpha) + 2.0 * alpha + 1.0;
                else if (ts == 3 && tv == 3)
                    m = 2.0 * (alpha * alpha);
                else
                    m = 2.0 * (alpha * alpha) + alpha;
                m = m * subArea[s] / 12.0;
                sm = sm + m;
            }
            const size_t k = (size_t)(iv * M + is) * pc + c;
            a.SU[k] = su;
            a.SV[k] = sv;
            a.SM[k] = sm;
        }
}

// Compact form of the basis gradients.  The Wachspress gradient matrix (iBasis, iGradient) is
// non-zero only for iBasis in {iGradient-1, iGradient, iGradient+1} (cyclic; wachspress.F:1178-1191),
// so corner g needs three coefficients.  They are stored in ascending-iBasis order, the order in
// which the reference's dense loop meets them, so that the strain sums round identically.
// flag[0] is raised if any other entry is non-zero (then the dense kernel must be used).
__global__ void k_pack_gradients(InitArgs a)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nCells) return;
    const int M = a.M;
    int E = a.nEdgesOnCell[c];
    bool dense = false, noUnity = false;
    for (int g = 0; g < M; g++) {
        int b3[3];
        if (g < E) {
            int p = wrapIdx(g - 1, E), n = wrapIdx(g + 1, E);
            if (g == 0) {
                b3[0] = 0;
                b3[1] = n;
                b3[2] = p;
            } else if (g == E - 1) {
                b3[0] = n;
                b3[1] = p;
                b3[2] = g;
            } else {
                b3[0] = p;
                b3[1] = g;
                b3[2] = n;
            }
        }
        for (int q = 0; q < 3; q++) {
            double gu = 0.0, gv = 0.0;
            if (g < E) {
                gu = a.gradU[(size_t)(g * M + b3[q]) * a.padC + c];
                gv = a.gradV[(size_t)(g * M + b3[q]) * a.padC + c];
            }
            a.gradU3[(size_t)(q * M + g) * a.padC + c] = gu;
            a.gradV3[(size_t)(q * M + g) * a.padC + c] = gv;
        }
        if (g < E) {
            for (int b = 0; b < E; b++) {
                if (b == b3[0] || b == b3[1] || b == b3[2]) continue;
                if (a.gradU[(size_t)(g * M + b) * a.padC + c] != 0.0 || a.gradV[(size_t)(g * M + b) * a.padC + c] != 0.0)
                    dense = true;
            }
            // The basis functions sum to one, so the three gradients met at corner g sum to zero: the corner's own
            // (slot 0 for g = 0, slot 2 for the last corner, else slot 1) is minus the sum of the other two up to
            // rounding.  A cell where that holds at every corner is marked, and the subcycle kernels do not fetch the
            // own gradients of its corners.  (Decided cell by cell, from the cell's own numbers: the same in every
            // block that holds the cell, so that results do not depend on the partition.)
            const int own = g == 0 ? 0 : (g == E - 1 ? 2 : 1);
            for (int uv = 0; uv < 2; uv++) {
                const double *g3 = uv ? a.gradV3 : a.gradU3;
                double x[3];
                for (int q = 0; q < 3; q++) x[q] = g3[(size_t)(q * M + g) * a.padC + c];
                const double o1 = x[own == 0 ? 1 : 0], o2 = x[own == 2 ? 1 : 2];
                const double derived = -(o1 + o2);
                const double big = fmax(fabs(x[own]), fmax(fabs(o1), fabs(o2)));
                // (the gradients of a distorted heptagon miss zero by 160 units in the last place of the largest: the
                // cancellation inside the Wachspress quotient; 1e-13 is three orders of magnitude inside the parity bar)
                if (!(fabs(derived - x[own]) <= 1.0e-13 * big)) noUnity = true;
            }
        }
    }
    if (a.unmarkEvery > 0 && c % a.unmarkEvery == 0) noUnity = true;
    a.nEdgesDerive[c] = E | ((!dense && !noUnity) ? 256 : 0);
    if (dense) atomicOr(a.flag, 1);
}

// variational_denominator, variational.F:358-445
__global__ void k_var_denominator(InitArgs a)
{
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= a.nVertices) return;
    if (a.alternateDenominator) {
        double d = 0.0;
        for (int k = 0; k < a.D; k++) {
            int c = a.cov[k * a.padV + v];
            int j = a.cvav[k * a.padV + v] - 1;
            if (c >= a.nCells || j < 0) continue;
            int E = a.nEdgesOnCell[c];
            for (int s = 0; s < E; s++) d = d + a.SM[(size_t)(j * a.M + s) * a.padC + c];
        }
        a.varDenom[v] = d;
    } else {
        a.varDenom[v] = a.areaTriangle[v];
    }
}

__global__ void k_zero_int(int *p, int n)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) p[t] = 0;
}

void sih_bind_dev(sih_ctx *c);

static InitArgs makeInitArgs(sih_ctx *c)
{
    auto fi = [&](const char *n) { return (int *)sih_find_field(c, n)->d; };
    auto fd = [&](const char *n) { return (double *)sih_find_field(c, n)->d; };
    InitArgs a{};
    a.nCells = c->nCells;
    a.nVertices = c->nVertices;
    a.nVerticesSolve = c->nVerticesSolve;
    a.M = c->M;
    a.D = c->D;
    a.padC = c->padC;
    a.padV = c->padV;
    a.rotate = c->cfg.rotateCartesianGrid;
    a.onASphere = c->onASphere;
    a.includeMetric = (c->cfg.includeMetricTerms && c->onASphere) ? 1 : 0;
    a.alternateDenominator = c->cfg.variationalDenominator == 1;
    a.R = c->R;
    a.nEdgesOnCell = fi("_nEdgesOnCell");
    a.voc = fi("_voc");
    a.cov = fi("_cov");
    a.xCell = fd("xCell");
    a.yCell = fd("yCell");
    a.zCell = fd("zCell");
    a.xVertex = fd("xVertex");
    a.yVertex = fd("yVertex");
    a.zVertex = fd("zVertex");
    a.areaTriangle = fd("areaTriangle");
    a.areaCell = fd("areaCell");
    a.dvOnCell = sih_find_field(c, "_dvEdgeOnCell") ? fd("_dvEdgeOnCell") : nullptr;
    a.cvav = fi("cellVerticesAtVertex");
    a.gidx = fi("_gidx");
    a.interiorVertex = fi("interiorVertex");
    a.vActive = fi("_vertexActive");
    a.computeHalo = c->cfg.computeHaloVertices ? 1 : 0;
    a.tanLat = fd("tanLatVertexRotatedOverRadius");
    a.varDenom = fd("variationalDenominator");
    a.gradU = fd("basisGradientU");
    a.gradV = fd("basisGradientV");
    a.SU = fd("basisIntegralsU");
    a.SV = fd("basisIntegralsV");
    a.SM = fd("basisIntegralsMetric");
    a.nEdgesDerive = (int *)sih_find_field(c, "_nEdgesDerive")->d;
    a.gradU3 = fd("_gradU3");
    a.gradV3 = fd("_gradV3");
    a.flag = c->d_flag;
    if (const char *e = getenv("SIH_TEST_UNMARK_EVERY")) a.unmarkEvery = atoi(e);
    return a;
}

// basisIntegralsMetric(s, j) = integral of N_s N_j over the cell is symmetric in exact arithmetic; the reference
// computes all M x M entries (wachspress.F:430-434: (jac w N_s) N_j and (jac w N_j) N_s round differently in the last
// place).  Mirroring the entries on one side of the diagonal makes the matrix symmetric bit for bit, and the cell
// kernels then fetch 21 instead of 36 values per hexagon every subcycle.  (s, j) sits at plane j * M + s; kept: s <= j.
__global__ void k_mirror_metric(InitArgs a)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nCells) return;
    const int M = a.M, E = a.nEdgesOnCell[c];
    for (int j = 0; j < E; j++)
        for (int s = j + 1; s < E; s++) a.SM[(size_t)(j * M + s) * a.padC + c] = a.SM[(size_t)(s * M + j) * a.padC + c];
}

// flag bit 1 (value 2) is raised if any cell's metric matrix is not symmetric bit for bit
__global__ void k_check_metric_symmetry(InitArgs a)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nCells) return;
    const int M = a.M, E = a.nEdgesOnCell[c];
    bool differs = false;
    for (int j = 0; j < E; j++)
        for (int s = j + 1; s < E; s++) {
            const double x = a.SM[(size_t)(j * M + s) * a.padC + c], y = a.SM[(size_t)(s * M + j) * a.padC + c];
            if (__double_as_longlong(x) != __double_as_longlong(y)) differs = true;
        }
    if (differs) atomicOr(a.flag, 2);
}

// Where the corner's own gradient is derived from the other two, the compact copy holds the derived value too: the
// kernels that fetch all three (persistent and fused forms, one thread per cell) then compute with the same bits
// as the one that derives it.  basisGradientU / V themselves stay as computed.
__global__ void k_store_derived_gradients(InitArgs a)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nCells) return;
    const int M = a.M, E = a.nEdgesOnCell[c];
    if (!(a.nEdgesDerive[c] & 256)) return;
    for (int g = 0; g < E; g++) {
        const int own = g == 0 ? 0 : (g == E - 1 ? 2 : 1);
        for (int uv = 0; uv < 2; uv++) {
            double *g3 = uv ? a.gradV3 : a.gradU3;
            double x[3];
            for (int q = 0; q < 3; q++) x[q] = g3[(size_t)(q * M + g) * a.padC + c];
            g3[(size_t)(own * M + g) * a.padC + c] = -(x[own == 0 ? 1 : 0] + x[own == 2 ? 1 : 2]);
        }
    }
}

// basisIntegralsU(s, j) = integral of N_s dN_j/dx (wachspress.F:436-441; plane j * M + s here): the basis functions sum to
// one, so for every stress corner s the entries of the velocity corners j sum to zero up to rounding.  In a cell where
// that holds for U and V and every s, the row of the first velocity corner is replaced by minus the sum of the other
// rows, added in ascending j -- the value changes in its last places at most -- and the cell is marked (bit 9 of
// nEdgesDerive).  The split cell kernel then forms that row on the chip from what the cell's other corners fetch,
// with the same additions, and every other kernel computes with the same bits from the stored row.  (Cell by cell,
// from the cell's own numbers: the same in every block that holds the cell.)
__global__ void k_derive_integral_rows(InitArgs a)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nCells) return;
    const int M = a.M, E = a.nEdgesOnCell[c];
    if (E < 3 || (a.unmarkEvery > 0 && c % a.unmarkEvery == 0)) return;
    bool ok = true;
    for (int uv = 0; uv < 2; uv++) {
        const double *S = uv ? a.SV : a.SU;
        for (int s = 0; s < E; s++) {
            double acc = S[(size_t)(1 * M + s) * a.padC + c], big = fabs(acc);
            for (int j = 2; j < E; j++) {
                const double x = S[(size_t)(j * M + s) * a.padC + c];
                acc = acc + x;
                big = fmax(big, fabs(x));
            }
            const double own = S[(size_t)s * a.padC + c];
            big = fmax(big, fabs(own));
            if (!(fabs(-acc - own) <= 1.0e-13 * big)) ok = false;
        }
    }
    if (!ok) return;
    a.nEdgesDerive[c] |= 512;
}

static int packStatic(sih_ctx *c, const InitArgs &a, bool ownArrays)
{
    hipLaunchKernelGGL(k_zero_int, dim3(1), dim3(64), 0, c->stream, c->d_flag, 16);
    hipLaunchKernelGGL(k_pack_gradients, dim3((c->nCells + 255) / 256), dim3(256), 0, c->stream, a);
    hipLaunchKernelGGL(k_check_metric_symmetry, dim3((c->nCells + 255) / 256), dim3(256), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    int flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->sparseGrad = (flag & 1) == 0;
    c->symMetric = (flag & 2) == 0;
    {
        // SIH_BASIS_GRADIENTS=as_computed: all three gradients of a corner are fetched as stored
        const char *e = getenv("SIH_BASIS_GRADIENTS");
        c->deriveGrad = c->sparseGrad && !(e && strcmp(e, "as_computed") == 0);
    }
    if (c->deriveGrad) {
        hipLaunchKernelGGL(k_store_derived_gradients, dim3((c->nCells + 255) / 256), dim3(256), 0, c->stream, a);
        HIPCHK(hipGetLastError());
    }
    {
        // Only arrays the set-up has just computed itself are touched (not what a host has put since).
        // SIH_BASIS_INTEGRALS=as_computed: every row of basisIntegralsU / V stays as the reference's arithmetic gives it
        const char *e = getenv("SIH_BASIS_INTEGRALS");
        c->deriveRow = ownArrays && c->cfg.strainScheme == 0 && c->cfg.variationalBasis != 0 &&
                       !(e && strcmp(e, "as_computed") == 0);
    }
    if (c->deriveRow) {
        hipLaunchKernelGGL(k_derive_integral_rows, dim3((c->nCells + 255) / 256), dim3(256), 0, c->stream, a);
        HIPCHK(hipGetLastError());
    }
    return SIH_OK;
}

extern "C" int sih_evp_init(sih_ctx *c, const sih_evp_config *cfg)
{
    if (!c || !cfg) {
        sih_set_error("sih_evp_init: null argument");
        return SIH_ERR_ARG;
    }
    if (c->D != 3 && (cfg->strainScheme != 0 || cfg->stressDivergenceScheme != 0 || cfg->computeHaloVertices)) {
        sih_set_error("velocity solver on a mesh with vertexDegree 4: the variational scheme on one block only");
        return SIH_ERR_UNSUPPORTED;
    }
    if (cfg->constitutiveRelation < 1 || cfg->constitutiveRelation > 3) {
        sih_set_error("config_constitutive_relation_type unknown");
        return SIH_ERR_ARG;
    }
    if (cfg->oceanStressType < 1 || cfg->oceanStressType > 2) {
        sih_set_error("config_ocean_stress_type unknown");
        return SIH_ERR_ARG;
    }
    if (cfg->variationalBasis < 0 || cfg->variationalBasis > 2) {
        sih_set_error("Unknown variational basis type");
        return SIH_ERR_ARG;
    }
    if (cfg->variationalBasis == 2 && !sih_find_field(c, "_dvEdgeOnCell")) {
        sih_set_error("the 'pwl' basis needs edgesOnCell and dvEdge in the mesh description");
        return SIH_ERR_ARG;
    }
    if (cfg->variationalBasis == 1) {
        // get_integration_factors, wachspress.F:1223-1285
        if (cfg->wachspressIntegrationType == 0 || cfg->wachspressIntegrationType == 1) {
            if (!findRule(1, cfg->wachspressIntegrationOrder)) {  // the reference's message, :1706
                sih_set_error("get_integration_factors_dunavant: Unimplemented integration order for Dunavant wachspress integration");
                return SIH_ERR_UNSUPPORTED;
            }
        } else if (cfg->wachspressIntegrationType == 3) {
            if (!findRule(3, cfg->wachspressIntegrationOrder)) {  // :1935
                sih_set_error("get_integration_factors_fekete: Unimplemented integration order for Fekete wachspress integration");
                return SIH_ERR_UNSUPPORTED;
            }
        } else if (cfg->wachspressIntegrationType == 2) {
            if (cfg->wachspressIntegrationOrder < 1 || cfg->wachspressIntegrationOrder > SIH_MAX_TRAPEZOID) {
                sih_set_error("trapezoidal integration: the order must lie between 1 and 21");
                return SIH_ERR_UNSUPPORTED;
            }
        } else {  // :1279
            sih_set_error("get_integration_factors: Unknown wachspress integration type");
            return SIH_ERR_UNSUPPORTED;
        }
    }
    if (cfg->variationalDenominator < 0 || cfg->variationalDenominator > 1) {
        sih_set_error("Unknown variational denominator type");
        return SIH_ERR_ARG;
    }
    if (cfg->strainScheme < 0 || cfg->strainScheme > 1 || cfg->stressDivergenceScheme < 0 ||
        cfg->stressDivergenceScheme > 1) {
        sih_set_error("config_strain_scheme / config_stress_divergence_scheme unknown");
        return SIH_ERR_ARG;
    }
    if (cfg->strainScheme == 0 && cfg->stressDivergenceScheme == 1) {  // the reference's message, velocity_solver.F:193
        sih_set_error("Cannot have variational strain scheme with variational stress divergence scheme");
        return SIH_ERR_ARG;
    }
    if (cfg->strainScheme == 1 && (cfg->computeHaloVertices || cfg->averageVariationalStrain)) {
        sih_set_error("the weak scheme has no redundant halo update and no strain averaging");
        return SIH_ERR_UNSUPPORTED;
    }
    if (cfg->computeHaloVertices && (cfg->averageVariationalStrain || cfg->useSpecialBoundariesVelocity)) {
        sih_set_error("computeHaloVertices: not together with strain averaging or special boundaries "
                      "(their loops cover owned or source vertices only)");
        return SIH_ERR_UNSUPPORTED;
    }
    if (cfg->dynamicsSubcycleNumber < 1 || cfg->elasticSubcycleNumber < 1 || !(cfg->dt > 0.0) || cfg->nCategories < 1) {
        sih_set_error("sih_evp_init: dt, subcycle numbers and nCategories must be positive");
        return SIH_ERR_ARG;
    }
    if (cfg->useColumnPackage && cfg->useColumnVerticalThermodynamics) {
        if (cfg->iceStrengthFormulation < 0 || cfg->iceStrengthFormulation > 1 || cfg->ridgingParticipation < 0 ||
            cfg->ridgingParticipation > 1 || cfg->ridgingRedistribution < 0 || cfg->ridgingRedistribution > 1) {
            sih_set_error("ice strength / ridging participation / redistribution option unknown");
            return SIH_ERR_ARG;
        }
        if (cfg->nCategories > SIH_MAXCAT) {
            sih_set_error("the column package's ice strength handles at most 32 thickness categories");
            return SIH_ERR_UNSUPPORTED;
        }
    }
    HIPCHK(hipSetDevice(c->device));
    c->cfg = *cfg;
    {
        int rc = sih_add_field(c, "openWaterArea", LOC_CELL, 1, false);
        if (rc) return rc;
    }
    // category tracers are sized by nCategories, known only now
    if (!sih_find_field(c, "iceAreaCategory")) {
        const char *cats[3] = {"iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory"};
        for (const char *n : cats) {
            int rc = sih_add_field(c, n, LOC_CELL, cfg->nCategories, false);
            if (rc) return rc;
        }
    } else if (sih_find_field(c, "iceAreaCategory")->lead != cfg->nCategories) {
        sih_set_error("sih_evp_init: nCategories cannot change after the first initialisation");
        return SIH_ERR_STATE;
    }
    InitArgs a = makeInitArgs(c);
    dim3 block(256);
    hipLaunchKernelGGL(k_init_vertices, dim3((c->nVertices + 255) / 256), block, 0, c->stream, a);
    if (cfg->variationalBasis == 1) {
        // the integration rule on the unit triangle (get_integration_factors, wachspress.F:1223-1285)
        std::vector<double> rule;  // [u..., v..., w...]
        int nQ;
        double qNorm;
        if (cfg->wachspressIntegrationType == 2) {  // get_integration_factors_trapezoidal :1301-1387
            const int n = cfg->wachspressIntegrationOrder;
            nQ = ((n + 1) * (n + 1) + (n + 1)) / 2;
            rule.assign((size_t)3 * nQ, 0.0);
            int ij = 0;
            for (int i = 0; i <= n; i++)
                for (int j = 0; j <= n - i; j++) {
                    rule[ij] = (double)i / (double)n;
                    rule[nQ + ij] = (double)j / (double)n;
                    double w;
                    if (i == n || j == n || (i == 0 && j == 0))
                        w = 1.0;
                    else if ((j == 0 && i != 0 && i != n) || (i == 0 && j != 0 && j != n) || (i == n - j && i != 0 && j != 0))
                        w = 3.0;
                    else
                        w = 6.0;
                    rule[2 * nQ + ij] = w;
                    ij++;
                }
            qNorm = 6.0 * ((double)n * (double)n);
        } else {
            const sih_rule_entry *r = findRule(cfg->wachspressIntegrationType == 3 ? 3 : 1, cfg->wachspressIntegrationOrder);
            nQ = r->n;
            rule.insert(rule.end(), sih_rule_u + r->first, sih_rule_u + r->first + nQ);
            rule.insert(rule.end(), sih_rule_v + r->first, sih_rule_v + r->first + nQ);
            rule.insert(rule.end(), sih_rule_w + r->first, sih_rule_w + r->first + nQ);
            qNorm = 2.0;
        }
        if (!c->d_rule) {
            int rc0 = sih_dev_alloc(c, (void **)&c->d_rule, sizeof(double) * 3 * SIH_MAX_RULE);
            if (rc0) return rc0;
        }
        HIPCHK(hipMemcpyAsync(c->d_rule, rule.data(), sizeof(double) * rule.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));  // `rule` is a local
        a.nQ = nQ;
        a.qNorm = qNorm;
        a.qU = c->d_rule;
        a.qV = c->d_rule + nQ;
        a.qW = c->d_rule + 2 * nQ;
        long tot = (long)c->nCells * c->M;
        hipLaunchKernelGGL(k_wach_gradients, dim3((unsigned)((tot + 63) / 64)), dim3(64), 0, c->stream, a);
        hipLaunchKernelGGL(k_wach_integrals, dim3((unsigned)((tot + 63) / 64)), dim3(64), 0, c->stream, a);
    } else if (cfg->variationalBasis == 2) {
        hipLaunchKernelGGL(k_pwl_basis, dim3((c->nCells + 63) / 64), dim3(64), 0, c->stream, a);
    }
    hipLaunchKernelGGL(k_var_denominator, dim3((c->nVertices + 255) / 256), block, 0, c->stream, a);
    {
        // SIH_METRIC_INTEGRALS=as_computed keeps all M x M metric integrals as the reference's arithmetic gives them
        const char *e = getenv("SIH_METRIC_INTEGRALS");
        const bool asComputed = e && strcmp(e, "as_computed") == 0;
        if (cfg->variationalBasis != 0 && !asComputed)
            hipLaunchKernelGGL(k_mirror_metric, dim3((c->nCells + 255) / 256), block, 0, c->stream, a);
    }
    HIPCHK(hipGetLastError());
    int rc = packStatic(c, a, true);
    if (rc) return rc;
    if (cfg->strainScheme == 1 && (rc = sih_weak_init(c))) return rc;
    // time steps and EVP constants: velocity_solver.F:155-157, constitutive_relation.F:125, 158-159
    double dtDyn = cfg->dt / (double)cfg->dynamicsSubcycleNumber;
    double dtE = dtDyn / (double)cfg->elasticSubcycleNumber;
    double dv = cfg->minEdgeLengthGlobal > 0.0 ? cfg->minEdgeLengthGlobal : c->minDvEdge;
    double gamma = 0.25 * 1.0e11 * dtDyn;
    c->inited = true;
    {
        // which cell kernel: one thread per (cell, corner) unless SIH_CELL_KERNEL=fused asks for the
        // one-thread-per-cell kernel (measured slower at every mesh size from 10 k to 164 k cells)
        const char *e = getenv("SIH_CELL_KERNEL");
        c->splitCells = !(e && strcmp(e, "fused") == 0);
    }
    sih_bind_dev(c);
    c->dev.dtDyn = dtDyn;
    c->dev.dtE = dtE;
    c->dev.dampT = 0.36 * dtDyn;
    c->dev.brlx = dv > 0.0 ? (2.0 * 0.86 * 5.5e-3 * gamma) / (dv * dv) : 0.0;
    if (cfg->strainScheme == 1) sih_weak_bind(c);
    if (cfg->constitutiveRelation == 2 && !(dv > 0.0)) {
        sih_set_error("evp_revised needs dvEdge (minEdgeLengthGlobal)");
        return SIH_ERR_ARG;
    }
    sih_drop_graphs(c);  // whatever was captured holds launches of the previous kernel variants
    return sih_persist_setup(c);  // tile tables of the persistent subcycle kernel, if the block qualifies
}

extern "C" int sih_evp_static_changed(sih_ctx *c)
{
    if (!c || !c->inited) {
        sih_set_error("sih_evp_static_changed: context not initialised");
        return SIH_ERR_STATE;
    }
    HIPCHK(hipSetDevice(c->device));
    InitArgs a = makeInitArgs(c);
    int rc = packStatic(c, a, false);
    sih_drop_graphs(c);  // whatever was captured holds launches of the previous kernel variants
    if (rc) return rc;
    return sih_persist_setup(c);
}

/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference's EVP velocity solver.
 *
 * Nothing in the product path may call into this file: only tests/, the smoke check and the
 * cpu_baseline leg of bench.py use it, as the checker / reported baseline.
 *
 * PARITY STATUS: "parity unpinned" at the bit level.  The reference ships no golden vectors
 * for this path and its Fortran cannot be built here (every source file uses the MPAS framework
 * modules mpass_derived_types / mpass_pool_routines / ..., which are not in the repository).
 * This restatement is pinned by (i) the analytic known-answer cases the reference's own test
 * scripts define (spherical-harmonic strain field, planar linear / sinusoid fields), and (ii)
 * invariants of the scheme (partition of unity, exactness for linear fields, area partition).
 *
 * Every routine follows the arithmetic of the reference routine cited above it, statement by
 * statement, in the same evaluation order.  Arrays use the reference's Fortran layout
 * (column-major, first index fastest) and its 1-based connectivity with the nX+1 sentinel.
 * Build with -ffp-contract=off so each source-level operation rounds once.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* 1-based Fortran-style accessors */
#define IX2(i, c, M) ((size_t)((c) - 1) * (size_t)(M) + (size_t)((i) - 1))
#define IX3(i, j, c, M) (((size_t)((c) - 1) * (size_t)(M) + (size_t)((j) - 1)) * (size_t)(M) + (size_t)((i) - 1))

/* constants: src/shared/mpas_seaice_velocity_solver.F:61-65,
 * src/shared/mpas_seaice_velocity_solver_constitutive_relation.F:41-47,
 * src/column/constants/cice/ice_constants_colpkg.F90 (rhoi, rhos, rhow, dragio, Pstar, Cstar) */
static const double sinOceanTurningAngle = 0.0;
static const double cosOceanTurningAngle = 1.0;
static const double seaiceAreaMinimum = 0.001;
static const double seaiceMassMinimum = 0.01;
static const double eccentricitySquared = 4.0;
static const double puny = 1.0e-11;
static const double seaiceDensityIce = 917.0;
static const double seaiceDensitySnow = 330.0;
static const double seaiceDensitySeaWater = 1026.0;
static const double seaiceIceOceanDragCoefficient = 0.00536;
static const double seaiceIceStrengthConstantHiblerP = 2.75e4;
static const double seaiceIceStrengthConstantHiblerC = 20.0;
static const double seaiceGravity = 9.80616;

int ora_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void ora_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

static double sign1(double x) { return copysign(1.0, x); } /* Fortran sign(1.0, x) */

static int wrapped_index(int input, int n)
{ /* variational_shared.F:372-385, modulo(input-1,n)+1 */
    int m = (input - 1) % n;
    if (m < 0) m += n;
    return m + 1;
}

/* ------------------------------------------------------------------ mesh helpers */

/* mpas_seaice_mesh.F:2350-2381 */
static void grid_rotation_forward(double *xp, double *yp, double *zp, double x, double y, double z, int rotate)
{
    if (rotate) {
        *xp = -z;
        *yp = y;
        *zp = x;
    } else {
        *xp = x;
        *yp = y;
        *zp = z;
    }
}

/* mpas_seaice_mesh.F:2272-2332 */
static void local_east_north(double xP, double yP, double zP, double e[3], double n[3])
{
    double mag;
    e[0] = -yP;
    e[1] = xP;
    e[2] = 0.0;
    mag = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    e[0] = e[0] / mag;
    e[1] = e[1] / mag;
    e[2] = e[2] / mag;
    if (zP != 0.0) {
        n[0] = -xP;
        n[1] = -yP;
        n[2] = (xP * xP + yP * yP) / zP;
        mag = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        n[0] = n[0] / mag;
        n[1] = n[1] / mag;
        n[2] = n[2] / mag;
        if (zP < 0.0) {
            n[0] = -n[0];
            n[1] = -n[1];
            n[2] = -n[2];
        }
    } else {
        n[0] = 0.0;
        n[1] = 0.0;
        n[2] = 1.0;
    }
}

/* mpas_seaice_velocity_solver_variational_shared.F:42-279 (+ mesh.F:2021-2061, 1758-1769) */
void ora_calc_local_coords(double *xLocal, double *yLocal, int nCells, int maxEdges, const int *nEdgesOnCell,
                           const int *verticesOnCell, const double *xVertex, const double *yVertex,
                           const double *zVertex, const double *xCell, const double *yCell, const double *zCell,
                           int rotate, int onASphere)
{
    for (int c = 1; c <= nCells; c++) {
        if (onASphere) {
            double xc, yc, zc, e[3], n[3];
            grid_rotation_forward(&xc, &yc, &zc, xCell[c - 1], yCell[c - 1], zCell[c - 1], rotate);
            for (int i = 1; i <= nEdgesOnCell[c - 1]; i++) {
                int v = verticesOnCell[IX2(i, c, maxEdges)];
                double p[3];
                grid_rotation_forward(&p[0], &p[1], &p[2], xVertex[v - 1], yVertex[v - 1], zVertex[v - 1], rotate);
                local_east_north(xc, yc, zc, e, n);
                xLocal[IX2(i, c, maxEdges)] = p[0] * e[0] + p[1] * e[1] + p[2] * e[2];
                yLocal[IX2(i, c, maxEdges)] = p[0] * n[0] + p[1] * n[1] + p[2] * n[2];
            }
        } else {
            for (int i = 1; i <= nEdgesOnCell[c - 1]; i++) {
                int v = verticesOnCell[IX2(i, c, maxEdges)];
                xLocal[IX2(i, c, maxEdges)] = xVertex[v - 1] - xCell[c - 1];
                yLocal[IX2(i, c, maxEdges)] = yVertex[v - 1] - yCell[c - 1];
            }
        }
    }
}

/* variational_shared.F:293-358 */
void ora_metric_terms(double *tanLatVertexRotatedOverRadius, int nVertices, const double *xVertex,
                      const double *yVertex, const double *zVertex, double sphereRadius, int rotate,
                      int includeMetricTerms)
{
    for (int v = 0; v < nVertices; v++) {
        if (includeMetricTerms) {
            double xr, yr, zr;
            grid_rotation_forward(&xr, &yr, &zr, xVertex[v], yVertex[v], zVertex[v], rotate);
            double lat = asin(zr / sphereRadius);
            tanLatVertexRotatedOverRadius[v] = tan(lat) / sphereRadius;
        } else {
            tanLatVertexRotatedOverRadius[v] = 0.0;
        }
    }
}

/* mpas_seaice_mesh.F:632-685 */
void ora_cell_vertices_at_vertex(int *cellVerticesAtVertex, int nVertices, int vertexDegree, int maxEdges,
                                 int nCells, const int *nEdgesOnCell, const int *verticesOnCell,
                                 const int *cellsOnVertex)
{
    for (int v = 1; v <= nVertices; v++) {
        for (int k = 1; k <= vertexDegree; k++) {
            cellVerticesAtVertex[IX2(k, v, vertexDegree)] = 0;
            int c = cellsOnVertex[IX2(k, v, vertexDegree)];
            if (c < 1 || c > nCells) continue; /* the reference reads the sentinel slot (nEdgesOnCell = 0 there) */
            for (int i = 1; i <= nEdgesOnCell[c - 1]; i++) {
                if (verticesOnCell[IX2(i, c, maxEdges)] == v) cellVerticesAtVertex[IX2(k, v, vertexDegree)] = i;
            }
        }
    }
}

/* mpas_seaice_mesh.F:423-488 */
void ora_interior_vertices(int *interiorVertex, int nVerticesSolve, int vertexDegree, int nCells,
                           const int *cellsOnVertex)
{
    for (int v = 1; v <= nVerticesSolve; v++) {
        int n = 0;
        for (int k = 1; k <= vertexDegree; k++) {
            int c = cellsOnVertex[IX2(k, v, vertexDegree)];
            if (c >= 1 && c <= nCells) n++;
        }
        interiorVertex[v - 1] = (n == vertexDegree) ? 1 : 0;
    }
}

/* ------------------------------------------------------------------ Wachspress basis */

/* the Dunavant and Fekete rules (get_integration_factors_dunavant / _fekete, wachspress.F:1401-1941) as a table of
 * numbers */
#include "wachspress_rules.h"

#define MAXE 16

typedef struct {
    int n;
    double A[MAXE + 1], B[MAXE + 1], kappa[MAXE + 1]; /* 1-based */
    int nSub[MAXE + 1];
    int sub[MAXE + 1][MAXE + 1];
} wach_cell;

/* wachspress.F:535-614 (coefficients) and :628-668 (index subsets) */
static void wach_setup(wach_cell *w, int n, const double *x, const double *y /* 1-based */)
{
    w->n = n;
    for (int i = 1; i <= n; i++) {
        int i1 = i - 1, i2 = i;
        if (i1 < 1) i1 = i1 + n;
        w->A[i] = (y[i2] - y[i1]) / (x[i1] * y[i2] - x[i2] * y[i1]);
        w->B[i] = (x[i1] - x[i2]) / (x[i1] * y[i2] - x[i2] * y[i1]);
    }
    w->kappa[1] = 1.0;
    for (int j = 2; j <= n; j++) {
        int i0 = j - 1, i1 = j, i2 = j + 1;
        if (i2 > n) i2 = i2 - n;
        w->kappa[j] = w->kappa[j - 1] * (w->A[i2] * (x[i0] - x[i1]) + w->B[i2] * (y[i0] - y[i1])) /
                      (w->A[i0] * (x[i1] - x[i0]) + w->B[i0] * (y[i1] - y[i0]));
    }
    for (int j = 1; j <= n; j++) {
        int i1 = j, i2 = wrapped_index(j + 1, n);
        w->nSub[j] = 0;
        for (int k = 1; k <= n; k++) {
            if (k != i1 && k != i2) {
                w->nSub[j]++;
                w->sub[j][w->nSub[j]] = k;
            }
        }
    }
}

/* wachspress.F:1049-1069 */
static double wach_edge(const wach_cell *w, int k, double x, double y) { return 1.0 - w->A[k] * x - w->B[k] * y; }

/* wachspress.F:864-925 */
static double wach_numerator(const wach_cell *w, int j, double x, double y)
{
    double num = 1.0;
    for (int k = 1; k <= w->nSub[j]; k++) num = num * wach_edge(w, w->sub[j][k], x, y);
    return num * w->kappa[j];
}

/* wachspress.F:939-1035 */
static void wach_numerator_derivative(const wach_cell *w, int j, double x, double y, double *dU, double *dV)
{
    double sumU = 0.0, sumV = 0.0;
    for (int k = 1; k <= w->nSub[j]; k++) {
        double pU = 1.0, pV = 1.0;
        for (int l = 1; l <= k - 1; l++) {
            double e = wach_edge(w, w->sub[j][l], x, y);
            pU = pU * e;
            pV = pV * e;
        }
        pU = pU * (-w->A[w->sub[j][k]]);
        pV = pV * (-w->B[w->sub[j][k]]);
        for (int l = k + 1; l <= w->nSub[j]; l++) {
            double e = wach_edge(w, w->sub[j][l], x, y);
            pU = pU * e;
            pV = pV * e;
        }
        sumU = sumU + pU;
        sumV = sumV + pV;
    }
    *dU = sumU * w->kappa[j];
    *dV = sumV * w->kappa[j];
}

/* wachspress.F:682-749: basis function i at (x,y) */
static double wach_basis(const wach_cell *w, int i, double x, double y)
{
    double den = 0.0, numi = 0.0;
    for (int j = 1; j <= w->n; j++) {
        double num = wach_numerator(w, j, x, y);
        den = den + num;
        if (j == i) numi = num;
    }
    return numi / den;
}

/* wachspress.F:763-850: gradient of basis function i at (x,y) */
static void wach_basis_derivative(const wach_cell *w, int i, double x, double y, double *dU, double *dV)
{
    double den = 0.0, sU = 0.0, sV = 0.0, numi = 0.0, dUi = 0.0, dVi = 0.0;
    for (int j = 1; j <= w->n; j++) {
        double num = wach_numerator(w, j, x, y);
        den = den + num;
        double du, dv;
        wach_numerator_derivative(w, j, x, y, &du, &dv);
        sU = sU + du;
        sV = sV + dv;
        if (j == i) {
            numi = num;
            dUi = du;
            dVi = dv;
        }
    }
    *dU = dUi / den - (numi / (den * den)) * sU;
    *dV = dVi / den - (numi / (den * den)) * sV;
}

/* seaice_init_velocity_solver_wachspress, wachspress.F:46-161 with calculate_wachspress_derivatives
 * (:1083-1206) and integrate_wachspress / integrate_wachspress_polygon (:179-467). */
/* get_integration_factors (:1223-1285): the rule on the unit triangle.  type 1 dunavant (:1401-1712, orders 1-10 and
 * 12), type 3 fekete (:1726-1941, orders 1-6, 8, 9), type 2 trapezoidal (:1301-1387, any order up to 21 = 253
 * points).  u, v, w need 256 entries.  Returns the number of points, 0 for a rule the reference does not have
 * ("Unimplemented integration order", :1706, :1935). */
int ora_integration_rule(int type, int order, double *u, double *v, double *w, double *normalizationFactor)
{
    if (type == 1 || type == 3) {
        for (int r = 0; r < SIH_RULE_COUNT; r++)
            if (sih_rule_table[r].kind == type && sih_rule_table[r].order == order) {
                const int n = sih_rule_table[r].n, f = sih_rule_table[r].first;
                for (int p = 0; p < n; p++) {
                    u[p] = sih_rule_u[f + p];
                    v[p] = sih_rule_v[f + p];
                    w[p] = sih_rule_w[f + p];
                }
                *normalizationFactor = 2.0; /* :1426, :1751 */
                return n;
            }
        return 0;
    }
    if (type == 2) {
        int n = order;
        if (n < 1 || n > 21) return 0;
        int nPoints = ((n + 1) * (n + 1) + (n + 1)) / 2;
        int ij = 0;
        for (int i = 0; i <= n; i++)
            for (int j = 0; j <= n - i; j++) {
                u[ij] = (double)i / (double)n;
                v[ij] = (double)j / (double)n;
                ij++;
            }
        ij = 0;
        for (int i = 0; i <= n; i++)
            for (int j = 0; j <= n - i; j++) {
                w[ij] = 0.0;
                if (i <= n - j) {
                    if (i == n || j == n || (i == 0 && j == 0))
                        w[ij] = 1.0;
                    else if ((j == 0 && i != 0 && i != n) || (i == 0 && j != 0 && j != n) ||
                             (i == n - j && i != 0 && j != 0))
                        w[ij] = 3.0;
                    else
                        w[ij] = 6.0;
                }
                ij++;
            }
        *normalizationFactor = 6.0 * ((double)n * (double)n);
        return nPoints;
    }
    return 0;
}

void ora_wachspress_init(int nCells, int maxEdges, const int *nEdgesOnCell, const double *xLocal,
                         const double *yLocal, double *basisGradientU, double *basisGradientV,
                         double *basisIntegralsU, double *basisIntegralsV, double *basisIntegralsMetric, int nq,
                         const double *qU, const double *qV, const double *qW, double qNorm)
{
    size_t mm = (size_t)maxEdges * (size_t)maxEdges;
    memset(basisGradientU, 0, sizeof(double) * mm * (size_t)nCells);
    memset(basisGradientV, 0, sizeof(double) * mm * (size_t)nCells);
    memset(basisIntegralsU, 0, sizeof(double) * mm * (size_t)nCells);
    memset(basisIntegralsV, 0, sizeof(double) * mm * (size_t)nCells);
    memset(basisIntegralsMetric, 0, sizeof(double) * mm * (size_t)nCells);
#pragma omp parallel for schedule(static)
    for (int c = 1; c <= nCells; c++) {
        int n = nEdgesOnCell[c - 1];
        double x[MAXE + 1], y[MAXE + 1];
        wach_cell w;
        for (int i = 1; i <= n; i++) {
            x[i] = xLocal[IX2(i, c, maxEdges)];
            y[i] = yLocal[IX2(i, c, maxEdges)];
        }
        wach_setup(&w, n, x, y);
        /* gradients at the cell's own vertices, kept only at the basis vertex and its two neighbours */
        for (int ib = 1; ib <= n; ib++) {
            int gs[3] = {ib, wrapped_index(ib - 1, n), wrapped_index(ib + 1, n)};
            for (int q = 0; q < 3; q++) {
                int ig = gs[q];
                double du, dv;
                wach_basis_derivative(&w, ib, x[ig], y[ig], &du, &dv);
                basisGradientU[IX3(ib, ig, c, maxEdges)] = du;
                basisGradientV[IX3(ib, ig, c, maxEdges)] = dv;
            }
        }
        /* integrals: sub-triangles (centre, vertex t, vertex t+1); for every pair (is, iv) the sum over the
         * points of a sub-triangle first (in point order), then over the sub-triangles (:440-463) */
        double SU[MAXE + 1][MAXE + 1], SV[MAXE + 1][MAXE + 1], SM[MAXE + 1][MAXE + 1];
        for (int iv = 1; iv <= n; iv++)
            for (int is = 1; is <= n; is++) SU[iv][is] = SV[iv][is] = SM[iv][is] = 0.0;
        for (int t = 1; t <= n; t++) {
            int i1 = t, i2 = wrapped_index(t + 1, n);
            /* get_triangle_mapping (:485-517) with (x1,y1)=(1,0), (x2,y2)=(0,1) */
            double x1 = 1.0, y1 = 0.0, x2 = 0.0, y2 = 1.0;
            double u1 = x[i1], v1 = y[i1], u2 = x[i2], v2 = y[i2];
            double m11 = (u2 * y1 - u1 * y2) / (x2 * y1 - x1 * y2);
            double m12 = (u1 * x2 - u2 * x1) / (y1 * x2 - y2 * x1);
            double m21 = (v2 * y1 - v1 * y2) / (x2 * y1 - x1 * y2);
            double m22 = (v1 * x2 - v2 * x1) / (y1 * x2 - y2 * x1);
            double jac = m11 * m22 - m12 * m21;
            double sU[MAXE + 1][MAXE + 1], sV[MAXE + 1][MAXE + 1], sM[MAXE + 1][MAXE + 1];
            for (int iv = 1; iv <= n; iv++)
                for (int is = 1; is <= n; is++) sU[iv][is] = sV[iv][is] = sM[iv][is] = 0.0;
            for (int p = 0; p < nq; p++) {
                double px = m11 * qU[p] + m12 * qV[p];
                double py = m21 * qU[p] + m22 * qV[p];
                double N[MAXE + 1], dNu[MAXE + 1], dNv[MAXE + 1];
                for (int i = 1; i <= n; i++) {
                    N[i] = wach_basis(&w, i, px, py);
                    wach_basis_derivative(&w, i, px, py, &dNu[i], &dNv[i]);
                }
                for (int iv = 1; iv <= n; iv++)
                    for (int is = 1; is <= n; is++) {
                        double tmp = jac * qW[p] * N[is];
                        sU[iv][is] = sU[iv][is] + tmp * dNu[iv];
                        sV[iv][is] = sV[iv][is] + tmp * dNv[iv];
                        sM[iv][is] = sM[iv][is] + tmp * N[iv];
                    }
            }
            for (int iv = 1; iv <= n; iv++)
                for (int is = 1; is <= n; is++) {
                    SU[iv][is] = SU[iv][is] + sU[iv][is] / qNorm;
                    SV[iv][is] = SV[iv][is] + sV[iv][is] / qNorm;
                    SM[iv][is] = SM[iv][is] + sM[iv][is] / qNorm;
                }
        }
        for (int iv = 1; iv <= n; iv++)
            for (int is = 1; is <= n; is++) {
                basisIntegralsU[IX3(is, iv, c, maxEdges)] = SU[iv][is];
                basisIntegralsV[IX3(is, iv, c, maxEdges)] = SV[iv][is];
                basisIntegralsMetric[IX3(is, iv, c, maxEdges)] = SM[iv][is];
            }
    }
}


/* ------------------------------------------------------------------ piecewise-linear basis */

/* seaice_solve_linear_basis_system, mpas_seaice_numerics.F:44-212: LU decomposition with implicitly
 * scaled partial pivoting and back substitution, n = 3 here */
static void solve_linear_basis_system(const double L[3][3], const double rhs[3], double sol[3])
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
        int jmax = j; /* maxloc: first maximum */
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

/* seaice_init_velocity_solver_pwl, mpas_seaice_velocity_solver_pwl.F:44-373 */
void ora_pwl_init(int nCells, int maxEdges, const int *nEdgesOnCell, const int *edgesOnCell, const double *dvEdge,
                  const double *areaCell, const double *xLocal, const double *yLocal, double *basisGradientU,
                  double *basisGradientV, double *basisIntegralsMetric, double *basisIntegralsU,
                  double *basisIntegralsV)
{
    for (int c = 1; c <= nCells; c++) {
        const int n = nEdgesOnCell[c - 1];
        double subGU[MAXE + 1][4], subGV[MAXE + 1][4], subC[MAXE + 1][4];
        double cellGU[MAXE + 1][MAXE + 1], cellGV[MAXE + 1][MAXE + 1], subArea[MAXE + 1];
        const double alpha = 1.0 / (double)n;
        double xc = 0.0, yc = 0.0;
        for (int i = 1; i <= n; i++) {
            xc = xc + alpha * xLocal[IX2(i, c, maxEdges)];
            yc = yc + alpha * yLocal[IX2(i, c, maxEdges)];
        }
        double areaSum = 0.0;
        for (int s = 1; s <= n; s++) {
            int e = edgesOnCell[IX2(s, c, maxEdges)];
            int i1 = s, i2 = wrapped_index(s + 1, n);
            double cc = dvEdge[e - 1];
            double dx1 = xLocal[IX2(i1, c, maxEdges)] - xc, dy1 = yLocal[IX2(i1, c, maxEdges)] - yc;
            double dx2 = xLocal[IX2(i2, c, maxEdges)] - xc, dy2 = yLocal[IX2(i2, c, maxEdges)] - yc;
            double a = sqrt(dx1 * dx1 + dy1 * dy1);
            double b = sqrt(dx2 * dx2 + dy2 * dy2);
            double sp = (a + b + cc) * 0.5;
            subArea[s] = sqrt(sp * (sp - a) * (sp - b) * (sp - cc)); /* Heron */
            areaSum = areaSum + subArea[s];
        }
        const double scale = areaCell[c - 1] / areaSum;
        for (int s = 1; s <= n; s++) subArea[s] = subArea[s] * scale;
        for (int s = 1; s <= n; s++) {
            int i1 = s, i2 = wrapped_index(s + 1, n);
            double L[3][3] = {{xLocal[IX2(i1, c, maxEdges)] - xc, yLocal[IX2(i1, c, maxEdges)] - yc, 1.0},
                              {xLocal[IX2(i2, c, maxEdges)] - xc, yLocal[IX2(i2, c, maxEdges)] - yc, 1.0},
                              {0.0, 0.0, 1.0}};
            double r1[3] = {1.0, 0.0, 0.0}, r2[3] = {0.0, 1.0, 0.0}, sol[3];
            solve_linear_basis_system(L, r1, sol);
            subGU[s][1] = sol[0];
            subGV[s][1] = sol[1];
            subC[s][1] = sol[2];
            solve_linear_basis_system(L, r2, sol);
            subGU[s][2] = sol[0];
            subGV[s][2] = sol[1];
            subC[s][2] = sol[2];
            subGU[s][3] = -subGU[s][1] - subGU[s][2];
            subGV[s][3] = -subGV[s][1] - subGV[s][2];
            subC[s][3] = 1.0 - subC[s][1] - subC[s][2];
        }
        for (int ib = 1; ib <= n; ib++)
            for (int s = 1; s <= n; s++) {
                cellGU[ib][s] = subGU[s][3] * alpha;
                cellGV[ib][s] = subGV[s][3] * alpha;
                if (s == ib) {
                    cellGU[ib][s] = cellGU[ib][s] + subGU[s][1];
                    cellGV[ib][s] = cellGV[ib][s] + subGV[s][1];
                } else if (s == wrapped_index(ib - 1, n)) {
                    cellGU[ib][s] = cellGU[ib][s] + subGU[s][2];
                    cellGV[ib][s] = cellGV[ib][s] + subGV[s][2];
                }
            }
        for (int ib = 1; ib <= n; ib++)
            for (int g = 1; g <= n; g++) {
                int s1 = g, s2 = wrapped_index(g - 1, n);
                basisGradientU[IX3(ib, g, c, maxEdges)] = 0.5 * (cellGU[ib][s1] + cellGU[ib][s2]);
                basisGradientV[IX3(ib, g, c, maxEdges)] = 0.5 * (cellGV[ib][s1] + cellGV[ib][s2]);
            }
        for (int is = 1; is <= n; is++)
            for (int iv = 1; iv <= n; iv++) {
                double su = 0.0, sv = 0.0;
                for (int s = 1; s <= n; s++) {
                    double bi;
                    if (s == is || s == wrapped_index(is - 1, n))
                        bi = ((alpha + 1) * subArea[s]) / 3.0;
                    else
                        bi = (alpha * subArea[s]) / 3.0;
                    su = su + cellGU[iv][s] * bi;
                    sv = sv + cellGV[iv][s] * bi;
                }
                basisIntegralsU[IX3(is, iv, c, maxEdges)] = su;
                basisIntegralsV[IX3(is, iv, c, maxEdges)] = sv;
            }
        for (int is = 1; is <= n; is++)
            for (int iv = 1; iv <= n; iv++) {
                double sm = 0.0;
                for (int s = 1; s <= n; s++) {
                    int ts = (s == is) ? 1 : (s == wrapped_index(is - 1, n) ? 2 : 3);
                    int tv = (s == iv) ? 1 : (s == wrapped_index(iv - 1, n) ? 2 : 3);
                    double m = 0.0;
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
                basisIntegralsMetric[IX3(is, iv, c, maxEdges)] = sm;
            }
    }
}

/* variational.F:358-445 */
void ora_variational_denominator(double *variationalDenominator, int nVertices, int vertexDegree, int maxEdges,
                                 const int *nEdgesOnCell, const double *areaTriangle, const int *cellsOnVertex,
                                 const int *cellVerticesAtVertex, const double *basisIntegralsMetric,
                                 int alternate)
{
    for (int v = 1; v <= nVertices; v++) {
        if (alternate) {
            double d = 0.0;
            for (int k = 1; k <= vertexDegree; k++) {
                int c = cellsOnVertex[IX2(k, v, vertexDegree)];
                int j = cellVerticesAtVertex[IX2(k, v, vertexDegree)];
                if (j < 1) continue;
                for (int s = 1; s <= nEdgesOnCell[c - 1]; s++) d = d + basisIntegralsMetric[IX3(s, j, c, maxEdges)];
            }
            variationalDenominator[v - 1] = d;
        } else {
            variationalDenominator[v - 1] = areaTriangle[v - 1];
        }
    }
}

/* ------------------------------------------------------------------ interpolation */

/* mpas_seaice_mesh.F:2760-2892 (the active "cell area" branch :2835-2851).  The reference reads
 * areaCell/variableCell at the nCells+1 sentinel slot for rim vertices; callers here pass arrays
 * with that extra slot. */
void ora_interpolate_cell_to_vertex(double *variableVertex, const double *variableCell, int nVerticesSolve,
                                    int vertexDegree, const int *cellsOnVertex, const double *areaCell)
{
    for (int v = 1; v <= nVerticesSolve; v++) {
        double acc = 0.0, totalArea = 0.0;
        for (int k = 1; k <= vertexDegree; k++) {
            int c = cellsOnVertex[IX2(k, v, vertexDegree)];
            acc = acc + areaCell[c - 1] * variableCell[c - 1];
            totalArea = totalArea + areaCell[c - 1];
        }
        variableVertex[v - 1] = acc / totalArea;
    }
}

/* mpas_seaice_mesh.F:2906-2976 */
void ora_interpolate_vertex_to_cell(double *variableCell, const double *variableVertex, int nCellsSolve,
                                    int maxEdges, const int *nEdgesOnCell, const int *verticesOnCell,
                                    const double *areaTriangle, const int *interiorVertex)
{
    for (int c = 1; c <= nCellsSolve; c++) {
        double acc = 0.0, totalArea = 0.0;
        for (int i = 1; i <= nEdgesOnCell[c - 1]; i++) {
            int v = verticesOnCell[IX2(i, c, maxEdges)];
            acc = acc + areaTriangle[v - 1] * variableVertex[v - 1] * (double)(float)interiorVertex[v - 1];
            totalArea = totalArea + areaTriangle[v - 1] * (double)(float)interiorVertex[v - 1];
        }
        if (totalArea > 0.0) acc = acc / totalArea;
        variableCell[c - 1] = acc;
    }
}

/* ------------------------------------------------------------------ pre-subcycle */

/* velocity_solver.F:685-752.  Category arrays are (ONE, nCategories, nCells). */
void ora_aggregate_mass_and_area(int nCells, int nCategories, const double *iceAreaCategory,
                                 const double *iceVolumeCategory, const double *snowVolumeCategory,
                                 double *iceAreaCell, double *iceVolumeCell, double *snowVolumeCell,
                                 double *totalMassCell)
{
    for (int c = 0; c < nCells; c++) {
        double a = 0.0, vi = 0.0, vs = 0.0;
        for (int k = 0; k < nCategories; k++) {
            a = a + iceAreaCategory[(size_t)c * nCategories + k];
            vi = vi + iceVolumeCategory[(size_t)c * nCategories + k];
            vs = vs + snowVolumeCategory[(size_t)c * nCategories + k];
        }
        iceAreaCell[c] = a;
        iceVolumeCell[c] = vi;
        snowVolumeCell[c] = vs;
        totalMassCell[c] = vi * seaiceDensityIce + vs * seaiceDensitySnow;
    }
}

/* velocity_solver.F:961-1059; arrays carry the nCells+1 sentinel slot (mask there must be "no ice") */
void ora_stress_calculation_mask(int *solveStress, int nCells, int maxEdges, const int *nEdgesOnCell,
                                 const int *cellsOnCell, const double *iceAreaCellInitial,
                                 const double *totalMassCell, const int *landIceMask)
{
    for (int c = 1; c <= nCells; c++) {
        solveStress[c - 1] = 0;
        if (iceAreaCellInitial[c - 1] > seaiceAreaMinimum && totalMassCell[c - 1] > seaiceMassMinimum &&
            landIceMask[c - 1] == 0) {
            solveStress[c - 1] = 1;
        } else {
            for (int i = 1; i <= nEdgesOnCell[c - 1]; i++) {
                int nb = cellsOnCell[IX2(i, c, maxEdges)];
                if (iceAreaCellInitial[nb - 1] > seaiceAreaMinimum && totalMassCell[nb - 1] > seaiceMassMinimum &&
                    landIceMask[nb - 1] == 0) {
                    solveStress[c - 1] = 1;
                    break;
                }
            }
        }
    }
}

/* velocity_solver.F:1073-1150 */
void ora_velocity_calculation_mask(int *solveVelocity, int nVerticesSolve, int nVertices,
                                   const int *interiorVertex, const int *landIceMaskVertex,
                                   const double *iceAreaVertex, const double *totalMassVertex)
{
    for (int v = 0; v < nVerticesSolve; v++) {
        solveVelocity[v] = 0;
        if (interiorVertex[v] == 1 && landIceMaskVertex[v] == 0 && iceAreaVertex[v] > seaiceAreaMinimum &&
            totalMassVertex[v] > seaiceMassMinimum)
            solveVelocity[v] = 1;
    }
    for (int v = nVerticesSolve; v < nVertices; v++) solveVelocity[v] = 0;
}

/* velocity_solver.F:1249-1277 (after the two ocean-velocity interpolations) */
void ora_new_ice_velocities(int nVerticesSolve, int nVertices, const int *solveVelocity, int *solveVelocityPrevious,
                            double *uVelocity, double *vVelocity, double *uVelocityInitial,
                            double *vVelocityInitial, const double *uOceanVelocityVertex,
                            const double *vOceanVelocityVertex, double *stressDivergenceU,
                            double *stressDivergenceV, double *oceanStressU, double *oceanStressV)
{
    for (int v = 0; v < nVerticesSolve; v++) {
        if (solveVelocity[v] == 1) {
            if (solveVelocityPrevious[v] == 0) {
                uVelocity[v] = uOceanVelocityVertex[v];
                vVelocity[v] = vOceanVelocityVertex[v];
            }
        } else {
            uVelocity[v] = 0.0;
            vVelocity[v] = 0.0;
            stressDivergenceU[v] = 0.0;
            stressDivergenceV[v] = 0.0;
            oceanStressU[v] = 0.0;
            oceanStressV[v] = 0.0;
        }
    }
    for (int v = 0; v < nVertices; v++) {
        solveVelocityPrevious[v] = solveVelocity[v];
        uVelocityInitial[v] = uVelocity[v];
        vVelocityInitial[v] = vVelocity[v];
    }
}

/* velocity_solver.F:1419-1435 (Hibler strength, the branch taken without the column package) */
void ora_ice_strength_hibler(int nCellsSolve, const int *solveStress, const double *iceAreaCell,
                             const double *iceVolumeCell, double *icePressure)
{
    for (int c = 0; c < nCellsSolve; c++) {
        if (solveStress[c] == 1)
            icePressure[c] = seaiceIceStrengthConstantHiblerP * iceVolumeCell[c] *
                             exp(-seaiceIceStrengthConstantHiblerC * (1.0 - iceAreaCell[c]));
        else
            icePressure[c] = 0.0;
    }
}

/* The column package's strength, velocity_solver.F:1437-1458 -> colpkg_ice_strength
 * (src/column/ice_colpkg.F90:3498-3610) with ridge_itd (src/column/ice_mechred.F90:741-1012).
 * kstrength 1 Rothrock75 / else Hibler79, krdg_partic 0 Thorndike75 / 1 exponential, krdg_redist 0 Hibler80 /
 * 1 exponential.  iceAreaCategory / iceVolumeCategory are (1, nCategories, nCells+1). */
static double colpkg_ice_strength(int ncat, double aice, double vice, double aice0, const double *aicen,
                                  const double *vicen, int kstrength, int krdg_partic, int krdg_redist,
                                  double mu_rdg, double Cf)
{
    const double puny = 1.0e-11, c0 = 0.0, c1 = 1.0, c2 = 2.0, p5 = 0.5, p333 = 1.0 / 3.0;
    const double Gstar = 0.15, astar = 0.05, maxraft = 1.0, Hstar = 25.0; /* ice_mechred.F90:54-61 */
    const double Gstari = c1 / Gstar, astari = c1 / astar;
    const double Cp = 0.5 * 9.80616 * (1026.0 - 917.0) * 917.0 / 1026.0; /* ice_constants_colpkg.F90:60 */
    const double Pstar = 2.75e4, Cstar = 20.0;
    double GsumA[64 + 2], apartic[64 + 1], hrmin[64 + 1], hrmax[64 + 1], hrexp[64 + 1], krdg[64 + 1];
    double *Gsum = GsumA + 1; /* Gsum(-1:ncat) */
    if (kstrength != 1) return Pstar * vice * exp(-Cstar * (c1 - aice));
    if (ncat > 64) return NAN; /* the work arrays above */
    /* ridge_itd */
    Gsum[-1] = c0;
    if (aice0 > puny)
        Gsum[0] = aice0;
    else
        Gsum[0] = Gsum[-1];
    apartic[0] = c0;
    for (int n = 1; n <= ncat; n++) {
        Gsum[n] = c1;
        apartic[n] = c0;
        hrmin[n] = c0;
        hrmax[n] = c0;
        hrexp[n] = c0;
        krdg[n] = c1;
        if (aicen[n - 1] > puny)
            Gsum[n] = Gsum[n - 1] + aicen[n - 1];
        else
            Gsum[n] = Gsum[n - 1];
    }
    double work = c1 / Gsum[ncat];
    for (int n = 0; n <= ncat; n++) Gsum[n] = Gsum[n] * work;
    if (krdg_partic == 0) {
        for (int n = 0; n <= ncat; n++) {
            if (Gsum[n] < Gstar)
                apartic[n] = Gstari * (Gsum[n] - Gsum[n - 1]) * (c2 - Gstari * (Gsum[n - 1] + Gsum[n]));
            else if (Gsum[n - 1] < Gstar)
                apartic[n] = Gstari * (Gstar - Gsum[n - 1]) * (c2 - Gstari * (Gstar + Gsum[n - 1]));
        }
    } else if (krdg_partic == 1) {
        double xtmp = c1 / (c1 - exp(-astari));
        Gsum[-1] = exp(-Gsum[-1] * astari) * xtmp;
        for (int n = 0; n <= ncat; n++) {
            Gsum[n] = exp(-Gsum[n] * astari) * xtmp;
            apartic[n] = Gsum[n - 1] - Gsum[n];
        }
    }
    if (krdg_redist == 0) {
        for (int n = 1; n <= ncat; n++)
            if (aicen[n - 1] > puny) {
                double hi = vicen[n - 1] / aicen[n - 1];
                hrmin[n] = fmin(c2 * hi, hi + maxraft);
                hrmax[n] = c2 * sqrt(Hstar * hi);
                hrmax[n] = fmax(hrmax[n], hrmin[n] + puny);
                double hrmean = p5 * (hrmin[n] + hrmax[n]);
                krdg[n] = hrmean / hi;
            }
    } else {
        for (int n = 1; n <= ncat; n++)
            if (aicen[n - 1] > puny) {
                double hi = vicen[n - 1] / aicen[n - 1];
                hi = fmax(hi, puny);
                hrmin[n] = fmin(c2 * hi, hi + maxraft);
                hrexp[n] = mu_rdg * sqrt(hi);
                krdg[n] = (hrmin[n] + hrexp[n]) / hi;
            }
    }
    double aksum = apartic[0];
    for (int n = 1; n <= ncat; n++) aksum = aksum + apartic[n] * (c1 - c1 / krdg[n]);
    /* colpkg_ice_strength, Rothrock branch */
    double strength = 0.0;
    if (krdg_redist == 0) {
        for (int n = 1; n <= ncat; n++)
            if (aicen[n - 1] > puny && apartic[n] > c0) {
                double hi = vicen[n - 1] / aicen[n - 1];
                double h2rdg = p333 * (hrmax[n] * hrmax[n] * hrmax[n] - hrmin[n] * hrmin[n] * hrmin[n]) /
                               (hrmax[n] - hrmin[n]);
                double dh2rdg = -hi * hi + h2rdg / krdg[n];
                strength = strength + apartic[n] * dh2rdg;
            }
    } else if (krdg_redist == 1) {
        for (int n = 1; n <= ncat; n++)
            if (aicen[n - 1] > puny && apartic[n] > c0) {
                double hi = vicen[n - 1] / aicen[n - 1];
                double h2rdg = hrmin[n] * hrmin[n] + c2 * hrmin[n] * hrexp[n] + c2 * hrexp[n] * hrexp[n];
                double dh2rdg = -hi * hi + h2rdg / krdg[n];
                strength = strength + apartic[n] * dh2rdg;
            }
    }
    return Cf * Cp * strength / aksum;
}

void ora_ice_strength_column(int nCellsSolve, int nCategories, const int *solveStress, const double *iceAreaCell,
                             const double *iceVolumeCell, const double *openWaterArea,
                             const double *iceAreaCategory, const double *iceVolumeCategory, int kstrength,
                             int krdg_partic, int krdg_redist, double mu_rdg, double Cf, double *icePressure)
{
    for (int c = 0; c < nCellsSolve; c++) {
        icePressure[c] = 0.0;
        if (solveStress[c] == 1)
            icePressure[c] = colpkg_ice_strength(nCategories, iceAreaCell[c], iceVolumeCell[c], openWaterArea[c],
                                                 iceAreaCategory + (size_t)c * nCategories,
                                                 iceVolumeCategory + (size_t)c * nCategories, kstrength, krdg_partic,
                                                 krdg_redist, mu_rdg, Cf);
    }
}

/* velocity_solver.F:1665-1728 */
void ora_constant_air_stress(int nCellsSolve, const double *uAirVelocity, const double *vAirVelocity,
                             const double *airDensity, const double *iceAreaCell, double *airStressCellU,
                             double *airStressCellV)
{
    const double airStressCoeff = 0.0012;
    for (int c = 0; c < nCellsSolve; c++) {
        double windSpeed = sqrt(uAirVelocity[c] * uAirVelocity[c] + vAirVelocity[c] * vAirVelocity[c]);
        airStressCellU[c] = airDensity[c] * windSpeed * airStressCoeff * uAirVelocity[c] * iceAreaCell[c];
        airStressCellV[c] = airDensity[c] * windSpeed * airStressCoeff * vAirVelocity[c] * iceAreaCell[c];
    }
}

/* velocity_solver.F:1742-1788 */
void ora_coriolis_force_coefficient(int nVerticesSolve, const double *totalMassVertex, const double *fVertex,
                                    double *totalMassVertexfVertex)
{
    for (int v = 0; v < nVerticesSolve; v++) totalMassVertexfVertex[v] = totalMassVertex[v] * fVertex[v];
}

/* velocity_solver.F:1802-1883 */
void ora_ocean_stress(int nVerticesSolve, int nVertices, int useOceanStress, const int *solveVelocity,
                      const double *uOceanVelocityVertex, const double *vOceanVelocityVertex,
                      const double *fVertex, double *oceanStressU, double *oceanStressV)
{
    if (useOceanStress) {
        for (int v = 0; v < nVerticesSolve; v++) {
            if (solveVelocity[v] == 1) {
                oceanStressU[v] = uOceanVelocityVertex[v] * cosOceanTurningAngle -
                                  vOceanVelocityVertex[v] * sinOceanTurningAngle * sign1(fVertex[v]);
                oceanStressV[v] = uOceanVelocityVertex[v] * sinOceanTurningAngle * sign1(fVertex[v]) +
                                  vOceanVelocityVertex[v] * cosOceanTurningAngle;
            } else {
                oceanStressU[v] = 0.0;
                oceanStressV[v] = 0.0;
            }
        }
    } else {
        for (int v = 0; v < nVertices; v++) {
            oceanStressU[v] = 0.0;
            oceanStressV[v] = 0.0;
        }
    }
}

/* velocity_solver.F:1941-2010 (geostrophic), :2149-2164 (ssh gradient, after interpolation), :2183-2213 */
void ora_surface_tilt(int nVerticesSolve, int nVertices, int mode /*0 none,1 geostrophic,2 ssh*/,
                      const int *solveVelocity, const double *totalMassVertex, const double *fVertex,
                      const double *uOceanVelocityVertex, const double *vOceanVelocityVertex,
                      const double *seaSurfaceTiltVertexU, const double *seaSurfaceTiltVertexV,
                      double *surfaceTiltForceU, double *surfaceTiltForceV)
{
    if (mode == 0) {
        for (int v = 0; v < nVertices; v++) {
            surfaceTiltForceU[v] = 0.0;
            surfaceTiltForceV[v] = 0.0;
        }
        return;
    }
    for (int v = 0; v < nVerticesSolve; v++) {
        if (solveVelocity[v] == 1) {
            if (mode == 1) {
                surfaceTiltForceU[v] = -fVertex[v] * totalMassVertex[v] * vOceanVelocityVertex[v];
                surfaceTiltForceV[v] = fVertex[v] * totalMassVertex[v] * uOceanVelocityVertex[v];
            } else {
                surfaceTiltForceU[v] = -seaiceGravity * totalMassVertex[v] * seaSurfaceTiltVertexU[v];
                surfaceTiltForceV[v] = -seaiceGravity * totalMassVertex[v] * seaSurfaceTiltVertexV[v];
            }
        } else {
            surfaceTiltForceU[v] = 0.0;
            surfaceTiltForceV[v] = 0.0;
        }
    }
}

/* velocity_solver.F:2227-2386 (variational branch) */
void ora_init_subcycle_variables(int nCells, int nVerticesSolve, int nVertices, int maxEdges,
                                 const int *solveStress, const int *solveVelocity, double *stressDivergenceU,
                                 double *stressDivergenceV, double *uVelocity, double *vVelocity,
                                 double *oceanStressCoeff, double *strain11, double *strain22, double *strain12,
                                 double *stress11, double *stress22, double *stress12)
{
    for (int v = 0; v < nVertices; v++) {
        stressDivergenceU[v] = 0.0;
        stressDivergenceV[v] = 0.0;
    }
    for (int v = 0; v < nVerticesSolve; v++) {
        if (solveVelocity[v] != 1) {
            uVelocity[v] = 0.0;
            vVelocity[v] = 0.0;
            oceanStressCoeff[v] = 0.0;
        }
    }
    size_t n = (size_t)maxEdges * (size_t)nCells;
    memset(strain11, 0, n * sizeof(double));
    memset(strain22, 0, n * sizeof(double));
    memset(strain12, 0, n * sizeof(double));
    for (int c = 1; c <= nCells; c++) {
        if (solveStress[c - 1] != 1) {
            for (int i = 1; i <= maxEdges; i++) {
                stress11[IX2(i, c, maxEdges)] = 0.0;
                stress22[IX2(i, c, maxEdges)] = 0.0;
                stress12[IX2(i, c, maxEdges)] = 0.0;
            }
        }
    }
}

/* ------------------------------------------------------------------ subcycle kernels */

/* variational.F:575-670 */
void ora_strain_tensor_variational(int nCells, int maxEdges, const int *nEdgesOnCell, const int *verticesOnCell,
                                   double *strain11, double *strain22, double *strain12, const double *uVelocity,
                                   const double *vVelocity, const double *basisGradientU,
                                   const double *basisGradientV, const double *tanLatVertexRotatedOverRadius,
                                   const int *solveStress)
{
#pragma omp parallel for schedule(static)
    for (int c = 1; c <= nCells; c++) {
        if (solveStress[c - 1] == 1) {
            for (int g = 1; g <= nEdgesOnCell[c - 1]; g++) {
                double s11 = 0.0, s22 = 0.0, s12 = 0.0;
                for (int b = 1; b <= nEdgesOnCell[c - 1]; b++) {
                    int v = verticesOnCell[IX2(b, c, maxEdges)];
                    s11 = s11 + uVelocity[v - 1] * basisGradientU[IX3(b, g, c, maxEdges)];
                    s22 = s22 + vVelocity[v - 1] * basisGradientV[IX3(b, g, c, maxEdges)];
                    s12 = s12 + 0.5 * (uVelocity[v - 1] * basisGradientV[IX3(b, g, c, maxEdges)] +
                                       vVelocity[v - 1] * basisGradientU[IX3(b, g, c, maxEdges)]);
                }
                int j = verticesOnCell[IX2(g, c, maxEdges)];
                strain11[IX2(g, c, maxEdges)] = s11 - vVelocity[j - 1] * tanLatVertexRotatedOverRadius[j - 1];
                strain12[IX2(g, c, maxEdges)] =
                    s12 + uVelocity[j - 1] * tanLatVertexRotatedOverRadius[j - 1] * 0.5;
                strain22[IX2(g, c, maxEdges)] = s22;
            }
        }
    }
}

/* seaice_average_strains_on_vertex, variational.F:684-763 (serial in the reference).  Every (corner,
 * cell) entry belongs to exactly one vertex, so the vertex loop can run in any order. */
void ora_average_strains_on_vertex(int nCells, int nVerticesSolve, int vertexDegree, int maxEdges,
                                   const int *cellsOnVertex, const int *cellVerticesAtVertex, const double *areaCell,
                                   double *strain11, double *strain22, double *strain12)
{
    for (int v = 1; v <= nVerticesSolve; v++) {
        double a11 = 0.0, a22 = 0.0, a12 = 0.0, den = 0.0;
        for (int k = 1; k <= vertexDegree; k++) {
            int c = cellsOnVertex[IX2(k, v, vertexDegree)];
            if (c <= nCells) {
                int j = cellVerticesAtVertex[IX2(k, v, vertexDegree)];
                a11 = a11 + strain11[IX2(j, c, maxEdges)] * areaCell[c - 1];
                a22 = a22 + strain22[IX2(j, c, maxEdges)] * areaCell[c - 1];
                a12 = a12 + strain12[IX2(j, c, maxEdges)] * areaCell[c - 1];
                den = den + areaCell[c - 1];
            }
        }
        a11 = a11 / den;
        a22 = a22 / den;
        a12 = a12 / den;
        for (int k = 1; k <= vertexDegree; k++) {
            int c = cellsOnVertex[IX2(k, v, vertexDegree)];
            if (c <= nCells) {
                int j = cellVerticesAtVertex[IX2(k, v, vertexDegree)];
                strain11[IX2(j, c, maxEdges)] = a11;
                strain22[IX2(j, c, maxEdges)] = a22;
                strain12[IX2(j, c, maxEdges)] = a12;
            }
        }
    }
}

/* seaice_set_special_boundaries_velocity, special_boundaries.F:253-331: sequential and in place, so a
 * source that is itself a boundary vertex with a lower index has already been replaced when it is read.
 * Types: 0 none, 1 periodic, 2 reverse, 3 zero (special_boundaries.F:35-39). */
void ora_set_special_boundaries_velocity(int nVertices, const int *vertexBoundaryType,
                                         const int *vertexBoundarySourceLocal, double *uVelocity, double *vVelocity)
{
    for (int v = 1; v <= nVertices; v++) {
        int t = vertexBoundaryType[v - 1];
        if (t == 1) {
            int s = vertexBoundarySourceLocal[v - 1];
            uVelocity[v - 1] = uVelocity[s - 1];
            vVelocity[v - 1] = vVelocity[s - 1];
        } else if (t == 2) {
            int s = vertexBoundarySourceLocal[v - 1];
            uVelocity[v - 1] = -uVelocity[s - 1];
            vVelocity[v - 1] = -vVelocity[s - 1];
        } else if (t == 3) {
            uVelocity[v - 1] = 0.0;
            vVelocity[v - 1] = 0.0;
        }
    }
}

/* constitutive_relation.F:178-248 */
static void evp_constitutive_relation(double *stress11, double *stress22, double *stress12, double strain11,
                                      double strain22, double strain12, double icePressure,
                                      double *replacementPressure, double dtElastic, double dampingTimescale)
{
    double strainDivergence = strain11 + strain22;
    double strainTension = strain11 - strain22;
    double strainShearing = strain12 * 2.0;
    double stress1 = *stress11 + *stress22;
    double stress2 = *stress11 - *stress22;
    double Delta = sqrt(strainDivergence * strainDivergence +
                        (strainTension * strainTension + strainShearing * strainShearing) / eccentricitySquared);
    double pressureCoefficient = icePressure / fmax(Delta, puny);
    *replacementPressure = pressureCoefficient * Delta;
    pressureCoefficient = (pressureCoefficient * dtElastic) / (2.0 * dampingTimescale);
    double denominator = 1.0 + (0.5 * dtElastic) / dampingTimescale;
    stress1 = (stress1 + pressureCoefficient * (strainDivergence - Delta)) / denominator;
    stress2 = (stress2 + (pressureCoefficient / eccentricitySquared) * strainTension) / denominator;
    *stress12 = (*stress12 + (pressureCoefficient / eccentricitySquared) * strainShearing * 0.5) / denominator;
    *stress11 = 0.5 * (stress1 + stress2);
    *stress22 = 0.5 * (stress1 - stress2);
}

/* constitutive_relation.F:262-330 */
static void evp_constitutive_relation_revised(double *stress11, double *stress22, double *stress12,
                                              double strain11, double strain22, double strain12,
                                              double icePressure, double *replacementPressure)
{
    const double dampingRatioDenominator = 0.86, dampingRatio = 5.5e-3;
    double strainDivergence = strain11 + strain22;
    double strainTension = strain11 - strain22;
    double strainShearing = strain12 * 2.0;
    double stress1 = *stress11 + *stress22;
    double stress2 = *stress11 - *stress22;
    double Delta = sqrt(strainDivergence * strainDivergence +
                        (strainTension * strainTension + strainShearing * strainShearing) / eccentricitySquared);
    double pressureCoefficient = icePressure / fmax(Delta, puny);
    *replacementPressure = pressureCoefficient * Delta;
    pressureCoefficient = (pressureCoefficient * 2.0 * dampingRatio) / dampingRatioDenominator;
    double denominator = 1.0 + (2.0 * dampingRatio) / dampingRatioDenominator;
    stress1 = (stress1 + pressureCoefficient * (strainDivergence - Delta)) / denominator;
    stress2 = (stress2 + (pressureCoefficient / eccentricitySquared) * strainTension) / denominator;
    *stress12 = (*stress12 + (pressureCoefficient / eccentricitySquared) * strainShearing * 0.5) / denominator;
    *stress11 = 0.5 * (stress1 + stress2);
    *stress22 = 0.5 * (stress1 - stress2);
}

/* variational.F:777-975; relation: 1 evp, 2 evp_revised, 3 linear (constitutive_relation.F:344-373) */
void ora_stress_tensor_variational(int nCells, int maxEdges, const int *nEdgesOnCell, double *stress11,
                                   double *stress22, double *stress12, const double *strain11,
                                   const double *strain22, const double *strain12, const double *icePressure,
                                   double *replacementPressure, const int *solveStress, double dtElastic,
                                   double dampingTimescale, int relation)
{
#pragma omp parallel for schedule(static)
    for (int c = 1; c <= nCells; c++) {
        if (relation == 1)
            for (int i = 1; i <= maxEdges; i++) replacementPressure[IX2(i, c, maxEdges)] = 0.0;
        if (solveStress[c - 1] == 1) {
            for (int i = 1; i <= nEdgesOnCell[c - 1]; i++) {
                size_t k = IX2(i, c, maxEdges);
                if (relation == 1)
                    evp_constitutive_relation(&stress11[k], &stress22[k], &stress12[k], strain11[k], strain22[k],
                                              strain12[k], icePressure[c - 1], &replacementPressure[k], dtElastic,
                                              dampingTimescale);
                else if (relation == 2)
                    evp_constitutive_relation_revised(&stress11[k], &stress22[k], &stress12[k], strain11[k],
                                                      strain22[k], strain12[k], icePressure[c - 1],
                                                      &replacementPressure[k]);
                else {
                    stress11[k] = 1.0 * strain11[k];
                    stress22[k] = 1.0 * strain22[k];
                    stress12[k] = 1.0 * strain12[k];
                }
            }
        }
    }
}

/* variational.F:1064-1184 */
void ora_stress_divergence_variational(int nVerticesSolve, int vertexDegree, int maxEdges, const int *nEdgesOnCell,
                                       const int *cellsOnVertex, const int *cellVerticesAtVertex,
                                       double *stressDivergenceU, double *stressDivergenceV,
                                       const double *stress11, const double *stress22, const double *stress12,
                                       const double *basisIntegralsU, const double *basisIntegralsV,
                                       const double *basisIntegralsMetric, const double *variationalDenominator,
                                       const double *tanLatVertexRotatedOverRadius, const int *solveVelocity)
{
#pragma omp parallel for schedule(static)
    for (int v = 1; v <= nVerticesSolve; v++) {
        if (solveVelocity[v - 1] == 1) {
            double divU = 0.0, divV = 0.0;
            for (int k = 1; k <= vertexDegree; k++) {
                int c = cellsOnVertex[IX2(k, v, vertexDegree)];
                int j = cellVerticesAtVertex[IX2(k, v, vertexDegree)];
                double cu = 0.0, cv = 0.0;
                for (int s = 1; s <= nEdgesOnCell[c - 1]; s++) {
                    size_t ks = IX2(s, c, maxEdges);
                    size_t kb = IX3(s, j, c, maxEdges);
                    cu = cu - stress11[ks] * basisIntegralsU[kb] - stress12[ks] * basisIntegralsV[kb] -
                         stress12[ks] * basisIntegralsMetric[kb] * tanLatVertexRotatedOverRadius[v - 1];
                    cv = cv - stress22[ks] * basisIntegralsV[kb] - stress12[ks] * basisIntegralsU[kb] +
                         stress11[ks] * basisIntegralsMetric[kb] * tanLatVertexRotatedOverRadius[v - 1];
                }
                divU = divU + cu;
                divV = divV + cv;
            }
            stressDivergenceU[v - 1] = divU / variationalDenominator[v - 1];
            stressDivergenceV[v - 1] = divV / variationalDenominator[v - 1];
        }
    }
}

/* velocity_solver.F:2986-3082 (serial in the reference) */
void ora_ocean_stress_coefficient(int nVerticesSolve, int nVertices, int useOceanStress, int quadratic,
                                  const int *solveVelocity, double *oceanStressCoeff,
                                  const double *uOceanVelocityVertex, const double *vOceanVelocityVertex,
                                  const double *uVelocity, const double *vVelocity, const double *iceAreaVertex)
{
    if (!useOceanStress) {
        for (int v = 0; v < nVertices; v++) oceanStressCoeff[v] = 0.0;
        return;
    }
    for (int v = 0; v < nVerticesSolve; v++) {
        if (solveVelocity[v] == 1) {
            if (quadratic) {
                double du = uOceanVelocityVertex[v] - uVelocity[v];
                double dv = vOceanVelocityVertex[v] - vVelocity[v];
                oceanStressCoeff[v] = seaiceIceOceanDragCoefficient * seaiceDensitySeaWater * iceAreaVertex[v] *
                                      sqrt(du * du + dv * dv);
            } else {
                oceanStressCoeff[v] = seaiceIceOceanDragCoefficient * seaiceDensitySeaWater * iceAreaVertex[v];
            }
        }
    }
}

/* velocity_solver.F:3096-3208 */
void ora_solve_velocity(int nVerticesSolve, const int *solveVelocity, double *uVelocity, double *vVelocity,
                        const double *totalMassVertex, const double *totalMassVertexfVertex,
                        const double *stressDivergenceU, const double *stressDivergenceV,
                        const double *airStressVertexU, const double *airStressVertexV,
                        const double *surfaceTiltForceU, const double *surfaceTiltForceV,
                        const double *oceanStressU, const double *oceanStressV, const double *oceanStressCoeff,
                        double elasticTimeStep)
{
#pragma omp parallel for schedule(static)
    for (int v = 0; v < nVerticesSolve; v++) {
        if (solveVelocity[v] == 1) {
            double l11 = totalMassVertex[v] / elasticTimeStep + oceanStressCoeff[v] * cosOceanTurningAngle;
            double l12 = -totalMassVertexfVertex[v] -
                         oceanStressCoeff[v] * sinOceanTurningAngle * sign1(totalMassVertexfVertex[v]);
            double l21 = totalMassVertexfVertex[v] +
                         oceanStressCoeff[v] * sinOceanTurningAngle * sign1(totalMassVertexfVertex[v]);
            double l22 = totalMassVertex[v] / elasticTimeStep + oceanStressCoeff[v] * cosOceanTurningAngle;
            double r1 = stressDivergenceU[v] + airStressVertexU[v] + surfaceTiltForceU[v] +
                        oceanStressCoeff[v] * oceanStressU[v] + (totalMassVertex[v] * uVelocity[v]) / elasticTimeStep;
            double r2 = stressDivergenceV[v] + airStressVertexV[v] + surfaceTiltForceV[v] +
                        oceanStressCoeff[v] * oceanStressV[v] + (totalMassVertex[v] * vVelocity[v]) / elasticTimeStep;
            double den = l11 * l22 - l12 * l21;
            uVelocity[v] = (l22 * r1 - l12 * r2) / den;
            vVelocity[v] = (l11 * r2 - l21 * r1) / den;
        }
    }
}

/* velocity_solver.F:3222-3342 */
void ora_solve_velocity_revised(int nVerticesSolve, const int *solveVelocity, double *uVelocity, double *vVelocity,
                                const double *uVelocityInitial, const double *vVelocityInitial,
                                const double *totalMassVertex, const double *totalMassVertexfVertex,
                                const double *stressDivergenceU, const double *stressDivergenceV,
                                const double *airStressVertexU, const double *airStressVertexV,
                                const double *surfaceTiltForceU, const double *surfaceTiltForceV,
                                const double *oceanStressU, const double *oceanStressV,
                                const double *oceanStressCoeff, double dynamicsTimeStep,
                                double numericalInertiaCoefficient)
{
    for (int v = 0; v < nVerticesSolve; v++) {
        if (solveVelocity[v] == 1) {
            double l11 = (numericalInertiaCoefficient + 1.0) * (totalMassVertex[v] / dynamicsTimeStep) +
                         oceanStressCoeff[v] * cosOceanTurningAngle;
            double l12 = -totalMassVertexfVertex[v] -
                         oceanStressCoeff[v] * sinOceanTurningAngle * sign1(totalMassVertexfVertex[v]);
            double l21 = totalMassVertexfVertex[v] +
                         oceanStressCoeff[v] * sinOceanTurningAngle * sign1(totalMassVertexfVertex[v]);
            double l22 = (numericalInertiaCoefficient + 1.0) * (totalMassVertex[v] / dynamicsTimeStep) +
                         oceanStressCoeff[v] * cosOceanTurningAngle;
            double r1 = stressDivergenceU[v] + airStressVertexU[v] + surfaceTiltForceU[v] +
                        oceanStressCoeff[v] * oceanStressU[v] +
                        (totalMassVertex[v] * (numericalInertiaCoefficient * uVelocity[v] + uVelocityInitial[v])) /
                            dynamicsTimeStep;
            double r2 = stressDivergenceV[v] + airStressVertexV[v] + surfaceTiltForceV[v] +
                        oceanStressCoeff[v] * oceanStressV[v] +
                        (totalMassVertex[v] * (numericalInertiaCoefficient * vVelocity[v] + vVelocityInitial[v])) /
                            dynamicsTimeStep;
            double den = l11 * l22 - l12 * l21;
            uVelocity[v] = (l22 * r1 - l12 * r2) / den;
            vVelocity[v] = (l11 * r2 - l21 * r1) / den;
        }
    }
}

/* ------------------------------------------------------------------ post-subcycle */

/* variational.F:1198-1330 */
void ora_final_divergence_shear(int nCells, int maxEdges, const int *nEdgesOnCell, const int *solveStress,
                                const double *strain11, const double *strain22, const double *strain12,
                                double *divergence, double *shear, int useRidging, double *ridgeConvergence,
                                double *ridgeShear)
{
    double *DeltaAverage = (double *)calloc((size_t)nCells, sizeof(double));
    for (int c = 1; c <= nCells; c++) {
        if (solveStress[c - 1] == 1) {
            double dS = 0.0, tS = 0.0, sS = 0.0, dA = 0.0;
            for (int i = 1; i <= nEdgesOnCell[c - 1]; i++) {
                size_t k = IX2(i, c, maxEdges);
                double sd = strain11[k] + strain22[k];
                double st = strain11[k] - strain22[k];
                double ss = strain12[k] * 2.0;
                double Delta = sqrt(sd * sd + (st * st + ss * ss) / eccentricitySquared);
                dS = dS + sd;
                tS = tS + st;
                sS = sS + ss;
                dA = dA + Delta;
            }
            double en = (double)nEdgesOnCell[c - 1];
            divergence[c - 1] = dS / en;
            shear[c - 1] = sqrt(tS * tS + sS * sS) / en;
            DeltaAverage[c - 1] = dA / en;
        } else {
            divergence[c - 1] = 0.0;
            shear[c - 1] = 0.0;
        }
    }
    if (useRidging) {
        for (int c = 0; c < nCells; c++) {
            if (solveStress[c] == 1) {
                ridgeConvergence[c] = -fmin(divergence[c], 0.0);
                ridgeShear[c] = 0.5 * (DeltaAverage[c] - fabs(divergence[c]));
            } else {
                ridgeConvergence[c] = 0.0;
                ridgeShear[c] = 0.0;
            }
        }
    }
    for (int c = 0; c < nCells; c++) {
        divergence[c] = divergence[c] * 100.0 * 86400.0;
        shear[c] = shear[c] * 100.0 * 86400.0;
    }
    free(DeltaAverage);
}

/* velocity_solver.F:3534-3546 with principal_stresses :3567-3610 */
void ora_principal_stresses(int nCellsSolve, int maxEdges, const int *nEdgesOnCell, const double *stress11,
                            const double *stress22, const double *stress12, const double *replacementPressure,
                            double *principalStress1, double *principalStress2)
{
    for (int c = 1; c <= nCellsSolve; c++) {
        for (int i = 1; i <= nEdgesOnCell[c - 1]; i++) {
            size_t k = IX2(i, c, maxEdges);
            if (replacementPressure[k] > puny) {
                double sq = (stress11[k] + stress22[k]) * (stress11[k] + stress22[k]) -
                            4.0 * stress11[k] * stress22[k] + 4.0 * (stress12[k] * stress12[k]);
                double p1 = 0.5 * (stress11[k] + stress22[k]) + 0.5 * sqrt(sq);
                double p2 = 0.5 * (stress11[k] + stress22[k]) - 0.5 * sqrt(sq);
                principalStress1[k] = p1 / replacementPressure[k];
                principalStress2[k] = p2 / replacementPressure[k];
            } else {
                principalStress1[k] = 1.0e30;
                principalStress2[k] = 1.0e30;
            }
        }
    }
}

/* velocity_solver.F:3709-3732: vertex part of ocean_stress_final (after the coefficient update) */
void ora_ocean_stress_final_vertex(int nVerticesSolve, int nVertices, int useOceanStress, const int *solveVelocity,
                                   const double *oceanStressCoeff, const double *uOceanVelocityVertex,
                                   const double *vOceanVelocityVertex, const double *uVelocity,
                                   const double *vVelocity, const double *fVertex, const double *iceAreaVertex,
                                   double *oceanStressU, double *oceanStressV)
{
    if (!useOceanStress) {
        for (int v = 0; v < nVertices; v++) {
            oceanStressU[v] = 0.0;
            oceanStressV[v] = 0.0;
        }
        return;
    }
    for (int v = 0; v < nVerticesSolve; v++) {
        if (solveVelocity[v] == 1) {
            oceanStressU[v] = oceanStressCoeff[v] *
                              ((uOceanVelocityVertex[v] - uVelocity[v]) * cosOceanTurningAngle -
                               (vOceanVelocityVertex[v] - vVelocity[v]) * sinOceanTurningAngle * sign1(fVertex[v]));
            oceanStressV[v] = oceanStressCoeff[v] *
                              ((vOceanVelocityVertex[v] - vVelocity[v]) * cosOceanTurningAngle +
                               (uOceanVelocityVertex[v] - uVelocity[v]) * sinOceanTurningAngle * sign1(fVertex[v]));
            oceanStressU[v] = oceanStressU[v] / iceAreaVertex[v];
            oceanStressV[v] = oceanStressV[v] / iceAreaVertex[v];
        } else {
            oceanStressU[v] = 0.0;
            oceanStressV[v] = 0.0;
        }
    }
}

/* velocity_solver.F:3816-3825 */
void ora_ocean_stress_final_rescale(int nVerticesSolve, const int *solveVelocity, const double *iceAreaVertex,
                                    double *oceanStressU, double *oceanStressV)
{
    for (int v = 0; v < nVerticesSolve; v++) {
        if (solveVelocity[v] == 1) {
            oceanStressU[v] = oceanStressU[v] * iceAreaVertex[v];
            oceanStressV[v] = oceanStressV[v] * iceAreaVertex[v];
        }
    }
}

/* ------------------------------------------------------------------ subcycle loop (single block) */

/* subcycle_velocity_solver / single_subcycle_velocity_solver, velocity_solver.F:2404-2592, for one
 * block with no halo exchange partner (a full-sphere block owns everything) and no special boundaries. */
void ora_evp_subcycle_loop(int nSubcycles, int nCells, int nVerticesSolve, int nVertices, int maxEdges,
                           int vertexDegree, const int *nEdgesOnCell, const int *verticesOnCell,
                           const int *cellsOnVertex, const int *cellVerticesAtVertex, const int *solveStress,
                           const int *solveVelocity, const double *basisGradientU, const double *basisGradientV,
                           const double *basisIntegralsU, const double *basisIntegralsV,
                           const double *basisIntegralsMetric, const double *variationalDenominator,
                           const double *tanLatVertexRotatedOverRadius, const double *icePressure,
                           const double *totalMassVertex, const double *totalMassVertexfVertex,
                           const double *iceAreaVertex, const double *airStressVertexU,
                           const double *airStressVertexV, const double *surfaceTiltForceU,
                           const double *surfaceTiltForceV, const double *oceanStressU,
                           const double *oceanStressV, const double *uOceanVelocityVertex,
                           const double *vOceanVelocityVertex, const double *uVelocityInitial,
                           const double *vVelocityInitial, double *uVelocity, double *vVelocity,
                           double *strain11, double *strain22, double *strain12, double *stress11,
                           double *stress22, double *stress12, double *replacementPressure,
                           double *stressDivergenceU, double *stressDivergenceV, double *oceanStressCoeff,
                           double elasticTimeStep, double dynamicsTimeStep, double dampingTimescale,
                           double numericalInertiaCoefficient, int relation, int useOceanStress,
                           int quadraticOceanStress, int averageVariationalStrains, const double *areaCell)
{
    for (int it = 0; it < nSubcycles; it++) {
        ora_strain_tensor_variational(nCells, maxEdges, nEdgesOnCell, verticesOnCell, strain11, strain22, strain12,
                                      uVelocity, vVelocity, basisGradientU, basisGradientV,
                                      tanLatVertexRotatedOverRadius, solveStress);
        if (averageVariationalStrains) /* velocity_solver.F:2756-2766 */
            ora_average_strains_on_vertex(nCells, nVerticesSolve, vertexDegree, maxEdges, cellsOnVertex,
                                          cellVerticesAtVertex, areaCell, strain11, strain22, strain12);
        ora_stress_tensor_variational(nCells, maxEdges, nEdgesOnCell, stress11, stress22, stress12, strain11,
                                      strain22, strain12, icePressure, replacementPressure, solveStress,
                                      elasticTimeStep, dampingTimescale, relation);
        ora_stress_divergence_variational(nVerticesSolve, vertexDegree, maxEdges, nEdgesOnCell, cellsOnVertex,
                                          cellVerticesAtVertex, stressDivergenceU, stressDivergenceV, stress11,
                                          stress22, stress12, basisIntegralsU, basisIntegralsV,
                                          basisIntegralsMetric, variationalDenominator,
                                          tanLatVertexRotatedOverRadius, solveVelocity);
        ora_ocean_stress_coefficient(nVerticesSolve, nVertices, useOceanStress, quadraticOceanStress, solveVelocity,
                                     oceanStressCoeff, uOceanVelocityVertex, vOceanVelocityVertex, uVelocity,
                                     vVelocity, iceAreaVertex);
        if (relation == 1)
            ora_solve_velocity(nVerticesSolve, solveVelocity, uVelocity, vVelocity, totalMassVertex,
                               totalMassVertexfVertex, stressDivergenceU, stressDivergenceV, airStressVertexU,
                               airStressVertexV, surfaceTiltForceU, surfaceTiltForceV, oceanStressU, oceanStressV,
                               oceanStressCoeff, elasticTimeStep);
        else if (relation == 2)
            ora_solve_velocity_revised(nVerticesSolve, solveVelocity, uVelocity, vVelocity, uVelocityInitial,
                                       vVelocityInitial, totalMassVertex, totalMassVertexfVertex,
                                       stressDivergenceU, stressDivergenceV, airStressVertexU, airStressVertexV,
                                       surfaceTiltForceU, surfaceTiltForceV, oceanStressU, oceanStressV,
                                       oceanStressCoeff, dynamicsTimeStep, numericalInertiaCoefficient);
    }
}

/* ================================================================== weak (cell-centred) scheme
 * mpas_seaice_velocity_solver_weak.F and the normal vectors of mpas_seaice_mesh.F:703-1606.
 * Layouts: normalVectorPolygon(2, maxEdges, nCells), normalVectorTriangle(2, vertexDegree, nVertices). */
#define NV3(i, k, e, K) (((size_t)((e) - 1) * (size_t)(K) + (size_t)((k) - 1)) * 2 + (size_t)((i) - 1))

/* the rotation onto the rotated equator of normal_vectors_spherical_*_metric (mesh.F:1148-1169, 1199-1201):
 * out = Ry(lat) * Rz(-lon) * in */
static void to_equator(double out[3], const double in[3], double lat, double lon, int removeMetricTerms)
{
    double y[3][3] = {{0, 0, 0}, {0, 1, 0}, {0, 0, 0}}, z[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 1}};
    if (removeMetricTerms) {
        y[0][0] = cos(lat);
        y[0][2] = sin(lat);
        y[2][0] = -sin(lat);
        y[2][2] = cos(lat);
        z[0][0] = cos(-lon);
        z[0][1] = -sin(-lon);
        z[1][0] = sin(-lon);
        z[1][1] = cos(-lon);
    } else {
        y[0][0] = y[2][2] = 1.0;
        z[0][0] = z[1][1] = 1.0;
    }
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = z[i][0] * in[0] + z[i][1] * in[1] + z[i][2] * in[2]; /* matmul order */
    for (int i = 0; i < 3; i++) out[i] = y[i][0] * t[0] + y[i][1] * t[1] + y[i][2] * t[2];
}

/* normal of the great circle through a and b (a, b, mid on the rotated equator frame) projected on the local
 * east / north directions at `mid`, mesh.F:1203-1233 */
static void great_circle_normal(double n2[2], const double a[3], const double b[3], const double mid[3], int flip)
{
    double v[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    double g[3] = {v[1] * mid[2] - v[2] * mid[1], v[2] * mid[0] - v[0] * mid[2], v[0] * mid[1] - v[1] * mid[0]};
    if (flip) {
        g[0] = -1.0 * g[0];
        g[1] = -1.0 * g[1];
        g[2] = -1.0 * g[2];
    }
    double norm = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    g[0] = g[0] / norm;
    g[1] = g[1] / norm;
    g[2] = g[2] / norm;
    double e[3] = {-mid[1], mid[0], 0.0};
    double en = sqrt(e[0] * e[0] + e[1] * e[1]);
    e[0] = e[0] / en;
    e[1] = e[1] / en;
    e[2] = e[2] / en;
    n2[0] = g[0] * e[0] + g[1] * e[1] + g[2] * e[2];
    double cl = fmax(fmin(n2[0], 1.0), -1.0);
    n2[1] = sign1(g[2]) * sqrt(1.0 - cl * cl);
}

/* seaice_normal_vectors_polygon, mesh.F:760-793 -> normal_vectors_spherical_polygon_metric :1038-1241 or
 * normal_vectors_planar_polygon :858-943 */
void ora_normal_vectors_polygon(int nCells, int maxEdges, const int *nEdgesOnCell, const int *edgesOnCell,
                                const int *verticesOnEdge, const int *cellsOnEdge, const double *xCell,
                                const double *yCell, const double *zCell, const double *xEdge, const double *yEdge,
                                const double *zEdge, const double *xVertex, const double *yVertex,
                                const double *zVertex, double sphere_radius, int onASphere, int rotate,
                                int removeMetricTerms, double *normalVectorPolygon, double *latCellRotated)
{
    for (int c = 1; c <= nCells; c++) {
        if (!onASphere) {
            for (int ie = 1; ie <= nEdgesOnCell[c - 1]; ie++) {
                int e = edgesOnCell[IX2(ie, c, maxEdges)];
                int v1 = verticesOnEdge[IX2(1, e, 2)], v2 = verticesOnEdge[IX2(2, e, 2)];
                double tx = xVertex[v2 - 1] - xVertex[v1 - 1], ty = yVertex[v2 - 1] - yVertex[v1 - 1];
                double tmag = sqrt(tx * tx + ty * ty);
                tx = tx / tmag;
                ty = ty / tmag;
                double nx = xEdge[e - 1] - xCell[c - 1], ny = yEdge[e - 1] - yCell[c - 1];
                if ((nx * ty - ny * tx) < 0.0) {
                    tx = -tx;
                    ty = -ty;
                }
                normalVectorPolygon[NV3(1, ie, c, maxEdges)] = ty;
                normalVectorPolygon[NV3(2, ie, c, maxEdges)] = -tx;
            }
            latCellRotated[c - 1] = 0.0;
            continue;
        }
        double cr[3];
        grid_rotation_forward(&cr[0], &cr[1], &cr[2], xCell[c - 1], yCell[c - 1], zCell[c - 1], rotate);
        double lon = atan2(cr[1], cr[0]), lat = asin(cr[2] / sphere_radius);
        for (int ie = 1; ie <= nEdgesOnCell[c - 1]; ie++) {
            int e = edgesOnCell[IX2(ie, c, maxEdges)];
            int v1 = verticesOnEdge[IX2(1, e, 2)], v2 = verticesOnEdge[IX2(2, e, 2)];
            double er[3], a[3], b[3], eq[3], aq[3], bq[3];
            grid_rotation_forward(&er[0], &er[1], &er[2], xEdge[e - 1], yEdge[e - 1], zEdge[e - 1], rotate);
            grid_rotation_forward(&a[0], &a[1], &a[2], xVertex[v1 - 1], yVertex[v1 - 1], zVertex[v1 - 1], rotate);
            grid_rotation_forward(&b[0], &b[1], &b[2], xVertex[v2 - 1], yVertex[v2 - 1], zVertex[v2 - 1], rotate);
            to_equator(eq, er, lat, lon, removeMetricTerms);
            to_equator(aq, a, lat, lon, removeMetricTerms);
            to_equator(bq, b, lat, lon, removeMetricTerms);
            great_circle_normal(&normalVectorPolygon[NV3(1, ie, c, maxEdges)], aq, bq, eq,
                                c == cellsOnEdge[IX2(2, e, 2)]);
        }
        latCellRotated[c - 1] = lat;
    }
}

/* seaice_normal_vectors_triangle, mesh.F:807-844 -> normal_vectors_spherical_triangle_metric :1393-1606 (owned
 * vertices) or normal_vectors_planar_triangle :957-1024 (all vertices) */
void ora_normal_vectors_triangle(int nVertices, int nVerticesSolve, int vertexDegree, const int *edgesOnVertex,
                                 const int *cellsOnEdge, const int *verticesOnEdge, const int *interiorVertex,
                                 const double *xCell, const double *yCell, const double *zCell, const double *xEdge,
                                 const double *yEdge, const double *zEdge, const double *xVertex,
                                 const double *yVertex, const double *zVertex, double sphere_radius, int onASphere,
                                 int rotate, int removeMetricTerms, double *normalVectorTriangle,
                                 double *latVertexRotated)
{
    int nLoop = onASphere ? nVerticesSolve : nVertices;
    if (!onASphere)
        for (int v = 0; v < nVertices; v++) latVertexRotated[v] = 0.0;
    for (int v = 1; v <= nLoop; v++) {
        for (int k = 1; k <= vertexDegree; k++) {
            normalVectorTriangle[NV3(1, k, v, vertexDegree)] = 0.0;
            normalVectorTriangle[NV3(2, k, v, vertexDegree)] = 0.0;
        }
        if (onASphere) latVertexRotated[v - 1] = 0.0;
        if (interiorVertex[v - 1] != 1) continue;
        if (!onASphere) {
            for (int k = 1; k <= vertexDegree; k++) {
                int e = edgesOnVertex[IX2(k, v, vertexDegree)];
                double dx = xEdge[e - 1] - xVertex[v - 1], dy = yEdge[e - 1] - yVertex[v - 1];
                normalVectorTriangle[NV3(1, k, v, vertexDegree)] = dx / sqrt(dx * dx + dy * dy);
                normalVectorTriangle[NV3(2, k, v, vertexDegree)] = dy / sqrt(dx * dx + dy * dy);
            }
            continue;
        }
        double vr[3];
        grid_rotation_forward(&vr[0], &vr[1], &vr[2], xVertex[v - 1], yVertex[v - 1], zVertex[v - 1], rotate);
        double lon = atan2(vr[1], vr[0]), lat = asin(vr[2] / sphere_radius);
        for (int k = 1; k <= vertexDegree; k++) {
            int e = edgesOnVertex[IX2(k, v, vertexDegree)];
            int c1 = cellsOnEdge[IX2(1, e, 2)], c2 = cellsOnEdge[IX2(2, e, 2)];
            double er[3], a[3], b[3], eq[3], aq[3], bq[3];
            grid_rotation_forward(&er[0], &er[1], &er[2], xEdge[e - 1], yEdge[e - 1], zEdge[e - 1], rotate);
            grid_rotation_forward(&a[0], &a[1], &a[2], xCell[c1 - 1], yCell[c1 - 1], zCell[c1 - 1], rotate);
            grid_rotation_forward(&b[0], &b[1], &b[2], xCell[c2 - 1], yCell[c2 - 1], zCell[c2 - 1], rotate);
            to_equator(eq, er, lat, lon, removeMetricTerms);
            to_equator(aq, a, lat, lon, removeMetricTerms);
            to_equator(bq, b, lat, lon, removeMetricTerms);
            great_circle_normal(&normalVectorTriangle[NV3(1, k, v, vertexDegree)], aq, bq, eq,
                                v == verticesOnEdge[IX2(1, e, 2)]);
        }
        latVertexRotated[v - 1] = lat;
    }
}

/* seaice_strain_tensor_weak, velocity_solver_weak.F:112-253 */
void ora_strain_tensor_weak(int nCells, int maxEdges, const int *nEdgesOnCell, const int *verticesOnCell,
                            const int *edgesOnCell, const int *verticesOnEdge, const double *dvEdge,
                            const double *areaCell, double sphere_radius, double *strain11, double *strain22,
                            double *strain12, const double *uVelocity, const double *vVelocity,
                            const double *normalVectorPolygon, const double *latCellRotated, const int *solveStress)
{
    double sphereRadius = sphere_radius;
    if (sphereRadius == 0.0) sphereRadius = 1.0;
    for (int c = 1; c <= nCells; c++) {
        double s11 = 0.0, s22 = 0.0, s12 = 0.0;
        if (solveStress[c - 1] == 1) {
            double uc = 0.0, vc = 0.0;
            for (int ie = 1; ie <= nEdgesOnCell[c - 1]; ie++) {
                int iv = verticesOnCell[IX2(ie, c, maxEdges)];
                uc = uc + uVelocity[iv - 1];
                vc = vc + vVelocity[iv - 1];
                int e = edgesOnCell[IX2(ie, c, maxEdges)];
                double ue = 0.0, ve = 0.0;
                for (int k = 1; k <= 2; k++) {
                    int w = verticesOnEdge[IX2(k, e, 2)];
                    ue = ue + uVelocity[w - 1];
                    ve = ve + vVelocity[w - 1];
                }
                ue = ue / 2.0;
                ve = ve / 2.0;
                double n1 = normalVectorPolygon[NV3(1, ie, c, maxEdges)], n2 = normalVectorPolygon[NV3(2, ie, c, maxEdges)];
                s11 = s11 + ue * n1 * dvEdge[e - 1];
                s22 = s22 + ve * n2 * dvEdge[e - 1];
                s12 = s12 + 0.5 * (ue * n2 + ve * n1) * dvEdge[e - 1];
            }
            uc = uc / (double)nEdgesOnCell[c - 1];
            vc = vc / (double)nEdgesOnCell[c - 1];
            s11 = s11 / areaCell[c - 1];
            s22 = s22 / areaCell[c - 1];
            s12 = s12 / areaCell[c - 1];
            s11 = s11 - (vc * tan(latCellRotated[c - 1])) / sphereRadius;
            s12 = s12 + (uc * tan(latCellRotated[c - 1]) * 0.5) / sphereRadius;
        }
        strain11[c - 1] = s11;
        strain22[c - 1] = s22;
        strain12[c - 1] = s12;
    }
}

/* seaice_stress_tensor_weak, velocity_solver_weak.F:267-405 */
void ora_stress_tensor_weak(int nCells, double *stress11, double *stress22, double *stress12, const double *strain11,
                            const double *strain22, const double *strain12, const double *icePressure,
                            double *replacementPressure, const int *solveStress, double dtElastic,
                            double dampingTimescale, int relation)
{
    for (int c = 0; c < nCells; c++) {
        if (solveStress[c] == 1) {
            if (relation == 1)
                evp_constitutive_relation(&stress11[c], &stress22[c], &stress12[c], strain11[c], strain22[c],
                                          strain12[c], icePressure[c], &replacementPressure[c], dtElastic,
                                          dampingTimescale);
            else if (relation == 2)
                evp_constitutive_relation_revised(&stress11[c], &stress22[c], &stress12[c], strain11[c], strain22[c],
                                                  strain12[c], icePressure[c], &replacementPressure[c]);
            else { /* seaice_linear_constitutive_relation, constitutive_relation.F:344-373 */
                stress11[c] = 1.0 * strain11[c];
                stress22[c] = 1.0 * strain22[c];
                stress12[c] = 1.0 * strain12[c];
            }
        } else {
            stress11[c] = 0.0;
            stress22[c] = 0.0;
            stress12[c] = 0.0;
        }
    }
}

/* seaice_stress_divergence_weak, velocity_solver_weak.F:493-640.  Loops over ALL vertices of the block (:558);
 * the stress arrays carry the extra slot that cell index nCells+1 points to. */
void ora_stress_divergence_weak(int nVertices, int vertexDegree, const int *cellsOnVertex, const int *edgesOnVertex,
                                const int *cellsOnEdge, const double *areaTriangle, const double *dcEdge,
                                double sphere_radius, double *stressDivergenceU, double *stressDivergenceV,
                                const double *stress11, const double *stress22, const double *stress12,
                                const double *normalVectorTriangle, const double *latVertexRotated,
                                const int *solveVelocity)
{
    double sphereRadius = sphere_radius;
    if (sphereRadius == 0.0) sphereRadius = 1.0;
    for (int v = 1; v <= nVertices; v++) {
        double dU = 0.0, dV = 0.0;
        if (solveVelocity[v - 1] == 1) {
            double s11v = 0.0, s22v = 0.0, s12v = 0.0;
            for (int k = 1; k <= vertexDegree; k++) {
                int c = cellsOnVertex[IX2(k, v, vertexDegree)];
                s11v = s11v + stress11[c - 1];
                s22v = s22v + stress22[c - 1];
                s12v = s12v + stress12[c - 1];
                int e = edgesOnVertex[IX2(k, v, vertexDegree)];
                double s11e = 0.0, s22e = 0.0, s12e = 0.0;
                for (int j = 1; j <= 2; j++) {
                    int cc = cellsOnEdge[IX2(j, e, 2)];
                    s11e = s11e + stress11[cc - 1];
                    s22e = s22e + stress22[cc - 1];
                    s12e = s12e + stress12[cc - 1];
                }
                s11e = s11e / 2.0;
                s22e = s22e / 2.0;
                s12e = s12e / 2.0;
                double n1 = normalVectorTriangle[NV3(1, k, v, vertexDegree)];
                double n2 = normalVectorTriangle[NV3(2, k, v, vertexDegree)];
                dU = dU + (s11e * n1 + s12e * n2) * dcEdge[e - 1];
                dV = dV + (s22e * n2 + s12e * n1) * dcEdge[e - 1];
            }
            s11v = s11v / (double)vertexDegree;
            s22v = s22v / (double)vertexDegree;
            s12v = s12v / (double)vertexDegree;
            dU = dU / areaTriangle[v - 1];
            dV = dV / areaTriangle[v - 1];
            dU = dU - (tan(latVertexRotated[v - 1]) * s12v * 2.0) / sphereRadius;
            dV = dV + (tan(latVertexRotated[v - 1]) * (s11v - s22v)) / sphereRadius;
        }
        stressDivergenceU[v - 1] = dU;
        stressDivergenceV[v - 1] = dV;
    }
}

/* seaice_final_divergence_shear_weak, velocity_solver_weak.F:654-749.  `Delta = sqrt(...)` at :720 assigns the
 * whole array in every iteration, so after the loop every entry holds the last owned cell's value; the ridging
 * diagnostics at :739-740 are computed from that, and so they are here. */
void ora_final_divergence_shear_weak(int nCellsSolve, const double *strain11, const double *strain22,
                                     const double *strain12, double *divergence, double *shear, int useColumn,
                                     double *ridgeConvergence, double *ridgeShear)
{
    double DeltaAll = 0.0;
    for (int c = 0; c < nCellsSolve; c++) {
        double sd = strain11[c] + strain22[c], st = strain11[c] - strain22[c], ss = strain12[c] * 2.0;
        DeltaAll = sqrt(sd * sd + (st * st + ss * ss) / eccentricitySquared);
        divergence[c] = sd;
        shear[c] = sqrt(st * st + ss * ss);
    }
    if (useColumn)
        for (int c = 0; c < nCellsSolve; c++) {
            ridgeConvergence[c] = -fmin(divergence[c], 0.0);
            ridgeShear[c] = 0.5 * (DeltaAll - fabs(divergence[c]));
        }
}

/* principal_stresses_driver weak branch (velocity_solver.F:3497-3516) */
void ora_principal_stresses_weak(int nCellsSolve, const double *stress11, const double *stress22,
                                 const double *stress12, const double *replacementPressure, double *p1, double *p2)
{
    for (int c = 0; c < nCellsSolve; c++) {
        if (replacementPressure[c] > puny) {
            double sq = (stress11[c] + stress22[c]) * (stress11[c] + stress22[c]) - 4.0 * stress11[c] * stress22[c] +
                        4.0 * (stress12[c] * stress12[c]);
            p1[c] = 0.5 * (stress11[c] + stress22[c]) + 0.5 * sqrt(sq);
            p2[c] = 0.5 * (stress11[c] + stress22[c]) - 0.5 * sqrt(sq);
            p1[c] = p1[c] / replacementPressure[c];
            p2[c] = p2[c] / replacementPressure[c];
        } else {
            p1[c] = 1.0e30;
            p2[c] = 1.0e30;
        }
    }
}

/* one block, no halo partner: single_subcycle_velocity_solver (velocity_solver.F:2478-2592) with
 * config_strain_scheme = config_stress_divergence_scheme = 'weak' (seaice_internal_stress :2722-2858) */
void ora_evp_subcycle_loop_weak(int nSubcycles, int nCells, int nVerticesSolve, int nVertices, int maxEdges,
                                int vertexDegree, const int *nEdgesOnCell, const int *verticesOnCell,
                                const int *edgesOnCell, const int *verticesOnEdge, const int *cellsOnVertex,
                                const int *edgesOnVertex, const int *cellsOnEdge, const double *dvEdge,
                                const double *dcEdge, const double *areaCell, const double *areaTriangle,
                                double sphere_radius, const int *solveStress, const int *solveVelocity,
                                const double *normalVectorPolygon, const double *normalVectorTriangle,
                                const double *latCellRotated, const double *latVertexRotated,
                                const double *icePressure, const double *totalMassVertex,
                                const double *totalMassVertexfVertex, const double *iceAreaVertex,
                                const double *airStressVertexU, const double *airStressVertexV,
                                const double *surfaceTiltForceU, const double *surfaceTiltForceV,
                                const double *oceanStressU, const double *oceanStressV,
                                const double *uOceanVelocityVertex, const double *vOceanVelocityVertex,
                                const double *uVelocityInitial, const double *vVelocityInitial, double *uVelocity,
                                double *vVelocity, double *strain11, double *strain22, double *strain12,
                                double *stress11, double *stress22, double *stress12, double *replacementPressure,
                                double *stressDivergenceU, double *stressDivergenceV, double *oceanStressCoeff,
                                double elasticTimeStep, double dynamicsTimeStep, double dampingTimescale,
                                double numericalInertiaCoefficient, int relation, int useOceanStress,
                                int quadraticOceanStress)
{
    for (int it = 0; it < nSubcycles; it++) {
        ora_strain_tensor_weak(nCells, maxEdges, nEdgesOnCell, verticesOnCell, edgesOnCell, verticesOnEdge, dvEdge,
                               areaCell, sphere_radius, strain11, strain22, strain12, uVelocity, vVelocity,
                               normalVectorPolygon, latCellRotated, solveStress);
        ora_stress_tensor_weak(nCells, stress11, stress22, stress12, strain11, strain22, strain12, icePressure,
                               replacementPressure, solveStress, elasticTimeStep, dampingTimescale, relation);
        ora_stress_divergence_weak(nVertices, vertexDegree, cellsOnVertex, edgesOnVertex, cellsOnEdge, areaTriangle,
                                   dcEdge, sphere_radius, stressDivergenceU, stressDivergenceV, stress11, stress22,
                                   stress12, normalVectorTriangle, latVertexRotated, solveVelocity);
        ora_ocean_stress_coefficient(nVerticesSolve, nVertices, useOceanStress, quadraticOceanStress, solveVelocity,
                                     oceanStressCoeff, uOceanVelocityVertex, vOceanVelocityVertex, uVelocity,
                                     vVelocity, iceAreaVertex);
        if (relation == 1)
            ora_solve_velocity(nVerticesSolve, solveVelocity, uVelocity, vVelocity, totalMassVertex,
                               totalMassVertexfVertex, stressDivergenceU, stressDivergenceV, airStressVertexU,
                               airStressVertexV, surfaceTiltForceU, surfaceTiltForceV, oceanStressU, oceanStressV,
                               oceanStressCoeff, elasticTimeStep);
        else if (relation == 2)
            ora_solve_velocity_revised(nVerticesSolve, solveVelocity, uVelocity, vVelocity, uVelocityInitial,
                                       vVelocityInitial, totalMassVertex, totalMassVertexfVertex,
                                       stressDivergenceU, stressDivergenceV, airStressVertexU, airStressVertexV,
                                       surfaceTiltForceU, surfaceTiltForceV, oceanStressU, oceanStressV,
                                       oceanStressCoeff, dynamicsTimeStep, numericalInertiaCoefficient);
    }
}

/* interpolate_strains_weak_to_variational, velocity_solver.F:2877-2960: area-weighted mean of the cell strains
 * at every owned vertex, copied to the corners of every cell */
void ora_interpolate_strains_weak_to_variational(int nCells, int nVerticesSolve, int vertexDegree, int maxEdges,
                                                 const int *cellsOnVertex, const int *nEdgesOnCell,
                                                 const int *verticesOnCell, const double *areaCell,
                                                 const double *s11w, const double *s22w, const double *s12w,
                                                 double *s11v, double *s22v, double *s12v, double *strain11var,
                                                 double *strain22var, double *strain12var)
{
    for (int v = 1; v <= nVerticesSolve; v++) {
        double a = 0.0, b = 0.0, c12 = 0.0, denom = 0.0;
        for (int k = 1; k <= vertexDegree; k++) {
            int c = cellsOnVertex[IX2(k, v, vertexDegree)];
            if (c >= 1 && c <= nCells) {
                a = a + areaCell[c - 1] * s11w[c - 1];
                b = b + areaCell[c - 1] * s22w[c - 1];
                c12 = c12 + areaCell[c - 1] * s12w[c - 1];
                denom = denom + areaCell[c - 1];
            }
        }
        s11v[v - 1] = a / denom;
        s22v[v - 1] = b / denom;
        s12v[v - 1] = c12 / denom;
    }
    for (int c = 1; c <= nCells; c++)
        for (int i = 1; i <= nEdgesOnCell[c - 1]; i++) {
            int v = verticesOnCell[IX2(i, c, maxEdges)];
            strain11var[IX2(i, c, maxEdges)] = s11v[v - 1];
            strain22var[IX2(i, c, maxEdges)] = s22v[v - 1];
            strain12var[IX2(i, c, maxEdges)] = s12v[v - 1];
        }
}

/* one block: weak strain, variational stress and stress divergence (seaice_internal_stress :2722-2858 with
 * config_strain_scheme = 'weak', config_stress_divergence_scheme = 'variational') */
void ora_evp_subcycle_loop_hybrid(int nSubcycles, int nCells, int nVerticesSolve, int nVertices, int maxEdges,
                                  int vertexDegree, const int *nEdgesOnCell, const int *verticesOnCell,
                                  const int *edgesOnCell, const int *verticesOnEdge, const int *cellsOnVertex,
                                  const int *cellVerticesAtVertex, const double *dvEdge, const double *areaCell,
                                  double sphere_radius, const int *solveStress, const int *solveVelocity,
                                  const double *normalVectorPolygon, const double *latCellRotated,
                                  const double *basisIntegralsU, const double *basisIntegralsV,
                                  const double *basisIntegralsMetric, const double *variationalDenominator,
                                  const double *tanLatVertexRotatedOverRadius, const double *icePressure,
                                  const double *totalMassVertex, const double *totalMassVertexfVertex,
                                  const double *iceAreaVertex, const double *airStressVertexU,
                                  const double *airStressVertexV, const double *surfaceTiltForceU,
                                  const double *surfaceTiltForceV, const double *oceanStressU,
                                  const double *oceanStressV, const double *uOceanVelocityVertex,
                                  const double *vOceanVelocityVertex, const double *uVelocityInitial,
                                  const double *vVelocityInitial, double *uVelocity, double *vVelocity,
                                  double *s11w, double *s22w, double *s12w, double *s11v, double *s22v, double *s12v,
                                  double *strain11, double *strain22, double *strain12, double *stress11,
                                  double *stress22, double *stress12, double *replacementPressure,
                                  double *stressDivergenceU, double *stressDivergenceV, double *oceanStressCoeff,
                                  double elasticTimeStep, double dynamicsTimeStep, double dampingTimescale,
                                  double numericalInertiaCoefficient, int relation, int useOceanStress,
                                  int quadraticOceanStress)
{
    for (int it = 0; it < nSubcycles; it++) {
        ora_strain_tensor_weak(nCells, maxEdges, nEdgesOnCell, verticesOnCell, edgesOnCell, verticesOnEdge, dvEdge,
                               areaCell, sphere_radius, s11w, s22w, s12w, uVelocity, vVelocity, normalVectorPolygon,
                               latCellRotated, solveStress);
        ora_interpolate_strains_weak_to_variational(nCells, nVerticesSolve, vertexDegree, maxEdges, cellsOnVertex,
                                                    nEdgesOnCell, verticesOnCell, areaCell, s11w, s22w, s12w, s11v,
                                                    s22v, s12v, strain11, strain22, strain12);
        ora_stress_tensor_variational(nCells, maxEdges, nEdgesOnCell, stress11, stress22, stress12, strain11,
                                      strain22, strain12, icePressure, replacementPressure, solveStress,
                                      elasticTimeStep, dampingTimescale, relation);
        ora_stress_divergence_variational(nVerticesSolve, vertexDegree, maxEdges, nEdgesOnCell, cellsOnVertex,
                                          cellVerticesAtVertex, stressDivergenceU, stressDivergenceV, stress11,
                                          stress22, stress12, basisIntegralsU, basisIntegralsV,
                                          basisIntegralsMetric, variationalDenominator,
                                          tanLatVertexRotatedOverRadius, solveVelocity);
        ora_ocean_stress_coefficient(nVerticesSolve, nVertices, useOceanStress, quadraticOceanStress, solveVelocity,
                                     oceanStressCoeff, uOceanVelocityVertex, vOceanVelocityVertex, uVelocity,
                                     vVelocity, iceAreaVertex);
        if (relation == 1)
            ora_solve_velocity(nVerticesSolve, solveVelocity, uVelocity, vVelocity, totalMassVertex,
                               totalMassVertexfVertex, stressDivergenceU, stressDivergenceV, airStressVertexU,
                               airStressVertexV, surfaceTiltForceU, surfaceTiltForceV, oceanStressU, oceanStressV,
                               oceanStressCoeff, elasticTimeStep);
        else if (relation == 2)
            ora_solve_velocity_revised(nVerticesSolve, solveVelocity, uVelocity, vVelocity, uVelocityInitial,
                                       vVelocityInitial, totalMassVertex, totalMassVertexfVertex,
                                       stressDivergenceU, stressDivergenceV, airStressVertexU, airStressVertexV,
                                       surfaceTiltForceU, surfaceTiltForceV, oceanStressU, oceanStressV,
                                       oceanStressCoeff, dynamicsTimeStep, numericalInertiaCoefficient);
    }
}

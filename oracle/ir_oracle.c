/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference's incremental-remap (IR)
 * tracer advection, src/shared/mpas_seaice_advection_incremental_remap.F.
 *
 * Nothing in the product path may call into this file: only tests/, the smoke check and the
 * cpu_baseline leg of bench.py use it, as the checker / reported baseline.
 *
 * PARITY STATUS: "parity unpinned" at the bit level.  The reference ships no golden vectors for
 * this path and its Fortran cannot be built here (it needs the MPAS framework modules and the
 * framework operator mpas_init_reconstruct, neither of which is in the repository).  The
 * restatement is pinned by the invariants the reference itself checks at run time: conservation of
 * mass and mass*tracer to 1e-11 relative (check_tracer_conservation, :8126-8253), tracer
 * monotonicity (check_tracer_monotonicity, :8416), exact transport of uniform fields, and the
 * exactness of the 6-point triangle rule for quartic polynomials (:6504-6544).
 *
 * Every routine follows the arithmetic of the reference routine cited above it statement by
 * statement, in the same evaluation order.  Arrays use the reference's Fortran layout
 * (column-major, first index fastest, one extra trailing slot per mesh dimension) and its 1-based
 * connectivity.  A "2D" tracer (nCategories, nCells) is a tracer with L = 1 layers here; the
 * reference's separate 2D / 3D routines perform the same element-wise arithmetic.
 * Only vertexDegree = 3 (hexagonal meshes) is restated; the quadrilateral-mesh branches
 * (:5807-5958, :1479-1507, :1541-1580) are not.
 * Build with -ffp-contract=off so each source-level operation rounds once.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define NTRI 6   /* nTriPerEdgeRemap, Registry.xml:63 */
#define NQP 6    /* nQuadPoints, Registry.xml:59 */
#define NVER 8   /* maxVerticesPerEdgeRemap, Registry.xml:75 */
#define NEER 6   /* maxEdgesPerEdgeRemap / maxCellsPerEdgeRemap */

/* :59-70 */
static const double eps11 = 1.0e-11;
static const double w1QP = 1.09951743655321885e-01, w2QP = 2.23381589678011389e-01;
static const double q1QP = 9.15762135097710761e-02, q2QP = 8.16847572980458514e-01,
                    q3QP = 1.08103018168070275e-01, q4QP = 4.45948490915965612e-01;

/* element (l, cat, cell) of a (L, nCat, nCells+1) array, all 1-based */
#define F3(l, cat, c, L, NC) ((((size_t)(c) - 1) * (size_t)(NC) + (size_t)((cat) - 1)) * (size_t)(L) + (size_t)((l) - 1))
/* same element of the parent array, which has either 1 or L layers */
#define P3(l, cat, c, Lp, NC) F3(((Lp) == 1 ? 1 : (l)), cat, c, Lp, NC)
#define I2(i, c, M) ((size_t)((c) - 1) * (size_t)(M) + (size_t)((i) - 1))

/* ------------------------------------------------------------------ small geometry helpers */

/* :8900-8918 */
static void unit_vector_3d(double v[3])
{
    double mag = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (mag > 0.0) {
        v[0] = v[0] / mag;
        v[1] = v[1] / mag;
        v[2] = v[2] / mag;
    } else {
        v[0] = v[1] = v[2] = 0.0;
    }
}

/* :9109-9123 */
static void cross_product_3d(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* :9083-9095 */
static double cross_product_2d(const double a[2], const double b[2]) { return a[0] * b[1] - a[1] * b[0]; }

/* :9201-9234; note the dummy-argument order (point, lineStart, lineEnd) */
static int point_in_half_plane(const double point[2], const double lineStart[2], const double lineEnd[2])
{
    double v1[2], v2[2];
    v1[0] = lineEnd[0] - lineStart[0];
    v1[1] = lineEnd[1] - lineStart[1];
    v2[0] = point[0] - lineStart[0];
    v2[1] = point[1] - lineStart[1];
    return cross_product_2d(v1, v2) >= 0.0;
}

/* :9172-9185 */
static double triangle_area(const double a[2], const double b[2], const double c[2])
{
    return fabs(0.5 * ((b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])));
}

/* :9137-9158 */
static double quadrilateral_area(const double a[2], const double b[2], const double c[2], const double d[2])
{
    double t1 = triangle_area(a, b, c);
    double t2 = triangle_area(a, c, d);
    return t1 + t2;
}

/* :8934-9016 */
static int find_line_intersection(const double p1[2], const double p2[2], const double p3[2], const double p4[2],
                                  double ip[2])
{
    double x1 = p1[0], y1 = p1[1], x2 = p2[0], y2 = p2[1], x3 = p3[0], y3 = p3[1], x4 = p4[0], y4 = p4[1];
    double rx = x2 - x1, ry = y2 - y1, sx = x4 - x3, sy = y4 - y3;
    double rsCross = rx * sy - ry * sx;
    double rsCrossMin = eps11 * sqrt((rx * rx + ry * ry) * (sx * sx + sy * sy));
    int hit = 0;
    if (fabs(rsCross) > rsCrossMin) {
        double t1 = (sy * (x3 - x1) - sx * (y3 - y1)) / rsCross;
        double t2 = (ry * (x3 - x1) - rx * (y3 - y1)) / rsCross;
        ip[0] = x1 + t1 * rx;
        ip[1] = y1 + t1 * ry;
        if (t1 > 0.0 && t1 < 1.0 && t2 > 0.0 && t2 < 1.0) hit = 1;
    } else {
        ip[0] = 1.79769313486231571e308; /* huge(0.0_RKIND) */
        ip[1] = 1.79769313486231571e308;
    }
    return hit;
}

/* ------------------------------------------------------------------ initialisation */

/* rotate_global_vectors :948-975 */
void ira_rotate_global_vectors(const double *x, const double *y, const double *z, double *xr, double *yr, double *zr,
                               int n)
{
    for (int i = 0; i < n; i++) {
        xr[i] = -z[i];
        yr[i] = y[i];
        zr[i] = x[i];
    }
}

/* define_local_to_global_transformations :990-1084.  trans*(3,3,n): element (i,j,e) at (e*3 + j)*3 + i.
 * matmul with the identity reduces to a transposition / copy. */
void ira_define_transformations(const double *xg, const double *yg, const double *zg, int n, double *l2g, double *g2l)
{
    for (int e = 0; e < n; e++) {
        double ul[3][3]; /* ul[j][i] = unitVectorLocal(i+1, j+1) */
        ul[2][0] = xg[e];
        ul[2][1] = yg[e];
        ul[2][2] = zg[e];
        unit_vector_3d(ul[2]);
        if (fabs(xg[e] * xg[e] + yg[e] * yg[e]) > eps11) {
            ul[0][0] = -yg[e];
            ul[0][1] = xg[e];
            ul[0][2] = 0.0;
            unit_vector_3d(ul[0]);
            cross_product_3d(ul[2], ul[0], ul[1]);
        } else if (zg[e] > 0.0) {
            ul[0][0] = 1.0; ul[0][1] = 0.0; ul[0][2] = 0.0;
            ul[1][0] = 0.0; ul[1][1] = 1.0; ul[1][2] = 0.0;
        } else {
            ul[0][0] = 0.0; ul[0][1] = 1.0; ul[0][2] = 0.0;
            ul[1][0] = 1.0; ul[1][1] = 0.0; ul[1][2] = 0.0;
        }
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                l2g[((size_t)e * 3 + j) * 3 + i] = ul[j][i]; /* transLocalToGlobal(i,j) = unitVectorLocal(i,j) */
                g2l[((size_t)e * 3 + j) * 3 + i] = ul[i][j]; /* transGlobalToLocal(i,j) = unitVectorLocal(j,i) */
            }
    }
}

#define T(t, i, j, e) (t)[(((size_t)(e) - 1) * 3 + ((j) - 1)) * 3 + ((i) - 1)]

/* get_vertex_on_cell_coordinates :1823-2036; returns the number of cells failing the CCW checks */
int ira_vertex_on_cell(int nCells, int M, const int *nEdgesOnCell, const int *verticesOnCell, const double *xV,
                       const double *yV, const double *zV, const double *xC, const double *yC, const double *zC,
                       int onSphere, const double *g2c, double *xVOC, double *yVOC)
{
    int bad = 0;
    for (int c = 1; c <= nCells; c++)
        for (int i = 1; i <= nEdgesOnCell[c - 1]; i++) {
            int v = verticesOnCell[I2(i, c, M)];
            if (onSphere) {
                double w[3] = {xV[v - 1] - xC[c - 1], yV[v - 1] - yC[c - 1], zV[v - 1] - zC[c - 1]};
                xVOC[I2(i, c, M)] = T(g2c, 1, 1, c) * w[0] + T(g2c, 1, 2, c) * w[1] + T(g2c, 1, 3, c) * w[2];
                yVOC[I2(i, c, M)] = T(g2c, 2, 1, c) * w[0] + T(g2c, 2, 2, c) * w[1] + T(g2c, 2, 3, c) * w[2];
            } else {
                xVOC[I2(i, c, M)] = xV[v - 1] - xC[c - 1];
                yVOC[I2(i, c, M)] = yV[v - 1] - yC[c - 1];
            }
        }
    for (int c = 1; c <= nCells; c++) {
        int n = nEdgesOnCell[c - 1], fail = 0;
        for (int i = 1; i <= n; i++) {
            int ip1 = i + 1 > n ? 1 : i + 1;
            if (onSphere) {
                int v = verticesOnCell[I2(i, c, M)], v2 = verticesOnCell[I2(ip1, c, M)];
                double w1[3] = {xV[v - 1] - xC[c - 1], yV[v - 1] - yC[c - 1], zV[v - 1] - zC[c - 1]};
                double w2[3] = {xV[v2 - 1] - xC[c - 1], yV[v2 - 1] - yC[c - 1], zV[v2 - 1] - zC[c - 1]};
                double w[3], w3[3] = {xC[c - 1], yC[c - 1], zC[c - 1]};
                cross_product_3d(w1, w2, w);
                if (!(w[0] * w3[0] + w[1] * w3[1] + w[2] * w3[2] >= 0.0)) fail = 1;
            }
            {
                double a[2] = {xVOC[I2(i, c, M)], yVOC[I2(i, c, M)]}, b[2] = {xVOC[I2(ip1, c, M)], yVOC[I2(ip1, c, M)]};
                if (!(cross_product_2d(a, b) >= 0.0)) fail = 1;
            }
        }
        bad += fail;
    }
    return bad;
}

/* get_geometry_incremental_remap :1105-1808, hexagonal (vertexDegree = 3) and quadrilateral (4) meshes.  Returns
 * the number of edges whose cell 1 is not in the left half-plane (a fatal mesh error in the reference, :1345-1350). */
int ira_get_geometry(int nEdges, int nVertices, int nCells, int nCellsSolve, int M, const int *nEdgesOnCell,
                     const double *xC, const double *yC, const double *zC, const double *xV, const double *yV,
                     const double *zV, const double *xE, const double *yE, const double *zE, const int *cellsOnEdge,
                     const int *verticesOnEdge, const int *edgesOnCell, const int *edgesOnVertex, int vertexDegree,
                     int onSphere, const double *g2e, int *remapEdge, int *cellsOnEdgeRemap, int *edgesOnEdgeRemap, double *xVOE,
                     double *yVOE, double *minLengthEdgesOnVertex)
{
    int bad = 0;
    const int nEdgesOnEdgeRemap = vertexDegree == 4 ? 6 : 4; /* :1587-1591 */
    for (int e = 0; e <= nEdges; e++) remapEdge[e] = 0;
    for (int c = 1; c <= nCellsSolve; c++)
        for (int i = 1; i <= nEdgesOnCell[c - 1]; i++) remapEdge[edgesOnCell[I2(i, c, M)] - 1] = 1;
    for (int e = 1; e <= nEdges; e++)
        if (remapEdge[e - 1] == 1)
            for (int k = 1; k <= 2; k++) {
                int c = cellsOnEdge[I2(k, e, 2)];
                if (c < 1 || c > nCells) {
                    remapEdge[e - 1] = 0;
                    break;
                }
            }
    /* cell 1 must lie in the left half-plane of V1 -> V2, :1296-1395 */
    for (int e = 1; e <= nEdges; e++)
        if (remapEdge[e - 1] == 1) {
            double ev[2][2], cc[2];
            int c = cellsOnEdge[I2(1, e, 2)];
            for (int k = 1; k <= 2; k++) {
                int v = verticesOnEdge[I2(k, e, 2)];
                if (onSphere) {
                    double w[3] = {xV[v - 1] - xE[e - 1], yV[v - 1] - yE[e - 1], zV[v - 1] - zE[e - 1]};
                    ev[k - 1][0] = T(g2e, 1, 1, e) * w[0] + T(g2e, 1, 2, e) * w[1] + T(g2e, 1, 3, e) * w[2];
                    ev[k - 1][1] = T(g2e, 2, 1, e) * w[0] + T(g2e, 2, 2, e) * w[1] + T(g2e, 2, 3, e) * w[2];
                } else {
                    ev[k - 1][0] = xV[v - 1] - xE[e - 1];
                    ev[k - 1][1] = yV[v - 1] - yE[e - 1];
                }
            }
            if (c >= 1 && c <= nCells) {
                if (onSphere) {
                    double w[3] = {xC[c - 1] - xE[e - 1], yC[c - 1] - yE[e - 1], zC[c - 1] - zE[e - 1]};
                    cc[0] = T(g2e, 1, 1, e) * w[0] + T(g2e, 1, 2, e) * w[1] + T(g2e, 1, 3, e) * w[2];
                    cc[1] = T(g2e, 2, 1, e) * w[0] + T(g2e, 2, 2, e) * w[1] + T(g2e, 2, 3, e) * w[2];
                } else {
                    cc[0] = xC[c - 1] - xE[e - 1];
                    cc[1] = yC[c - 1] - yE[e - 1];
                }
                if (!point_in_half_plane(ev[0], ev[1], cc)) bad++;
            }
        }

    /* neighbour edges and cells, :1406-1583 */
    memset(edgesOnEdgeRemap, 0, sizeof(int) * NEER * ((size_t)nEdges + 1));
    memset(cellsOnEdgeRemap, 0, sizeof(int) * NEER * ((size_t)nEdges + 1));
    for (int e = 1; e <= nEdges; e++)
        if (remapEdge[e - 1] == 1) {
            for (int k = 1; k <= 2; k++) cellsOnEdgeRemap[I2(k, e, NEER)] = cellsOnEdge[I2(k, e, 2)];
            for (int k = 1; k <= 2; k++) {
                int c = cellsOnEdge[I2(k, e, 2)];
                if (c >= 1 && c <= nCells) {
                    int n = nEdgesOnCell[c - 1], iMain = 0, i;
                    for (i = 1; i <= n; i++)
                        if (edgesOnCell[I2(i, c, M)] == e) {
                            iMain = i;
                            break;
                        }
                    if (k == 1) {
                        i = iMain - 1;
                        if (i < 1) i += n;
                        edgesOnEdgeRemap[I2(1, e, NEER)] = edgesOnCell[I2(i, c, M)];
                        i = iMain + 1;
                        if (i > n) i -= n;
                        edgesOnEdgeRemap[I2(2, e, NEER)] = edgesOnCell[I2(i, c, M)];
                    } else {
                        i = iMain + 1;
                        if (i > n) i -= n;
                        edgesOnEdgeRemap[I2(3, e, NEER)] = edgesOnCell[I2(i, c, M)];
                        i = iMain - 1;
                        if (i < 1) i += n;
                        edgesOnEdgeRemap[I2(4, e, NEER)] = edgesOnCell[I2(i, c, M)];
                    }
                }
            }
            if (vertexDegree == 4) { /* E5 and E6: the edge at V1 / V2 not picked up yet, :1479-1507 */
                for (int k = 1; k <= 2; k++) {
                    int v = verticesOnEdge[I2(k, e, 2)];
                    for (int j = 1; j <= vertexDegree; j++) {
                        int en = edgesOnVertex[I2(j, v, vertexDegree)];
                        if (en >= 1 && en <= nEdges) {
                            int isNew = 1;
                            for (int q = 1; q <= 4; q++)
                                if (en == edgesOnEdgeRemap[I2(q, e, NEER)] || en == e) {
                                    isNew = 0;
                                    break;
                                }
                            if (isNew) {
                                edgesOnEdgeRemap[I2(k + 4, e, NEER)] = en;
                                break;
                            }
                        }
                    }
                }
            }
            if (vertexDegree == 3) { /* C3 and C4, :1513-1539 */
                cellsOnEdgeRemap[I2(3, e, NEER)] = nCells + 1;
                cellsOnEdgeRemap[I2(4, e, NEER)] = nCells + 1;
                for (int k = 1; k <= 2; k++) {
                    int en = edgesOnEdgeRemap[I2(k, e, NEER)];
                    if (en < 1 || en > nEdges) en = edgesOnEdgeRemap[I2(k + 2, e, NEER)];
                    for (int j = 1; j <= 2; j++) {
                        int cn = cellsOnEdge[I2(j, en, 2)];
                        if (cn >= 1 && cn <= nCells)
                            if (cn != cellsOnEdgeRemap[I2(1, e, NEER)] && cn != cellsOnEdgeRemap[I2(2, e, NEER)])
                                cellsOnEdgeRemap[I2(k + 2, e, NEER)] = cn;
                    }
                }
            } else { /* C3, C4 beyond E1, E2 (not C1); C5, C6 beyond E3, E4 (not C2), :1541-1578 */
                for (int q = 3; q <= 6; q++) cellsOnEdgeRemap[I2(q, e, NEER)] = nCells + 1;
                for (int k = 1; k <= 2; k++) {
                    int en = edgesOnEdgeRemap[I2(k, e, NEER)];
                    if (en >= 1 && en <= nEdges)
                        for (int j = 1; j <= 2; j++) {
                            int cn = cellsOnEdge[I2(j, en, 2)];
                            if (cn >= 1 && cn <= nCells && cn != cellsOnEdgeRemap[I2(1, e, NEER)])
                                cellsOnEdgeRemap[I2(k + 2, e, NEER)] = cn;
                        }
                    en = edgesOnEdgeRemap[I2(k + 2, e, NEER)];
                    if (en >= 1 && en <= nEdges)
                        for (int j = 1; j <= 2; j++) {
                            int cn = cellsOnEdge[I2(j, en, 2)];
                            if (cn >= 1 && cn <= nCells && cn != cellsOnEdgeRemap[I2(2, e, NEER)])
                                cellsOnEdgeRemap[I2(k + 4, e, NEER)] = cn;
                        }
                }
            }
        }

    /* vertex coordinates relative to each edge, :1587-1762 */
    if (onSphere) {
        for (int e = 1; e <= nEdges; e++) {
            int v1 = verticesOnEdge[I2(1, e, 2)], v2 = verticesOnEdge[I2(2, e, 2)];
            double w[3] = {xV[v2 - 1] - xV[v1 - 1], yV[v2 - 1] - yV[v1 - 1], zV[v2 - 1] - zV[v1 - 1]};
            double xv = T(g2e, 1, 1, e) * w[0] + T(g2e, 1, 2, e) * w[1] + T(g2e, 1, 3, e) * w[2];
            double yv = T(g2e, 2, 1, e) * w[0] + T(g2e, 2, 2, e) * w[1] + T(g2e, 2, 3, e) * w[2];
            xVOE[I2(1, e, NVER)] = -0.5 * xv;
            yVOE[I2(1, e, NVER)] = -0.5 * yv;
            xVOE[I2(2, e, NVER)] = 0.5 * xv;
            yVOE[I2(2, e, NVER)] = 0.5 * yv;
        }
        for (int e = 1; e <= nEdges; e++)
            if (remapEdge[e - 1] == 1)
                for (int k = 1; k <= nEdgesOnEdgeRemap; k++) {
                    int en = edgesOnEdgeRemap[I2(k, e, NEER)];
                    if (en >= 1 && en <= nEdges) {
                        int n1 = 0, m1 = 0, m2 = 0;
                        for (int n = 1; n <= 2; n++)
                            for (int m = 1; m <= 2; m++)
                                if (verticesOnEdge[I2(m, en, 2)] == verticesOnEdge[I2(n, e, 2)]) {
                                    n1 = n;
                                    m1 = m;
                                    m2 = m == 1 ? 2 : 1;
                                    break;
                                }
                        xVOE[I2(2 + k, e, NVER)] = xVOE[I2(n1, e, NVER)] + (xVOE[I2(m2, en, NVER)] - xVOE[I2(m1, en, NVER)]);
                        yVOE[I2(2 + k, e, NVER)] = yVOE[I2(n1, e, NVER)] + (yVOE[I2(m2, en, NVER)] - yVOE[I2(m1, en, NVER)]);
                    }
                }
    } else {
        for (int e = 1; e <= nEdges; e++)
            if (remapEdge[e - 1] == 1) {
                int count = 2;
                for (int k = 1; k <= 2; k++) {
                    int v = verticesOnEdge[I2(k, e, 2)];
                    xVOE[I2(k, e, NVER)] = xV[v - 1] - xE[e - 1];
                    yVOE[I2(k, e, NVER)] = yV[v - 1] - yE[e - 1];
                }
                for (int k = 1; k <= nEdgesOnEdgeRemap; k++) {
                    int en = edgesOnEdgeRemap[I2(k, e, NEER)];
                    if (en >= 1 && en <= nEdges) {
                        int v = 0;
                        for (int n = 1; n <= 2; n++)
                            for (int m = 1; m <= 2; m++)
                                if (verticesOnEdge[I2(m, en, 2)] == verticesOnEdge[I2(n, e, 2)]) {
                                    v = verticesOnEdge[I2(m == 1 ? 2 : 1, en, 2)];
                                    count = count + 1;
                                    break;
                                }
                        xVOE[I2(count, e, NVER)] = xV[v - 1] - xE[e - 1];
                        yVOE[I2(count, e, NVER)] = yV[v - 1] - yE[e - 1];
                    }
                }
            }
    }

    /* :1764-1806 */
    for (int v = 1; v <= nVertices; v++) {
        double mn = 1.79769313486231571e308;
        for (int k = 1; k <= vertexDegree; k++) {
            int e = edgesOnVertex[I2(k, v, vertexDegree)];
            if (e >= 1 && e <= nEdges) {
                int v1 = verticesOnEdge[I2(1, e, 2)], v2 = verticesOnEdge[I2(2, e, 2)];
                double w[3] = {xV[v2 - 1] - xV[v1 - 1], yV[v2 - 1] - yV[v1 - 1], zV[v2 - 1] - zV[v1 - 1]};
                double len = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
                if (len < mn) mn = len;
            }
        }
        minLengthEdgesOnVertex[v - 1] = mn;
    }
    return bad;
}

/* compute_geometric_cell_averages :2097-2323.  geom(nCells+1, 14): x y xx xy yy xxx xxy xyy yyy
 * xxxx xxxy xxyy xyyy yyyy.  x**3 is taken as (x*x)*x and x**4 as (x*x)*(x*x). */
void ira_geometric_cell_averages(const double *xVOC, const double *yVOC, int nCells, int M, const int *edgesOnCell,
                                 const int *nEdgesOnCell, const double *dcEdge, const double *dvEdge, double *geom)
{
    size_t ld = (size_t)nCells + 1;
    for (int c = 1; c <= nCells; c++) {
        double frac[16], sumArea = 0.0, acc[14];
        int n = nEdgesOnCell[c - 1];
        for (int i = 1; i <= n; i++) {
            int e = edgesOnCell[I2(i, c, M)];
            frac[i - 1] = 0.25 * dcEdge[e - 1] * dvEdge[e - 1];
            sumArea = sumArea + frac[i - 1];
        }
        for (int i = 1; i <= n; i++) frac[i - 1] = frac[i - 1] / sumArea;
        for (int k = 0; k < 14; k++) acc[k] = 0.0;
        for (int i = 1; i <= n; i++) {
            double x1 = 0.0, y1 = 0.0, x2, y2, x3, y3, xq[6], yq[6], wq[6], av[14];
            int ip1 = i + 1 > n ? 1 : i + 1;
            x2 = xVOC[I2(i, c, M)];
            y2 = yVOC[I2(i, c, M)];
            x3 = xVOC[I2(ip1, c, M)];
            y3 = yVOC[I2(ip1, c, M)];
            xq[0] = q1QP * x1 + q1QP * x2 + q2QP * x3;  yq[0] = q1QP * y1 + q1QP * y2 + q2QP * y3;
            xq[1] = q1QP * x1 + q2QP * x2 + q1QP * x3;  yq[1] = q1QP * y1 + q2QP * y2 + q1QP * y3;
            xq[2] = q2QP * x1 + q1QP * x2 + q1QP * x3;  yq[2] = q2QP * y1 + q1QP * y2 + q1QP * y3;
            xq[3] = q3QP * x1 + q4QP * x2 + q4QP * x3;  yq[3] = q3QP * y1 + q4QP * y2 + q4QP * y3;
            xq[4] = q4QP * x1 + q3QP * x2 + q4QP * x3;  yq[4] = q4QP * y1 + q3QP * y2 + q4QP * y3;
            xq[5] = q4QP * x1 + q4QP * x2 + q3QP * x3;  yq[5] = q4QP * y1 + q4QP * y2 + q3QP * y3;
            wq[0] = wq[1] = wq[2] = w1QP;
            wq[3] = wq[4] = wq[5] = w2QP;
            for (int k = 0; k < 14; k++) av[k] = 0.0;
            for (int q = 0; q < 6; q++) {
                double x = xq[q], y = yq[q], w = wq[q];
                double xx = x * x, yy = y * y, xxx = xx * x, yyy = yy * y, xxxx = xx * xx, yyyy = yy * yy;
                av[0] = av[0] + w * x;
                av[1] = av[1] + w * y;
                av[2] = av[2] + w * xx;
                av[3] = av[3] + w * x * y;
                av[4] = av[4] + w * yy;
                av[5] = av[5] + w * xxx;
                av[6] = av[6] + w * xx * y;
                av[7] = av[7] + w * x * yy;
                av[8] = av[8] + w * yyy;
                av[9] = av[9] + w * xxxx;
                av[10] = av[10] + w * xxx * y;
                av[11] = av[11] + w * xx * yy;
                av[12] = av[12] + w * x * yyy;
                av[13] = av[13] + w * yyyy;
            }
            for (int k = 0; k < 14; k++) acc[k] = acc[k] + frac[i - 1] * av[k];
        }
        for (int k = 0; k < 14; k++) geom[(size_t)k * ld + (c - 1)] = acc[k];
    }
}

/* ------------------------------------------------------------------ per-step routines */

/* volume_to_thickness :9248-9281 / thickness_to_volume :9295-9324 over every allocated column */
void ira_volume_to_thickness(int nCols, int nCat, const double *area, double *vol)
{
    for (size_t k = 0; k < (size_t)nCols * nCat; k++) vol[k] = area[k] > 0.0 ? vol[k] / area[k] : 0.0;
}

void ira_thickness_to_volume(int nCols, int nCat, const double *area, double *h)
{
    for (size_t k = 0; k < (size_t)nCols * nCat; k++) h[k] = area[k] * h[k];
}

/* make_masks :3404-3565.  Mass-like field (parent == NULL): mask 1 in every real cell and
 * maskCell = (sum over categories > 0).  Otherwise mask = (parent value > threshold). */
void ira_make_masks(int nCells, int L, int nCat, int Lp, const double *array, const double *parent, double threshold,
                    int *mask, int *maskCell)
{
    memset(mask, 0, sizeof(int) * (size_t)L * nCat * ((size_t)nCells + 1));
    if (!parent) {
        for (int c = 1; c <= nCells; c++) {
            double s = 0.0;
            for (int cat = 1; cat <= nCat; cat++)
                for (int l = 1; l <= L; l++) {
                    mask[F3(l, cat, c, L, nCat)] = 1;
                    s = s + array[F3(l, cat, c, L, nCat)];
                }
            maskCell[c - 1] = s > 0.0 ? 1 : 0;
        }
    } else {
        for (int c = 1; c <= nCells; c++)
            for (int cat = 1; cat <= nCat; cat++)
                for (int l = 1; l <= L; l++)
                    if (parent[P3(l, cat, c, Lp, nCat)] > threshold) mask[F3(l, cat, c, L, nCat)] = 1;
    }
}

/* compute_gradient_2d :4204-4414 / compute_gradient_3d :4429-4642 */
void ira_compute_gradient(int nCells, int M, int L, int nCat, const int *nEdgesOnCell, const int *edgesOnCell,
                          const int *cellsOnCell, const int *cellsOnEdge, const double *dcEdge, const double *coeffs,
                          int onSphere, int rotate, const double *g2c, const int *maskCell, const double *field,
                          const int *mask, double *xGrad, double *yGrad)
{
    size_t np = (size_t)L * nCat;
    memset(xGrad, 0, sizeof(double) * np * ((size_t)nCells + 1));
    memset(yGrad, 0, sizeof(double) * np * ((size_t)nCells + 1));
#pragma omp parallel
    {
        double *normalGrad = (double *)malloc(sizeof(double) * np);
        double *g = (double *)malloc(sizeof(double) * np * 3);
#pragma omp for
        for (int c = 1; c <= nCells; c++) {
            if (maskCell[c - 1] != 1) continue;
            for (size_t p = 0; p < np * 3; p++) g[p] = 0.0;
            for (int ie = 1; ie <= nEdgesOnCell[c - 1]; ie++) {
                int nb = cellsOnCell[I2(ie, c, M)];
                for (size_t p = 0; p < np; p++) normalGrad[p] = 0.0;
                if (nb >= 1 && nb <= nCells) {
                    int e = edgesOnCell[I2(ie, c, M)];
                    double sg = (c == cellsOnEdge[I2(1, e, 2)]) ? 1.0 : -1.0;
                    for (size_t p = 0; p < np; p++)
                        if (mask[((size_t)c - 1) * np + p] == 1 && mask[((size_t)nb - 1) * np + p] == 1)
                            normalGrad[p] =
                                sg * (field[((size_t)nb - 1) * np + p] - field[((size_t)c - 1) * np + p]) / dcEdge[e - 1];
                }
                for (int k = 0; k < 3; k++) {
                    double cf = coeffs[((size_t)(c - 1) * M + (ie - 1)) * 3 + k];
                    for (size_t p = 0; p < np; p++) g[k * np + p] = g[k * np + p] + cf * normalGrad[p];
                }
            }
            if (rotate && onSphere)
                for (size_t p = 0; p < np; p++) {
                    double t = g[p];
                    g[p] = -g[2 * np + p];
                    g[2 * np + p] = t;
                }
            for (size_t p = 0; p < np; p++) {
                if (onSphere) {
                    xGrad[((size_t)c - 1) * np + p] =
                        T(g2c, 1, 1, c) * g[p] + T(g2c, 1, 2, c) * g[np + p] + T(g2c, 1, 3, c) * g[2 * np + p];
                    yGrad[((size_t)c - 1) * np + p] =
                        T(g2c, 2, 1, c) * g[p] + T(g2c, 2, 2, c) * g[np + p] + T(g2c, 2, 3, c) * g[2 * np + p];
                } else {
                    xGrad[((size_t)c - 1) * np + p] = g[p];
                    yGrad[((size_t)c - 1) * np + p] = g[np + p];
                }
            }
        }
        free(normalGrad);
        free(g);
    }
}

/* limit_tracer_gradient_2d :4802-4984 / _3d :4999-5240.  xBary/yBary have Lb (1 or L) layers. */
void ira_limit_gradient(int nCells, int M, int L, int nCat, int Lb, const int *nEdgesOnCell, const int *cellsOnCell,
                        const double *xVOC, const double *yVOC, const int *maskCell, const double *field,
                        const int *mask, const double *xBary, const double *yBary, double *xGrad, double *yGrad)
{
#pragma omp parallel for
    for (int c = 1; c <= nCells; c++) {
        if (maskCell[c - 1] != 1) continue;
        int n = nEdgesOnCell[c - 1];
        for (int cat = 1; cat <= nCat; cat++)
            for (int l = 1; l <= L; l++) {
                size_t k = F3(l, cat, c, L, nCat), kb = P3(l, cat, c, Lb, nCat);
                double mx = field[k], mn = field[k], lmx = 0.0, lmn = 0.0, f1, f2, f;
                for (int ie = 1; ie <= n; ie++) {
                    int nb = cellsOnCell[I2(ie, c, M)];
                    size_t kn = F3(l, cat, nb, L, nCat);
                    if (mask[kn] == 1) {
                        mx = fmax(mx, field[kn]);
                        mn = fmin(mn, field[kn]);
                    }
                }
                mx = mx - field[k];
                mn = mn - field[k];
                for (int iv = 1; iv <= n; iv++) {
                    double dev = xGrad[k] * (xVOC[I2(iv, c, M)] - xBary[kb]) + yGrad[k] * (yVOC[I2(iv, c, M)] - yBary[kb]);
                    lmx = fmax(lmx, dev);
                    lmn = fmin(lmn, dev);
                }
                f1 = fabs(lmx) > fabs(mx) ? fmax(0.0, mx / lmx) : 1.0;
                f2 = fabs(lmn) > fabs(mn) ? fmax(0.0, mn / lmn) : 1.0;
                f = fmin(f1, f2);
                f = fmax(0.0, f - eps11);
                xGrad[k] = xGrad[k] * f;
                yGrad[k] = yGrad[k] * f;
            }
    }
}

/* value at the geometric cell centre, :3785-3791, :3930-3969 */
void ira_center_value(int nCells, int L, int nCat, int Lb, const double *field, const double *xGrad,
                      const double *yGrad, const double *xBary, const double *yBary, double *center)
{
    memset(center, 0, sizeof(double) * (size_t)L * nCat * ((size_t)nCells + 1));
    for (int c = 1; c <= nCells; c++)
        for (int cat = 1; cat <= nCat; cat++)
            for (int l = 1; l <= L; l++) {
                size_t k = F3(l, cat, c, L, nCat), kb = P3(l, cat, c, Lb, nCat);
                center[k] = field[k] - xGrad[k] * xBary[kb] - yGrad[k] * yBary[kb];
            }
}

/* dummy parent of the mass-like field: barycentre at the cell centroid, :3704-3730 */
void ira_centroid_barycenter(int nCells, int nCat, const double *geom, double *xB, double *yB)
{
    size_t ld = (size_t)nCells + 1;
    for (int c = 1; c <= nCells; c++)
        for (int cat = 1; cat <= nCat; cat++) {
            xB[F3(1, cat, c, 1, nCat)] = geom[0 * ld + (c - 1)];
            yB[F3(1, cat, c, 1, nCat)] = geom[1 * ld + (c - 1)];
        }
}

/* compute_barycenter_coordinates :4658-4787 applied as in :3802-3883 / :3982-4167.
 * nLev = 1, 2, 3 levels (mass field first); level k has Lk[k] layers and arrays mean/center/gx/gy. */
void ira_barycenter(int nCells, int L, int nCat, const double *geom, const int *maskCell, int nLev, const int *Lk,
                    const double *const *mean, const double *const *cen, const double *const *gx,
                    const double *const *gy, double *xB, double *yB)
{
    size_t ld = (size_t)nCells + 1;
    memset(xB, 0, sizeof(double) * (size_t)L * nCat * ld);
    memset(yB, 0, sizeof(double) * (size_t)L * nCat * ld);
#define G(k) geom[(size_t)(k) * ld + (c - 1)]
    for (int c = 1; c <= nCells; c++) {
        if (maskCell[c - 1] != 1) continue;
        for (int cat = 1; cat <= nCat; cat++)
            for (int l = 1; l <= L; l++) {
                size_t k = F3(l, cat, c, L, nCat);
                size_t k0 = P3(l, cat, c, Lk[0], nCat);
                double mean0 = mean[0][k0], center0 = cen[0][k0], xGrad0 = gx[0][k0], yGrad0 = gy[0][k0];
                double reciprocal;
                if (nLev == 1) {
                    double c0 = center0, cx = xGrad0, cy = yGrad0;
                    reciprocal = fabs(mean0) > 0.0 ? 1.0 / mean0 : 0.0;
                    xB[k] = (c0 * G(0) + cx * G(2) + cy * G(3)) * reciprocal;
                    yB[k] = (c0 * G(1) + cx * G(3) + cy * G(4)) * reciprocal;
                } else if (nLev == 2) {
                    size_t k1 = P3(l, cat, c, Lk[1], nCat);
                    double mean1 = mean[1][k1], center1 = cen[1][k1], xGrad1 = gx[1][k1], yGrad1 = gy[1][k1];
                    double c0 = center0 * center1;
                    double cx = center0 * xGrad1 + xGrad0 * center1;
                    double cy = center0 * yGrad1 + yGrad0 * center1;
                    double cxx = xGrad0 * xGrad1;
                    double cxy = xGrad0 * yGrad1 + yGrad0 * xGrad1;
                    double cyy = yGrad0 * yGrad1;
                    double prod = mean0 * mean1;
                    reciprocal = fabs(prod) > 0.0 ? 1.0 / prod : 0.0;
                    xB[k] = (c0 * G(0) + cx * G(2) + cy * G(3) + cxx * G(5) + cxy * G(6) + cyy * G(7)) * reciprocal;
                    yB[k] = (c0 * G(1) + cx * G(3) + cy * G(4) + cxx * G(6) + cxy * G(7) + cyy * G(8)) * reciprocal;
                } else {
                    size_t k1 = P3(l, cat, c, Lk[1], nCat), k2 = P3(l, cat, c, Lk[2], nCat);
                    double mean1 = mean[1][k1], center1 = cen[1][k1], xGrad1 = gx[1][k1], yGrad1 = gy[1][k1];
                    double mean2 = mean[2][k2], center2 = cen[2][k2], xGrad2 = gx[2][k2], yGrad2 = gy[2][k2];
                    double c0 = center0 * center1 * center2;
                    double cx = center0 * center1 * xGrad2 + center0 * xGrad1 * center2 + xGrad0 * center1 * center2;
                    double cy = center0 * center1 * yGrad2 + center0 * yGrad1 * center2 + yGrad0 * center1 * center2;
                    double cxx = center0 * xGrad1 * xGrad2 + xGrad0 * center1 * xGrad2 + xGrad0 * xGrad1 * center2;
                    double cxy = center0 * xGrad1 * yGrad2 + xGrad0 * yGrad1 * center2 + yGrad0 * center1 * xGrad2 +
                                 center0 * yGrad1 * xGrad2 + xGrad0 * center1 * yGrad2 + yGrad0 * xGrad1 * center2;
                    double cyy = center0 * yGrad1 * yGrad2 + yGrad0 * center1 * yGrad2 + yGrad0 * yGrad1 * center2;
                    double cxxx = xGrad0 * xGrad1 * xGrad2;
                    double cxxy = xGrad0 * xGrad1 * yGrad2 + xGrad0 * yGrad1 * xGrad2 + yGrad0 * xGrad1 * xGrad2;
                    double cxyy = yGrad0 * yGrad1 * xGrad2 + yGrad0 * xGrad1 * yGrad2 + xGrad0 * yGrad1 * yGrad2;
                    double cyyy = yGrad0 * yGrad1 * yGrad2;
                    double prod = mean0 * mean1 * mean2;
                    reciprocal = fabs(prod) > 0.0 ? 1.0 / prod : 0.0;
                    xB[k] = (c0 * G(0) + cx * G(2) + cy * G(3) + cxx * G(5) + cxy * G(6) + cyy * G(7) + cxxx * G(9) +
                             cxxy * G(10) + cxyy * G(11) + cyyy * G(12)) *
                            reciprocal;
                    yB[k] = (c0 * G(1) + cx * G(3) + cy * G(4) + cxx * G(6) + cxy * G(7) + cyy * G(8) + cxxx * G(10) +
                             cxxy * G(11) + cxyy * G(12) + cyyy * G(13)) *
                            reciprocal;
                }
            }
    }
#undef G
}

/* find_departure_points :5255-5349; returns the number of vertices over the CFL warning length */
int ira_departure_points(int nVertices, double dt, const double *u, const double *v, const double *minLen, double *dp)
{
    int warn = 0;
    for (int i = 0; i <= nVertices; i++) dp[2 * i] = dp[2 * i + 1] = 0.0;
    for (int i = 0; i < nVertices; i++) {
        dp[2 * i] = -u[i] * dt;
        dp[2 * i + 1] = -v[i] * dt;
    }
    for (int i = 0; i < nVertices; i++)
        if (sqrt(dp[2 * i] * dp[2 * i] + dp[2 * i + 1] * dp[2 * i + 1]) > minLen[i]) warn++;
    return warn;
}

/* shift_vertices_of_departure_triangle :6270-6500; returns 1 if the basis vectors are parallel */
static int shift_vertices(int iEdge, int iCell, int M, const int *nEdgesOnCell, int onSphere, const double *xVOE,
                          const double *yVOE, const double *xVOC, const double *yVOC, double ev1[2], double ev2[2],
                          double xt[3], double yt[3], int tvoe, int tvoc, double *area)
{
    int err = 0;
    if (onSphere) {
        double cp = cross_product_2d(ev1, ev2), e1c[2], e2c[2], denom;
        int n = nEdgesOnCell[iCell - 1], im1, ip1;
        if (fabs(cp) < eps11) err = 1;
        if (cp > 0.0) {
            double t0 = ev1[0], t1 = ev1[1];
            ev1[0] = ev2[0];
            ev1[1] = ev2[1];
            ev2[0] = t0;
            ev2[1] = t1;
        }
        im1 = tvoc - 1;
        if (im1 < 1) im1 = im1 + n;
        ip1 = tvoc + 1;
        if (ip1 > n) ip1 = ip1 - n;
        e1c[0] = xVOC[I2(im1, iCell, M)] - xVOC[I2(tvoc, iCell, M)];
        e1c[1] = yVOC[I2(im1, iCell, M)] - yVOC[I2(tvoc, iCell, M)];
        e2c[0] = xVOC[I2(ip1, iCell, M)] - xVOC[I2(tvoc, iCell, M)];
        e2c[1] = yVOC[I2(ip1, iCell, M)] - yVOC[I2(tvoc, iCell, M)];
        denom = ev1[0] * ev2[1] - ev2[0] * ev1[1];
        for (int k = 0; k < 3; k++) {
            double a, b;
            xt[k] = xt[k] - xVOE[I2(tvoe, iEdge, NVER)];
            yt[k] = yt[k] - yVOE[I2(tvoe, iEdge, NVER)];
            a = (xt[k] * ev2[1] - yt[k] * ev2[0]) / denom;
            b = (yt[k] * ev1[0] - xt[k] * ev1[1]) / denom;
            xt[k] = xVOC[I2(tvoc, iCell, M)] + a * e1c[0] + b * e2c[0];
            yt[k] = yVOC[I2(tvoc, iCell, M)] + a * e1c[1] + b * e2c[1];
        }
    } else {
        for (int k = 0; k < 3; k++) {
            xt[k] = xt[k] - xVOE[I2(tvoe, iEdge, NVER)] + xVOC[I2(tvoc, iCell, M)];
            yt[k] = yt[k] - yVOE[I2(tvoe, iEdge, NVER)] + yVOC[I2(tvoc, iCell, M)];
        }
    }
    *area = fabs(0.5 * ((xt[1] - xt[0]) * (yt[2] - yt[0]) - (yt[1] - yt[0]) * (xt[2] - xt[0])));
    return err;
}

#define XT(q, t, e) xTri[(((size_t)(e) - 1) * NTRI + ((t) - 1)) * NQP + ((q) - 1)]
#define YT(q, t, e) yTri[(((size_t)(e) - 1) * NTRI + ((t) - 1)) * NQP + ((q) - 1)]

/* find_departure_triangles :5365-6254 (vertexDegree 3, or 4 with the split of a side triangle at E5 / E6,
 * :5807-5958).  Returns the number of parallel-basis errors. */
int ira_departure_triangles(int nEdges, int nCells, int M, const int *nEdgesOnCell, const double *xVOE,
                            const double *yVOE, const double *xVOC, const double *yVOC, const int *verticesOnCell,
                            const int *verticesOnEdge, const int *remapEdge, const int *cellsOnEdgeRemap,
                            const int *edgesOnEdgeRemap, int vertexDegree, const double *dpIn, int onSphere,
                            int *maskEdge, double *xTri,
                            double *yTri, int *iCellTri, double *triArea)
{
    int errs = 0;
    int tvoe[NTRI] = {0}, tvoc[NTRI] = {0}, fluxSign[NTRI] = {0};
    double ev1[NTRI][2], ev2[NTRI][2];
    memset(ev1, 0, sizeof ev1);
    memset(ev2, 0, sizeof ev2);
    memset(xTri, 0, sizeof(double) * NQP * NTRI * ((size_t)nEdges + 1));
    memset(yTri, 0, sizeof(double) * NQP * NTRI * ((size_t)nEdges + 1));
    memset(triArea, 0, sizeof(double) * NTRI * ((size_t)nEdges + 1));
    memset(iCellTri, 0, sizeof(int) * NTRI * ((size_t)nEdges + 1));
    for (int e = 0; e <= nEdges; e++) maskEdge[e] = 0;
    for (int e = 1; e <= nEdges; e++)
        if (remapEdge[e - 1] == 1)
            for (int k = 1; k <= 2; k++) {
                int v = verticesOnEdge[I2(k, e, 2)];
                double len2 = dpIn[2 * (v - 1)] * dpIn[2 * (v - 1)] + dpIn[2 * (v - 1) + 1] * dpIn[2 * (v - 1) + 1];
                if (len2 > 0.0) maskEdge[e - 1] = 1;
            }

    for (int e = 1; e <= nEdges; e++) {
        double dp[2][2], ev[2][2], ip[2], ipMain[2];
        int tc = 0, hit;
        if (maskEdge[e - 1] != 1) continue;
        for (int k = 1; k <= 2; k++) {
            int v = verticesOnEdge[I2(k, e, 2)];
            dp[k - 1][0] = xVOE[I2(k, e, NVER)] + dpIn[2 * (v - 1)];
            dp[k - 1][1] = yVOE[I2(k, e, NVER)] + dpIn[2 * (v - 1) + 1];
            ev[k - 1][0] = xVOE[I2(k, e, NVER)];
            ev[k - 1][1] = yVOE[I2(k, e, NVER)];
        }
        /* side triangles, :5711-5968 */
        for (int k = 1; k <= 2; k++) {
            double en1[2] = {ev[k - 1][0], ev[k - 1][1]}, en2[2];
            for (int side = 0; side <= 1; side++) {
                int ieoe = k + 2 * side, ivoer = ieoe + 2;
                int en = edgesOnEdgeRemap[I2(ieoe, e, NEER)];
                if (en >= 1 && en <= nEdges) {
                    en2[0] = xVOE[I2(ivoer, e, NVER)];
                    en2[1] = yVOE[I2(ivoer, e, NVER)];
                    hit = find_line_intersection(dp[0], dp[1], en1, en2, ip);
                } else
                    hit = 0;
                if (hit) {
                    int c, iOtherEdge, iOtherVertex;
                    tc = tc + 1;
                    XT(1, tc, e) = ev[k - 1][0];
                    YT(1, tc, e) = ev[k - 1][1];
                    XT(2, tc, e) = dp[k - 1][0];
                    YT(2, tc, e) = dp[k - 1][1];
                    XT(3, tc, e) = ip[0];
                    YT(3, tc, e) = ip[1];
                    tvoe[tc - 1] = k;
                    c = cellsOnEdgeRemap[I2(k + 2, e, NEER)];
                    iCellTri[I2(tc, e, NTRI)] = c;
                    for (int iv = 1; iv <= nEdgesOnCell[c - 1]; iv++)
                        if (verticesOnEdge[I2(k, e, 2)] == verticesOnCell[I2(iv, c, M)]) tvoc[tc - 1] = iv;
                    if (onSphere) {
                        ev1[tc - 1][0] = en2[0] - en1[0];
                        ev1[tc - 1][1] = en2[1] - en1[1];
                        if (vertexDegree == 3) {
                            iOtherEdge = ieoe + 2;
                            if (iOtherEdge > 4) iOtherEdge = iOtherEdge - 4;
                        } else
                            iOtherEdge = k + 4; /* E5 for V1, E6 for V2 */
                        iOtherVertex = iOtherEdge + 2;
                        en2[0] = xVOE[I2(iOtherVertex, e, NVER)];
                        en2[1] = yVOE[I2(iOtherVertex, e, NVER)];
                        ev2[tc - 1][0] = en2[0] - en1[0];
                        ev2[tc - 1][1] = en2[1] - en1[1];
                    }
                    fluxSign[tc - 1] = side == 0 ? 1 : -1;
                    if (vertexDegree == 4) { /* :5807-5958 */
                        int hitMain, e56 = edgesOnEdgeRemap[I2(k + 4, e, NEER)];
                        double ip0[2];
                        if (e56 >= 1 && e56 <= nEdges) {
                            en2[0] = xVOE[I2(k + 6, e, NVER)]; /* V7 or V8 */
                            en2[1] = yVOE[I2(k + 6, e, NVER)];
                            hitMain = find_line_intersection(dp[0], dp[1], en1, en2, ip0);
                        } else
                            hitMain = 0;
                        if (hitMain) {
                            /* the triangle just made ends at the line of the main edge instead */
                            XT(3, tc, e) = ip0[0];
                            YT(3, tc, e) = ip0[1];
                            if (side == 0) { /* IP in the left half-plane: this triangle lies in the right one */
                                c = cellsOnEdgeRemap[I2(k + 4, e, NEER)]; /* C5 or C6 */
                                iCellTri[I2(tc, e, NTRI)] = c;
                                for (int iv = 1; iv <= nEdgesOnCell[c - 1]; iv++)
                                    if (verticesOnEdge[I2(k, e, 2)] == verticesOnCell[I2(iv, c, M)]) tvoc[tc - 1] = iv;
                                if (onSphere) {
                                    iOtherVertex = ivoer + 2; /* V3 -> V5, V4 -> V6 */
                                    ev1[tc - 1][0] = xVOE[I2(iOtherVertex, e, NVER)] - en1[0];
                                    ev1[tc - 1][1] = yVOE[I2(iOtherVertex, e, NVER)] - en1[1];
                                }
                            } else if (onSphere) {
                                iOtherVertex = ivoer - 2; /* V5 -> V3, V6 -> V4 */
                                ev1[tc - 1][0] = xVOE[I2(iOtherVertex, e, NVER)] - en1[0];
                                ev1[tc - 1][1] = yVOE[I2(iOtherVertex, e, NVER)] - en1[1];
                            }
                            /* second triangle: V1 / V2, IP0, IP on the side where IP lies */
                            tc = tc + 1;
                            XT(1, tc, e) = ev[k - 1][0];
                            YT(1, tc, e) = ev[k - 1][1];
                            XT(2, tc, e) = ip0[0];
                            YT(2, tc, e) = ip0[1];
                            XT(3, tc, e) = ip[0];
                            YT(3, tc, e) = ip[1];
                            c = side == 0 ? cellsOnEdgeRemap[I2(k + 2, e, NEER)] : cellsOnEdgeRemap[I2(k + 4, e, NEER)];
                            iCellTri[I2(tc, e, NTRI)] = c;
                            tvoe[tc - 1] = tvoe[tc - 2];
                            for (int iv = 1; iv <= nEdgesOnCell[c - 1]; iv++)
                                if (verticesOnEdge[I2(k, e, 2)] == verticesOnCell[I2(iv, c, M)]) tvoc[tc - 1] = iv;
                            if (onSphere) {
                                ev1[tc - 1][0] = xVOE[I2(ivoer, e, NVER)] - en1[0];
                                ev1[tc - 1][1] = yVOE[I2(ivoer, e, NVER)] - en1[1];
                                ev2[tc - 1][0] = ev2[tc - 2][0];
                                ev2[tc - 1][1] = ev2[tc - 2][1];
                            }
                            fluxSign[tc - 1] = fluxSign[tc - 2];
                        } else if (side == 1) { /* no crossing of E5 / E6; IP and triangle in the right half-plane */
                            c = cellsOnEdgeRemap[I2(k + 4, e, NEER)];
                            iCellTri[I2(tc, e, NTRI)] = c;
                            for (int iv = 1; iv <= nEdgesOnCell[c - 1]; iv++)
                                if (verticesOnEdge[I2(k, e, 2)] == verticesOnCell[I2(iv, c, M)]) tvoc[tc - 1] = iv;
                        }
                    }
                    dp[k - 1][0] = ip[0];
                    dp[k - 1][1] = ip[1];
                }
            }
        }
        /* central triangles, :5970-6183 */
        hit = find_line_intersection(dp[0], dp[1], ev[0], ev[1], ipMain);
        if (hit) {
            for (int k = 1; k <= 2; k++) {
                int c, inHP, iOtherEdge, o = k == 1 ? 2 : 1;
                tc = tc + 1;
                XT(1, tc, e) = ev[k - 1][0];
                YT(1, tc, e) = ev[k - 1][1];
                XT(2, tc, e) = dp[k - 1][0];
                YT(2, tc, e) = dp[k - 1][1];
                XT(3, tc, e) = ipMain[0];
                YT(3, tc, e) = ipMain[1];
                tvoe[tc - 1] = k;
                inHP = point_in_half_plane(ev[0], ev[1], dp[k - 1]);
                if (inHP) {
                    iCellTri[I2(tc, e, NTRI)] = cellsOnEdgeRemap[I2(1, e, NEER)];
                    fluxSign[tc - 1] = 1;
                } else {
                    iCellTri[I2(tc, e, NTRI)] = cellsOnEdgeRemap[I2(2, e, NEER)];
                    fluxSign[tc - 1] = -1;
                }
                c = iCellTri[I2(tc, e, NTRI)];
                for (int iv = 1; iv <= nEdgesOnCell[c - 1]; iv++)
                    if (verticesOnEdge[I2(k, e, 2)] == verticesOnCell[I2(iv, c, M)]) tvoc[tc - 1] = iv;
                if (onSphere) {
                    ev1[tc - 1][0] = ev[o - 1][0] - ev[k - 1][0];
                    ev1[tc - 1][1] = ev[o - 1][1] - ev[k - 1][1];
                    iOtherEdge = inHP ? k : k + 2;
                    ev2[tc - 1][0] = xVOE[I2(iOtherEdge + 2, e, NVER)] - ev[k - 1][0];
                    ev2[tc - 1][1] = yVOE[I2(iOtherEdge + 2, e, NVER)] - ev[k - 1][1];
                }
            }
        } else {
            double quadArea = quadrilateral_area(ev[0], ev[1], dp[1], dp[0]);
            if (quadArea > 0.0) {
                for (int k = 1; k <= 2; k++) {
                    int c, inHP, iOtherEdge, o = k == 1 ? 2 : 1;
                    tc = tc + 1;
                    if (k == 1) { /* V1, V2, D1 */
                        XT(1, tc, e) = ev[0][0];  YT(1, tc, e) = ev[0][1];
                        XT(2, tc, e) = ev[1][0];  YT(2, tc, e) = ev[1][1];
                        XT(3, tc, e) = dp[0][0];  YT(3, tc, e) = dp[0][1];
                    } else { /* V2, D1, D2 */
                        XT(1, tc, e) = ev[1][0];  YT(1, tc, e) = ev[1][1];
                        XT(2, tc, e) = dp[0][0];  YT(2, tc, e) = dp[0][1];
                        XT(3, tc, e) = dp[1][0];  YT(3, tc, e) = dp[1][1];
                    }
                    tvoe[tc - 1] = k;
                    inHP = point_in_half_plane(ev[0], ev[1], dp[k - 1]);
                    if (inHP) {
                        iCellTri[I2(tc, e, NTRI)] = cellsOnEdgeRemap[I2(1, e, NEER)];
                        fluxSign[tc - 1] = 1;
                    } else {
                        iCellTri[I2(tc, e, NTRI)] = cellsOnEdgeRemap[I2(2, e, NEER)];
                        fluxSign[tc - 1] = -1;
                    }
                    c = iCellTri[I2(tc, e, NTRI)];
                    for (int iv = 1; iv <= nEdgesOnCell[c - 1]; iv++)
                        if (verticesOnEdge[I2(k, e, 2)] == verticesOnCell[I2(iv, c, M)]) tvoc[tc - 1] = iv;
                    if (onSphere) {
                        ev1[tc - 1][0] = ev[o - 1][0] - ev[k - 1][0];
                        ev1[tc - 1][1] = ev[o - 1][1] - ev[k - 1][1];
                        iOtherEdge = inHP ? k : k + 2;
                        ev2[tc - 1][0] = xVOE[I2(iOtherEdge + 2, e, NVER)] - ev[k - 1][0];
                        ev2[tc - 1][1] = yVOE[I2(iOtherEdge + 2, e, NVER)] - ev[k - 1][1];
                    }
                }
            }
        }
        /* edge frame -> source-cell frame, signed area, :6189-6249 */
        for (int t = 1; t <= tc; t++) {
            int c = iCellTri[I2(t, e, NTRI)];
            if (c >= 1 && c <= nCells) {
                double xt[3] = {XT(1, t, e), XT(2, t, e), XT(3, t, e)}, yt[3] = {YT(1, t, e), YT(2, t, e), YT(3, t, e)};
                double area;
                errs += shift_vertices(e, c, M, nEdgesOnCell, onSphere, xVOE, yVOE, xVOC, yVOC, ev1[t - 1], ev2[t - 1],
                                       xt, yt, tvoe[t - 1], tvoc[t - 1], &area);
                for (int k = 0; k < 3; k++) {
                    XT(k + 1, t, e) = xt[k];
                    YT(k + 1, t, e) = yt[k];
                }
                triArea[I2(t, e, NTRI)] = area * fluxSign[t - 1];
            }
        }
    }
    return errs;
}

/* get_triangle_quadrature_points :6546-6652 (nQuadPoints = 6) */
void ira_quadrature_points(int nEdges, const int *maskEdge, double *xTri, double *yTri)
{
    for (int e = 1; e <= nEdges; e++)
        if (maskEdge[e - 1] == 1)
            for (int t = 1; t <= NTRI; t++) {
                double x1 = XT(1, t, e), y1 = YT(1, t, e), x2 = XT(2, t, e), y2 = YT(2, t, e), x3 = XT(3, t, e),
                       y3 = YT(3, t, e);
                XT(1, t, e) = q1QP * x1 + q1QP * x2 + q2QP * x3;  YT(1, t, e) = q1QP * y1 + q1QP * y2 + q2QP * y3;
                XT(2, t, e) = q1QP * x1 + q2QP * x2 + q1QP * x3;  YT(2, t, e) = q1QP * y1 + q2QP * y2 + q1QP * y3;
                XT(3, t, e) = q2QP * x1 + q1QP * x2 + q1QP * x3;  YT(3, t, e) = q2QP * y1 + q1QP * y2 + q1QP * y3;
                XT(4, t, e) = q3QP * x1 + q4QP * x2 + q4QP * x3;  YT(4, t, e) = q3QP * y1 + q4QP * y2 + q4QP * y3;
                XT(5, t, e) = q4QP * x1 + q3QP * x2 + q4QP * x3;  YT(5, t, e) = q4QP * y1 + q3QP * y2 + q4QP * y3;
                XT(6, t, e) = q4QP * x1 + q4QP * x2 + q3QP * x3;  YT(6, t, e) = q4QP * y1 + q4QP * y2 + q3QP * y3;
            }
}

/* integrate_fluxes_over_triangles :6667-6967 for one tracer.  triVal(L, nCat, NQP, NTRI, nEdges+1)
 * is this tracer's value of mass*tracer... at the quadrature points; parentTriVal has Lp layers or is
 * NULL for the mass-like field (parent value 1).  Returns the number of negative reconstructed
 * values when checkNegative is set (the reference aborts on the first, :6903-6951). */
int ira_integrate_fluxes(int nEdges, int L, int nCat, int Lp, const int *maskEdge, const double *xTri,
                         const double *yTri, const double *triArea, const int *iCellTri, const double *center,
                         const double *xGrad, const double *yGrad, const double *parentTriVal, double *triVal,
                         double *edgeFlux, int checkNegative)
{
    static const double wq[NQP] = {1.09951743655321885e-01, 1.09951743655321885e-01, 1.09951743655321885e-01,
                                   2.23381589678011389e-01, 2.23381589678011389e-01, 2.23381589678011389e-01};
    size_t np = (size_t)L * nCat, npp = (size_t)Lp * nCat;
    int neg = 0;
    memset(edgeFlux, 0, sizeof(double) * np * ((size_t)nEdges + 1));
#pragma omp parallel for reduction(+ : neg)
    for (int e = 1; e <= nEdges; e++) {
        if (maskEdge[e - 1] != 1) continue;
        for (int t = 1; t <= NTRI; t++) {
            double area = triArea[I2(t, e, NTRI)];
            int c;
            if (area == 0.0) continue;
            c = iCellTri[I2(t, e, NTRI)];
            for (int cat = 1; cat <= nCat; cat++)
                for (int l = 1; l <= L; l++) {
                    size_t k = F3(l, cat, c, L, nCat);
                    double integral = 0.0;
                    for (int q = 1; q <= NQP; q++) {
                        size_t kt = ((((size_t)e - 1) * NTRI + (t - 1)) * NQP + (q - 1)) * np + ((size_t)(cat - 1) * L + (l - 1));
                        double pv = 1.0, val;
                        if (parentTriVal) {
                            size_t kp = ((((size_t)e - 1) * NTRI + (t - 1)) * NQP + (q - 1)) * npp +
                                        ((size_t)(cat - 1) * Lp + (Lp == 1 ? 0 : l - 1));
                            pv = parentTriVal[kp];
                        }
                        val = pv * (center[k] + xGrad[k] * XT(q, t, e) + yGrad[k] * YT(q, t, e));
                        triVal[kt] = val;
                        integral = integral + wq[q - 1] * val;
                        if (checkNegative && val < 0.0) neg++;
                    }
                    edgeFlux[((size_t)e - 1) * np + ((size_t)(cat - 1) * L + (l - 1))] += area * integral;
                }
        }
    }
    return neg;
}

/* compute_mass_tracer_products :6982-7109 for one tracer (parentProd NULL: value 1) */
void ira_mass_tracer_product(int nCells, int L, int nCat, int Lp, const double *parentProd, const double *array,
                             double *prod)
{
    for (int c = 1; c <= nCells; c++)
        for (int cat = 1; cat <= nCat; cat++)
            for (int l = 1; l <= L; l++) {
                double pp = parentProd ? parentProd[P3(l, cat, c, Lp, nCat)] : 1.0;
                prod[F3(l, cat, c, L, nCat)] = pp * array[F3(l, cat, c, L, nCat)];
            }
}

/* update_mass_and_tracers :7125-7523 for one tracer.  For the mass-like field (parentProd NULL) also
 * applies the negative-value rule :7465-7507 and returns the number of values below -puny^2
 * (the reference aborts on the first). */
int ira_update_tracer(int nCellsSolve, int M, int L, int nCat, int Lp, const int *nEdgesOnCell, const int *edgesOnCell,
                      const int *cellsOnEdge, const double *areaCell, const double *edgeFlux, const double *parentProd,
                      double *prod, double *array)
{
    const double puny = 1.0e-11;
    size_t np = (size_t)L * nCat;
    int bad = 0;
#pragma omp parallel for
    for (int c = 1; c <= nCellsSolve; c++) {
        double flux[512];
        for (size_t p = 0; p < np; p++) flux[p] = 0.0;
        for (int ie = 1; ie <= nEdgesOnCell[c - 1]; ie++) {
            int e = edgesOnCell[I2(ie, c, M)];
            int sg = (c == cellsOnEdge[I2(1, e, 2)]) ? 1 : -1;
            for (size_t p = 0; p < np; p++) flux[p] = flux[p] + edgeFlux[((size_t)e - 1) * np + p] * sg;
        }
        for (int cat = 1; cat <= nCat; cat++)
            for (int l = 1; l <= L; l++) {
                size_t k = F3(l, cat, c, L, nCat);
                double pp = parentProd ? parentProd[P3(l, cat, c, Lp, nCat)] : 1.0;
                if (pp > 0.0)
                    array[k] = (prod[k] - (flux[(size_t)(cat - 1) * L + (l - 1)] / areaCell[c - 1])) / pp;
                else
                    array[k] = 0.0;
                prod[k] = pp * array[k];
            }
    }
    if (!parentProd)
        for (int c = 1; c <= nCellsSolve; c++)
            for (size_t p = 0; p < np; p++) {
                size_t k = ((size_t)c - 1) * np + p;
                if (array[k] < -puny * puny)
                    bad++;
                else if (array[k] >= -puny * puny && array[k] < 0.0)
                    array[k] = 0.0;
            }
    return bad;
}

/* zap_small_mass :8764-8886: first the mass-like field (fills maskCategoryCell), then each tracer */
void ira_zap_small_mass(int nCellsSolve, int nCat, double *mass, int *maskCategoryCell, int nColsAlloc)
{
    const double smallMassThreshold = 1.0e-22;
    memset(maskCategoryCell, 0, sizeof(int) * (size_t)nCat * nColsAlloc);
    for (int c = 1; c <= nCellsSolve; c++)
        for (int cat = 1; cat <= nCat; cat++) {
            size_t k = F3(1, cat, c, 1, nCat);
            if (mass[k] > 0.0 && mass[k] < smallMassThreshold) {
                mass[k] = 0.0;
                maskCategoryCell[k] = 1;
            }
        }
}

void ira_zap_tracer(int nCellsSolve, int L, int nCat, const int *maskCategoryCell, double *array)
{
    for (int c = 1; c <= nCellsSolve; c++)
        for (int cat = 1; cat <= nCat; cat++)
            if (maskCategoryCell[F3(1, cat, c, 1, nCat)] == 1)
                for (int l = 1; l <= L; l++) array[F3(l, cat, c, L, nCat)] = 0.0;
}

/* recover_tracer_means :7541-7982 for one tracer (config_recover_tracer_means_check): integrates the product
 * mass * tracer ... of the reconstructions over each cell, one sub-triangle (centre, vertex i, vertex i+1) per
 * edge with the 6-point rule, weighted by 0.25 dcEdge dvEdge / sum.  nLev levels, the mass-like field first and
 * this tracer last; level k has Lk[k] layers.  recovered(L, nCat, nCells+1) receives the integrals; returns the
 * number of entries for which the reference logs "Reconstructed tracer does not recover the mean". */
int ira_recover_tracer_means(int nCells, int M, int L, int nCat, const int *nEdgesOnCell, const int *edgesOnCell,
                             const double *xVOC, const double *yVOC, const double *dcEdge, const double *dvEdge,
                             int nLev, const int *Lk, const double *const *cen, const double *const *gx,
                             const double *const *gy, const double *prod, double *recovered)
{
    int warnings = 0;
    for (int c = 1; c <= nCells; c++) {
        int n = nEdgesOnCell[c - 1];
        double frac[64], sumArea = 0.0;
        for (int ie = 1; ie <= n; ie++) {
            int e = edgesOnCell[I2(ie, c, M)];
            frac[ie - 1] = 0.25 * dcEdge[e - 1] * dvEdge[e - 1];
            sumArea = sumArea + frac[ie - 1];
        }
        for (int cat = 1; cat <= nCat; cat++)
            for (int l = 1; l <= L; l++) recovered[F3(l, cat, c, L, nCat)] = 0.0;
        for (int ie = 1; ie <= n; ie++) {
            double fr = frac[ie - 1] / sumArea;
            int ip = ie + 1 > n ? 1 : ie + 1;
            double x1 = 0.0, y1 = 0.0;
            double x2 = xVOC[I2(ie, c, M)], y2 = yVOC[I2(ie, c, M)];
            double x3 = xVOC[I2(ip, c, M)], y3 = yVOC[I2(ip, c, M)];
            double xq[NQP], yq[NQP], wq[NQP];
            xq[0] = q1QP * x1 + q1QP * x2 + q2QP * x3;  yq[0] = q1QP * y1 + q1QP * y2 + q2QP * y3;
            xq[1] = q1QP * x1 + q2QP * x2 + q1QP * x3;  yq[1] = q1QP * y1 + q2QP * y2 + q1QP * y3;
            xq[2] = q2QP * x1 + q1QP * x2 + q1QP * x3;  yq[2] = q2QP * y1 + q1QP * y2 + q1QP * y3;
            xq[3] = q3QP * x1 + q4QP * x2 + q4QP * x3;  yq[3] = q3QP * y1 + q4QP * y2 + q4QP * y3;
            xq[4] = q4QP * x1 + q3QP * x2 + q4QP * x3;  yq[4] = q4QP * y1 + q3QP * y2 + q4QP * y3;
            xq[5] = q4QP * x1 + q4QP * x2 + q3QP * x3;  yq[5] = q4QP * y1 + q4QP * y2 + q3QP * y3;
            for (int q = 0; q < NQP; q++) wq[q] = q < 3 ? w1QP : w2QP;
            for (int cat = 1; cat <= nCat; cat++)
                for (int l = 1; l <= L; l++) {
                    double tri = 0.0;
                    for (int q = 0; q < NQP; q++) {
                        double v = 1.0; /* the dummy parent of the mass-like field :7619-7626 */
                        for (int k = 0; k < nLev; k++) {
                            size_t j = P3(l, cat, c, Lk[k], nCat);
                            v = (cen[k][j] + gx[k][j] * xq[q] + gy[k][j] * yq[q]) * v;
                        }
                        tri = tri + wq[q] * v;
                    }
                    recovered[F3(l, cat, c, L, nCat)] = recovered[F3(l, cat, c, L, nCat)] + fr * tri;
                }
        }
        for (int cat = 1; cat <= nCat; cat++)
            for (int l = 1; l <= L; l++) {
                double mean = prod[F3(l, cat, c, L, nCat)];
                double difference = recovered[F3(l, cat, c, L, nCat)] - mean;
                if (fabs(difference) > eps11 * fabs(mean) && fabs(difference) > eps11) warnings++;
            }
    }
    return warnings;
}

/* sum_tracers :7998-8111 for one tracer: sums(L, nCat) over the owned cells */
void ira_sum_tracer(int nCellsSolve, int L, int nCat, const double *areaCell, const double *prod, double *sums)
{
    size_t np = (size_t)L * nCat;
    for (size_t p = 0; p < np; p++) sums[p] = 0.0;
    for (int c = 1; c <= nCellsSolve; c++)
        for (size_t p = 0; p < np; p++) sums[p] = sums[p] + areaCell[c - 1] * prod[((size_t)c - 1) * np + p];
}

/* tracer_local_min_max :8268-8392 for one tracer: min and max of the old values over a cell and its edge
 * neighbours, counting only entries whose mask (made with threshold 0) is 1; zero elsewhere. */
void ira_local_min_max(int nCells, int nCellsSolve, int M, int L, int nCat, const int *nEdgesOnCell,
                       const int *cellsOnCell, const double *array, const int *mask, double *localMin,
                       double *localMax)
{
    memset(localMin, 0, sizeof(double) * (size_t)L * nCat * ((size_t)nCells + 1));
    memset(localMax, 0, sizeof(double) * (size_t)L * nCat * ((size_t)nCells + 1));
    for (int c = 1; c <= nCellsSolve; c++)
        for (int cat = 1; cat <= nCat; cat++)
            for (int l = 1; l <= L; l++) {
                size_t k = F3(l, cat, c, L, nCat);
                if (mask[k] == 1) {
                    localMin[k] = array[k];
                    localMax[k] = array[k];
                }
                for (int ie = 1; ie <= nEdgesOnCell[c - 1]; ie++) {
                    int nb = cellsOnCell[I2(ie, c, M)];
                    if (nb >= 1 && nb <= nCells) {
                        size_t kn = F3(l, cat, nb, L, nCat);
                        if (mask[kn] == 1) {
                            if (array[kn] < localMin[k]) localMin[k] = array[kn];
                            if (array[kn] > localMax[k]) localMax[k] = array[kn];
                        }
                    }
                }
            }
}

/* check_tracer_monotonicity :8416-8749 for one tracer with a parent, after the halo update of localMin /
 * localMax: extend both by one layer of neighbours (in place, cells in ascending order, so a neighbour with
 * a lower index contributes its already extended value), then compare the new values.  `mask` is the one
 * made from the old values with threshold 1e-11.  Returns the number of violations; the reference stops at
 * the first one. */
int ira_check_monotonicity(int nCells, int nCellsSolve, int M, int L, int nCat, const int *nEdgesOnCell,
                           const int *cellsOnCell, const double *array, const int *mask, double *localMin,
                           double *localMax)
{
    for (int c = 1; c <= nCellsSolve; c++)
        for (int cat = 1; cat <= nCat; cat++)
            for (int l = 1; l <= L; l++) {
                size_t k = F3(l, cat, c, L, nCat);
                if (mask[k] != 1) continue;
                for (int ie = 1; ie <= nEdgesOnCell[c - 1]; ie++) {
                    int nb = cellsOnCell[I2(ie, c, M)];
                    if (nb >= 1 && nb <= nCells) {
                        size_t kn = F3(l, cat, nb, L, nCat);
                        if (mask[kn] == 1) {
                            if (localMin[kn] < localMin[k]) localMin[k] = localMin[kn];
                            if (localMax[kn] > localMax[k]) localMax[k] = localMax[kn];
                        }
                    }
                }
            }
    int bad = 0;
    for (int c = 1; c <= nCellsSolve; c++)
        for (int cat = 1; cat <= nCat; cat++)
            for (int l = 1; l <= L; l++) {
                size_t k = F3(l, cat, c, L, nCat);
                if (mask[k] != 1) continue;
                double tolMin = eps11 * fmax(1.0, fabs(localMin[k]));
                double tolMax = eps11 * fmax(1.0, fabs(localMax[k]));
                if (array[k] < localMin[k] - tolMin)
                    bad++;
                else if (array[k] > localMax[k] + tolMax)
                    bad++;
            }
    return bad;
}

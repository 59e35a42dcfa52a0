// Incremental-remap (IR) tracer advection on the device.
//
// Follows src/shared/mpas_seaice_advection_incremental_remap.F of the reference (routine : line cited
// at each kernel) for hexagonal (vertexDegree = 3) and quadrilateral (vertexDegree = 4) meshes.  Arithmetic is written
// statement by statement
// in the reference's evaluation order and compiled with -ffp-contract=off.
//
// Data layout: structure-of-arrays like the rest of the library.  A tracer with L layers and nCat
// categories is L*nCat planes of padC doubles, plane index = category * L + layer (the reference's
// (layer, category, cell) array with the cell index made fastest).  Per-edge quantities are planes of
// padE values.  Cells, vertices and edges use the library's internal numbering ("none" = nCells /
// nVertices / nEdges, a zero-filled slot).
//
// Differences of organisation (not of arithmetic) from the reference:
//  * tracer masks (make_masks :3404) are not stored: mask(tracer, cell) is recomputed where it is used as
//    "parent value > 1e-11" (1 for the mass-like field);
//  * the values of mass*tracer products at the quadrature points (triangleValue, :6826) are not
//    stored either: a child recomputes the product along its chain of ancestors in the same order of
//    multiplications, which gives the same bits;
//  * the quadrature points are recomputed from the stored triangle vertices inside the flux kernel;
//  * a departure triangle is transformed to its source cell's frame (shift_vertices :6270) as soon as it
//    is found; nothing it depends on changes in between in the reference.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "sih_internal.h"

#define NTRI 6          // nTriPerEdgeRemap, Registry.xml:63
#define EPS11 1.0e-11   // :60
#define W1QP 1.09951743655321885e-01  // :63-70
#define W2QP 2.23381589678011389e-01
#define Q1QP 9.15762135097710761e-02
#define Q2QP 8.16847572980458514e-01
#define Q3QP 1.08103018168070275e-01
#define Q4QP 4.45948490915965612e-01
#define HUGE_R 1.79769313486231571e308

#define FLAG_CCW 1
#define FLAG_HALFPLANE 2
#define FLAG_AREA 4
#define FLAG_PARALLEL 8
#define FLAG_NEGAREA 16
#define FLAG_NEGMASS 32
#define FLAG_MONO 64

struct IrDev {
    int nCells, nCellsSolve, nVertices, nEdges, M, nCat, padC, padV, padE, onSphere, rotate;
    int vdeg, nRemap;  // vertexDegree; side edges and cells around an edge: 4 on hexagonal meshes, 6 on quadrilateral ones
    const int *nEdgesOnCell, *voc, *coc;
    int *eoc, *coe, *voe, *eov, *remapEdge, *eoer, *coer;
    double *g2c, *xVOC, *yVOC, *xVOE, *yVOE, *geom, *minLen;
    const double *dcEdge, *dvEdge, *coeffs, *areaCell;
    const double *u, *v;
    double *dp;
    int *triCount, *triCell;
    double *xTri, *yTri, *triArea;
    int *maskCell, *flag;
    int *triSlot;       // per (remap cell 0..nRemap-1, edge vertex 0..1, edge): where that vertex sits in that cell (emitTriangle)
    const int *updWin;  // per block of 256 owned cells: first edge of the window of fluxes its update stages in LDS
};

// device-resident tracer table (one entry per tracer, in list order: parents before children)
// The reconstruction of a tracer (value at the cell's origin and the limited gradient) is stored as records:
// for category cat and cell c the L layers follow each other, three doubles each (centre, x and y gradient).
// The flux kernel gathers them by source cell; one cell's layers then sit in one or two cache lines instead
// of 3 L different ones.
#define IR_LCHUNK 8  // layers handled together (flux threads, staging of the records)
#define IR_REC(T, cat, pc, c, layer) ((((size_t)(cat) * (pc) + (c)) * (T).S + (layer)) * 3)
// The table's pointers are read from memory by the kernels, so the compiler cannot tell that they point to global
// memory and would use flat loads and stores for everything behind them; a flat access also counts as an LDS access,
// and a kernel that waits for its LDS traffic (all three large kernels stage through LDS) then waits for every
// outstanding load behind these pointers as well, look-ahead included.  Declared as global for the device pass.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) double IrG;
#else
typedef double IrG;
#endif
struct IrTabEntry {
    IrG *val, *rec, *xB, *yB, *prod, *flux;
    int L, parent, nLev, hasChild;
    int chain[4];  // tracer indices from the mass-like field (chain[0]) to this tracer (chain[nLev-1])
    int S;         // records per (category, cell) row of `rec` (= L: one row per tracer; a row shared by all tracers was slower)
};

struct IrTracer {
    std::string name;
    int parent = -1, L = 1, nParents = 0;
    bool hasChild = false;
    double *val = nullptr, *rec = nullptr, *xB = nullptr, *yB = nullptr, *prod = nullptr, *flux = nullptr;
    double *center = nullptr, *gx = nullptr, *gy = nullptr;  // plane copies for inspection (sih_ir_unpack_reconstruction)
    double *recovered = nullptr;  // mass*tracer product integrated from the reconstruction (recover-means check)
    std::vector<double> sumInit, sumFinal;
    // monotonicity check: min / max of the old values in the neighbourhood, whether the old parent value
    // exceeded 1e-11 (allocated on first use)
    double *lmin = nullptr, *lmax = nullptr;
    int *mask11 = nullptr;
};

struct sih_ir_state {
    IrDev d{};
    std::vector<IrTracer> tracers;
    int nCat = 0;
    bool inited = false;
    IrTabEntry *d_tab = nullptr;
    struct IrPlane *d_planes = nullptr;
    int nPlanes = 0;
    // 2: one thread per (edge, category) walking the tracer tree (needs walkOk); 1: one thread per (edge, plane);
    // 0: one thread per (edge, category), every tracer from scratch
    int fluxMode = 2;
    int *d_updWin = nullptr;  // per block of 256 owned cells: where its window of edge fluxes begins (k_ir_update)
    struct IrStep *d_steps = nullptr;
    int nSteps = 0, nSlots = 0;
    bool walkOk = false;
    double *d_sums = nullptr;
    int iVol = -1, iSnow = -1;
    int cflWarnings = 0;
    int recoverWarnings = 0;  // entries whose reconstruction did not recover the mean (last run with that check)
    bool monoReady = false, monoPending = false;
    double *monoTmp[4] = {nullptr, nullptr, nullptr, nullptr};  // ping-pong buffers of the extension sweeps
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

void sih_ir_destroy(sih_ctx *c)
{
    if (!c->ir) return;
    if (c->ir->d_tab) (void)hipFree(c->ir->d_tab);
    if (c->ir->d_planes) (void)hipFree(c->ir->d_planes);
    if (c->ir->d_steps) (void)hipFree(c->ir->d_steps);
    if (c->ir->d_updWin) (void)hipFree(c->ir->d_updWin);
    if (c->ir->d_sums) (void)hipFree(c->ir->d_sums);
    for (hipEvent_t e : c->ir->ev)
        if (e) (void)hipEventDestroy(e);
    delete c->ir;
    c->ir = nullptr;
}

bool sih_ir_tracer_table(sih_ctx *c, std::vector<SihTracerInfo> &out, int *nCategories)
{
    out.clear();
    if (!c->ir || c->ir->tracers.empty()) return false;
    for (const IrTracer &t : c->ir->tracers) out.push_back({t.name, t.parent, t.L, t.val});
    if (nCategories) *nCategories = c->ir->nCat;
    return true;
}

// Blocks are dealt round-robin over the 8 XCDs, each with its own L2, so blocks b and b + 8 share one.
// A launch over (tile, part) pairs whose `nPart` parts re-read the tile's geometry puts all parts of a tile
// on one XCD, back to back: linear block id -> (tile, part).  Affinity only; any placement is correct.
// The grid must hold roundUp(nTile, 8) * nPart blocks; returns false for the padding blocks.
__device__ inline bool xcdTile(unsigned nTile, unsigned nPart, unsigned &tile, unsigned &part)
{
    const unsigned L = blockIdx.x, xcd = L % 8u, j = L / 8u;
#ifdef SIH_XCD_STRIDED
    tile = xcd + 8u * (j / nPart);
#else
    // each XCD walks a contiguous range of tiles: neighbouring tiles, which gather from the same cache lines
    // along their common border, meet in the same L2 one after the other
    tile = xcd * ((nTile + 7u) / 8u) + j / nPart;
#endif
    part = j % nPart;
    return tile < nTile;
}
static inline unsigned xcdGrid(unsigned nTile, unsigned nPart) { return ((nTile + 7u) / 8u) * 8u * nPart; }

// ------------------------------------------------------------------------------------------ helpers
__device__ inline void unitVector3(double v[3])
{  // :8900-8918
    double mag = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (mag > 0.0) {
        v[0] = v[0] / mag;
        v[1] = v[1] / mag;
        v[2] = v[2] / mag;
    } else {
        v[0] = v[1] = v[2] = 0.0;
    }
}

__device__ inline double cross2(double ax, double ay, double bx, double by) { return ax * by - ay * bx; }  // :9083

// point_in_half_plane :9201-9234 with its dummy arguments (point, lineStart, lineEnd)
__device__ inline bool inHalfPlane(double px, double py, double sx, double sy, double ex, double ey)
{
    return cross2(ex - sx, ey - sy, px - sx, py - sy) >= 0.0;
}

__device__ inline double triArea2(double ax, double ay, double bx, double by, double cx, double cy)
{  // :9172-9185
    return fabs(0.5 * ((bx - ax) * (cy - ay) - (by - ay) * (cx - ax)));
}

// find_line_intersection :8934-9016
__device__ inline bool lineIntersect(double x1, double y1, double x2, double y2, double x3, double y3, double x4,
                                     double y4, double &ix, double &iy)
{
    double rx = x2 - x1, ry = y2 - y1, sx = x4 - x3, sy = y4 - y3;
    double rsCross = rx * sy - ry * sx;
    double rsCrossMin = EPS11 * sqrt((rx * rx + ry * ry) * (sx * sx + sy * sy));
    bool hit = false;
    if (fabs(rsCross) > rsCrossMin) {
        double t1 = (sy * (x3 - x1) - sx * (y3 - y1)) / rsCross;
        double t2 = (ry * (x3 - x1) - rx * (y3 - y1)) / rsCross;
        ix = x1 + t1 * rx;
        iy = y1 + t1 * ry;
        if (t1 > 0.0 && t1 < 1.0 && t2 > 0.0 && t2 < 1.0) hit = true;
    } else {
        ix = HUGE_R;
        iy = HUGE_R;
    }
    return hit;
}

// rows 1 and 2 of transGlobalToLocal, define_local_to_global_transformations :990-1084
__device__ inline void localFrame(double x, double y, double z, double r1[3], double r2[3])
{
    double up[3] = {x, y, z};
    unitVector3(up);
    if (fabs(x * x + y * y) > EPS11) {
        r1[0] = -y;
        r1[1] = x;
        r1[2] = 0.0;
        unitVector3(r1);
        r2[0] = up[1] * r1[2] - up[2] * r1[1];  // cross_product_3d(vertical, east) :9109-9123
        r2[1] = up[2] * r1[0] - up[0] * r1[2];
        r2[2] = up[0] * r1[1] - up[1] * r1[0];
    } else if (z > 0.0) {
        r1[0] = 1.0; r1[1] = 0.0; r1[2] = 0.0;
        r2[0] = 0.0; r2[1] = 1.0; r2[2] = 0.0;
    } else {
        r1[0] = 0.0; r1[1] = 1.0; r1[2] = 0.0;
        r2[0] = 1.0; r2[1] = 0.0; r2[2] = 0.0;
    }
}

__device__ inline void rotated(const double *x, const double *y, const double *z, int i, int rot, double p[3])
{  // rotate_global_vectors :948-975
    if (rot) {
        p[0] = -z[i];
        p[1] = y[i];
        p[2] = x[i];
    } else {
        p[0] = x[i];
        p[1] = y[i];
        p[2] = z[i];
    }
}

// ------------------------------------------------------------------------------------------ init kernels
struct IrInitArgs {
    const double *xC, *yC, *zC, *xV, *yV, *zV, *xE, *yE, *zE;
};

// per cell: local frame, get_vertex_on_cell_coordinates :1823-2036 (with its two orientation checks)
__global__ void k_ir_init_cells(IrDev d, IrInitArgs a)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    const int rot = d.rotate && d.onSphere;
    double pc[3], r1[3] = {0, 0, 0}, r2[3] = {0, 0, 0};
    rotated(a.xC, a.yC, a.zC, c, rot, pc);
    if (d.onSphere) {
        localFrame(pc[0], pc[1], pc[2], r1, r2);
        for (int j = 0; j < 3; j++) {
            d.g2c[(size_t)j * d.padC + c] = r1[j];
            d.g2c[(size_t)(3 + j) * d.padC + c] = r2[j];
        }
    }
    const int n = d.nEdgesOnCell[c];
    for (int i = 0; i < n; i++) {
        int v = d.voc[(size_t)i * d.padC + c];
        double pv[3];
        rotated(a.xV, a.yV, a.zV, v, rot, pv);
        double w0 = pv[0] - pc[0], w1 = pv[1] - pc[1], w2 = pv[2] - pc[2];
        if (d.onSphere) {
            d.xVOC[(size_t)i * d.padC + c] = r1[0] * w0 + r1[1] * w1 + r1[2] * w2;
            d.yVOC[(size_t)i * d.padC + c] = r2[0] * w0 + r2[1] * w1 + r2[2] * w2;
        } else {
            d.xVOC[(size_t)i * d.padC + c] = w0;
            d.yVOC[(size_t)i * d.padC + c] = w1;
        }
    }
    int bad = 0;
    for (int i = 0; i < n; i++) {
        int ip1 = i + 1 >= n ? 0 : i + 1;
        if (d.onSphere) {
            double p1[3], p2[3];
            rotated(a.xV, a.yV, a.zV, d.voc[(size_t)i * d.padC + c], rot, p1);
            rotated(a.xV, a.yV, a.zV, d.voc[(size_t)ip1 * d.padC + c], rot, p2);
            double a0 = p1[0] - pc[0], a1 = p1[1] - pc[1], a2 = p1[2] - pc[2];
            double b0 = p2[0] - pc[0], b1 = p2[1] - pc[1], b2 = p2[2] - pc[2];
            double w0 = a1 * b2 - a2 * b1, w1 = a2 * b0 - a0 * b2, w2 = a0 * b1 - a1 * b0;
            if (!(w0 * pc[0] + w1 * pc[1] + w2 * pc[2] >= 0.0)) bad = 1;
        }
        if (!(cross2(d.xVOC[(size_t)i * d.padC + c], d.yVOC[(size_t)i * d.padC + c], d.xVOC[(size_t)ip1 * d.padC + c],
                     d.yVOC[(size_t)ip1 * d.padC + c]) >= 0.0))
            bad = 1;
    }
    if (bad) atomicOr(d.flag, FLAG_CCW);
    if (c < d.nCellsSolve && !(d.areaCell[c] > 0.0)) atomicOr(d.flag, FLAG_AREA);  // :789-797
}

// per edge: get_geometry_incremental_remap :1255-1583 and the first loop of :1604-1635
__global__ void k_ir_init_edges(IrDev d, IrInitArgs a)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d.nEdges) return;
    const int rot = d.rotate && d.onSphere;
    const int nC = d.nCells, nE = d.nEdges;
    int c1 = d.coe[e], c2 = d.coe[(size_t)d.padE + e];
    int remap = (c1 < d.nCellsSolve || c2 < d.nCellsSolve) ? 1 : 0;
    if (c1 >= nC || c2 >= nC) remap = 0;
    d.remapEdge[e] = remap;
    int v1 = d.voe[e], v2 = d.voe[(size_t)d.padE + e];
    double pe[3], p1[3], p2[3], r1[3] = {0, 0, 0}, r2[3] = {0, 0, 0};
    rotated(a.xE, a.yE, a.zE, e, rot, pe);
    rotated(a.xV, a.yV, a.zV, v1, rot, p1);
    rotated(a.xV, a.yV, a.zV, v2, rot, p2);
    if (d.onSphere) {
        localFrame(pe[0], pe[1], pe[2], r1, r2);
        double w0 = p2[0] - p1[0], w1 = p2[1] - p1[1], w2 = p2[2] - p1[2];
        double xv = r1[0] * w0 + r1[1] * w1 + r1[2] * w2;
        double yv = r2[0] * w0 + r2[1] * w1 + r2[2] * w2;
        d.xVOE[e] = -0.5 * xv;
        d.yVOE[e] = -0.5 * yv;
        d.xVOE[(size_t)d.padE + e] = 0.5 * xv;
        d.yVOE[(size_t)d.padE + e] = 0.5 * yv;
    } else if (remap) {
        d.xVOE[e] = p1[0] - pe[0];
        d.yVOE[e] = p1[1] - pe[1];
        d.xVOE[(size_t)d.padE + e] = p2[0] - pe[0];
        d.yVOE[(size_t)d.padE + e] = p2[1] - pe[1];
    }
    for (int k = 0; k < d.nRemap; k++) {
        d.eoer[(size_t)k * d.padE + e] = nE;
        d.coer[(size_t)k * d.padE + e] = nC;
    }
    if (!remap) return;
    {  // cell 1 must lie in the left half-plane of V1 -> V2, :1296-1395
        double ev[2][2], cc[2], pc[3];
        const double *pp[2] = {p1, p2};
        for (int k = 0; k < 2; k++) {
            double w0 = pp[k][0] - pe[0], w1 = pp[k][1] - pe[1], w2 = pp[k][2] - pe[2];
            if (d.onSphere) {
                ev[k][0] = r1[0] * w0 + r1[1] * w1 + r1[2] * w2;
                ev[k][1] = r2[0] * w0 + r2[1] * w1 + r2[2] * w2;
            } else {
                ev[k][0] = w0;
                ev[k][1] = w1;
            }
        }
        rotated(a.xC, a.yC, a.zC, c1, rot, pc);
        double w0 = pc[0] - pe[0], w1 = pc[1] - pe[1], w2 = pc[2] - pe[2];
        if (d.onSphere) {
            cc[0] = r1[0] * w0 + r1[1] * w1 + r1[2] * w2;
            cc[1] = r2[0] * w0 + r2[1] * w1 + r2[2] * w2;
        } else {
            cc[0] = w0;
            cc[1] = w1;
        }
        if (!inHalfPlane(ev[0][0], ev[0][1], ev[1][0], ev[1][1], cc[0], cc[1])) atomicOr(d.flag, FLAG_HALFPLANE);
    }
    d.coer[e] = c1;
    d.coer[(size_t)d.padE + e] = c2;
    int eo[6] = {nE, nE, nE, nE, nE, nE};
    for (int k = 0; k < 2; k++) {
        int c = k == 0 ? c1 : c2;
        int n = d.nEdgesOnCell[c], iMain = 0;
        for (int i = 0; i < n; i++)
            if (d.eoc[(size_t)i * d.padC + c] == e) {
                iMain = i;
                break;
            }
        int im = iMain - 1 < 0 ? iMain - 1 + n : iMain - 1;
        int ip = iMain + 1 >= n ? iMain + 1 - n : iMain + 1;
        if (k == 0) {
            eo[0] = d.eoc[(size_t)im * d.padC + c];
            eo[1] = d.eoc[(size_t)ip * d.padC + c];
        } else {
            eo[2] = d.eoc[(size_t)ip * d.padC + c];
            eo[3] = d.eoc[(size_t)im * d.padC + c];
        }
    }
    if (d.vdeg == 4) {  // E5 and E6: the edge at V1 / V2 that is neither the main edge nor one of E1..E4, :1479-1507
        for (int k = 0; k < 2; k++) {
            const int v = k == 0 ? v1 : v2;
            for (int j = 0; j < 4; j++) {
                const int en = d.eov[(size_t)j * d.padV + v];
                if (en >= nE) continue;
                bool isNew = en != e;
                for (int q = 0; q < 4; q++)
                    if (en == eo[q]) isNew = false;
                if (isNew) {
                    eo[4 + k] = en;
                    break;
                }
            }
        }
    }
    for (int k = 0; k < d.nRemap; k++) d.eoer[(size_t)k * d.padE + e] = eo[k];
    if (d.vdeg == 3) {
        for (int k = 0; k < 2; k++) {  // C3 and C4, :1513-1539
            int en = eo[k];
            if (en >= nE) en = eo[k + 2];
            int found = nC;
            for (int j = 0; j < 2; j++) {
                int cn = d.coe[(size_t)j * d.padE + en];
                if (cn < nC && cn != c1 && cn != c2) found = cn;
            }
            d.coer[(size_t)(k + 2) * d.padE + e] = found;
        }
    } else {
        for (int k = 0; k < 2; k++) {  // C3, C4 across E1, E2 from C1; C5, C6 across E3, E4 from C2, :1541-1578
            int found = nC;
            if (eo[k] < nE)
                for (int j = 0; j < 2; j++) {
                    int cn = d.coe[(size_t)j * d.padE + eo[k]];
                    if (cn < nC && cn != c1) found = cn;
                }
            d.coer[(size_t)(k + 2) * d.padE + e] = found;
            found = nC;
            if (eo[k + 2] < nE)
                for (int j = 0; j < 2; j++) {
                    int cn = d.coe[(size_t)j * d.padE + eo[k + 2]];
                    if (cn < nC && cn != c2) found = cn;
                }
            d.coer[(size_t)(k + 4) * d.padE + e] = found;
        }
    }
}

// per remap edge: coordinates of the distant vertices V3..V6 (V3..V8 on quadrilateral meshes), second loop of :1650-1699 / planar :1703-1760
__global__ void k_ir_init_edge_vertices(IrDev d, IrInitArgs a)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d.nEdges || d.remapEdge[e] != 1) return;
    const int nE = d.nEdges;
    int count = 2;
    for (int k = 0; k < d.nRemap; k++) {
        int en = d.eoer[(size_t)k * d.padE + e];
        if (en >= nE) continue;
        int n1 = 0, m1 = 0, m2 = 0;
        for (int n = 0; n < 2; n++)
            for (int m = 0; m < 2; m++)
                if (d.voe[(size_t)m * d.padE + en] == d.voe[(size_t)n * d.padE + e]) {
                    n1 = n;
                    m1 = m;
                    m2 = 1 - m;
                    count = count + 1;
                    break;
                }
        if (d.onSphere) {
            d.xVOE[(size_t)(2 + k) * d.padE + e] =
                d.xVOE[(size_t)n1 * d.padE + e] + (d.xVOE[(size_t)m2 * d.padE + en] - d.xVOE[(size_t)m1 * d.padE + en]);
            d.yVOE[(size_t)(2 + k) * d.padE + e] =
                d.yVOE[(size_t)n1 * d.padE + e] + (d.yVOE[(size_t)m2 * d.padE + en] - d.yVOE[(size_t)m1 * d.padE + en]);
        } else {
            int v = d.voe[(size_t)m2 * d.padE + en];
            d.xVOE[(size_t)(count - 1) * d.padE + e] = a.xV[v] - a.xE[e];
            d.yVOE[(size_t)(count - 1) * d.padE + e] = a.yV[v] - a.yE[e];
        }
    }
}

// per vertex: minLengthEdgesOnVertex :1764-1806
__global__ void k_ir_init_vertices(IrDev d, IrInitArgs a)
{
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= d.nVertices) return;
    double mn = HUGE_R;
    for (int k = 0; k < d.vdeg; k++) {
        int e = d.eov[(size_t)k * d.padV + v];
        if (e < d.nEdges) {
            int v1 = d.voe[e], v2 = d.voe[(size_t)d.padE + e];
            // the rotation of the frame permutes the components, so the squared length is summed in the
            // rotated order like the reference does
            double p1[3], p2[3];
            rotated(a.xV, a.yV, a.zV, v1, d.rotate && d.onSphere, p1);
            rotated(a.xV, a.yV, a.zV, v2, d.rotate && d.onSphere, p2);
            double w0 = p2[0] - p1[0], w1 = p2[1] - p1[1], w2 = p2[2] - p1[2];
            double len = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
            if (len < mn) mn = len;
        }
    }
    d.minLen[v] = mn;
}

// per cell: compute_geometric_cell_averages :2097-2323
__global__ void k_ir_init_geom(IrDev d)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    const int n = d.nEdgesOnCell[c];
    double sumArea = 0.0;
    for (int i = 0; i < n; i++) {
        int e = d.eoc[(size_t)i * d.padC + c];
        double fr = 0.25 * d.dcEdge[e] * d.dvEdge[e];
        sumArea = sumArea + fr;
    }
    double acc[14];
    for (int k = 0; k < 14; k++) acc[k] = 0.0;
    for (int i = 0; i < n; i++) {
        int e = d.eoc[(size_t)i * d.padC + c];
        double frac = (0.25 * d.dcEdge[e] * d.dvEdge[e]) / sumArea;
        int ip1 = i + 1 >= n ? 0 : i + 1;
        const double x1 = 0.0, y1 = 0.0;
        double x2 = d.xVOC[(size_t)i * d.padC + c], y2 = d.yVOC[(size_t)i * d.padC + c];
        double x3 = d.xVOC[(size_t)ip1 * d.padC + c], y3 = d.yVOC[(size_t)ip1 * d.padC + c];
        double xq[6], yq[6];
        xq[0] = Q1QP * x1 + Q1QP * x2 + Q2QP * x3;  yq[0] = Q1QP * y1 + Q1QP * y2 + Q2QP * y3;
        xq[1] = Q1QP * x1 + Q2QP * x2 + Q1QP * x3;  yq[1] = Q1QP * y1 + Q2QP * y2 + Q1QP * y3;
        xq[2] = Q2QP * x1 + Q1QP * x2 + Q1QP * x3;  yq[2] = Q2QP * y1 + Q1QP * y2 + Q1QP * y3;
        xq[3] = Q3QP * x1 + Q4QP * x2 + Q4QP * x3;  yq[3] = Q3QP * y1 + Q4QP * y2 + Q4QP * y3;
        xq[4] = Q4QP * x1 + Q3QP * x2 + Q4QP * x3;  yq[4] = Q4QP * y1 + Q3QP * y2 + Q4QP * y3;
        xq[5] = Q4QP * x1 + Q4QP * x2 + Q3QP * x3;  yq[5] = Q4QP * y1 + Q4QP * y2 + Q3QP * y3;
        double av[14];
        for (int k = 0; k < 14; k++) av[k] = 0.0;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            double x = xq[q], y = yq[q], w = q < 3 ? W1QP : W2QP;
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
        for (int k = 0; k < 14; k++) acc[k] = acc[k] + frac * av[k];
    }
    for (int k = 0; k < 14; k++) d.geom[(size_t)k * d.padC + c] = acc[k];
}

// ------------------------------------------------------------------------------------------ step kernels

// volume_to_thickness :9248-9281 (toThickness) / thickness_to_volume :9295-9324, every block cell
__global__ void k_ir_convert(IrDev d, const double *area, double *vol, int toThickness)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    size_t k = (size_t)blockIdx.y * d.padC + c;
    double a = area[k];
    if (toThickness)
        vol[k] = a > 0.0 ? vol[k] / a : 0.0;
    else
        vol[k] = a * vol[k];
}

// maskCell of make_masks :3486-3492
__global__ void k_ir_mask_cell(IrDev d, const double *mass)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    double s = 0.0;
    for (int cat = 0; cat < d.nCat; cat++) s = s + mass[(size_t)cat * d.padC + c];
    d.maskCell[c] = s > 0.0 ? 1 : 0;
}

// ---- monotonicity check (optional diagnostic) ----------------------------------------------------
// tracer_local_min_max :8268-8392 with the masks of make_masks(threshold 0) :2999-3021, and the mask of
// make_masks(threshold 1e-11) :3023-3029 kept for the check after the update.  Thread (cell, plane).
__global__ void k_ir_mono_init(IrDev d, const double *__restrict__ val, const double *__restrict__ parentVal, int L,
                               int Lp, double *__restrict__ lmin, double *__restrict__ lmax, int *__restrict__ mask11)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    const int p = blockIdx.y, cat = p / L, layer = p % L;
    const size_t pc = d.padC, k = (size_t)p * pc + c;
    const size_t kp = (size_t)(Lp == 1 ? cat : cat * Lp + layer) * pc;
    const double pv = parentVal[kp + c];
    mask11[k] = pv > EPS11 ? 1 : 0;
    double lo = 0.0, hi = 0.0;
    if (c < d.nCellsSolve) {
        if (pv > 0.0) lo = hi = val[k];
        const int n = d.nEdgesOnCell[c];
        for (int ie = 0; ie < n; ie++) {
            const int nb = d.coc[(size_t)ie * pc + c];
            if (nb < d.nCells && parentVal[kp + nb] > 0.0) {
                const double x = val[(size_t)p * pc + nb];
                if (x < lo) lo = x;
                if (x > hi) hi = x;
            }
        }
    }
    lmin[k] = lo;
    lmax[k] = hi;
}

// One sweep of the extension by a layer of neighbours (check_tracer_monotonicity :8506-8535).  The
// reference extends in place with cells in ascending (host) order, so a neighbour with a lower index
// contributes its extended value, one with a higher index (or a halo cell) its original one.  Sweeps are
// repeated until nothing changes, which reproduces that order exactly.
__global__ void k_ir_mono_extend(IrDev d, const int *__restrict__ hostIndex, const int *__restrict__ mask11,
                                 const double *__restrict__ lmin0, const double *__restrict__ lmax0,
                                 const double *__restrict__ inMin, const double *__restrict__ inMax,
                                 double *__restrict__ outMin, double *__restrict__ outMax, int *changed)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    const size_t pc = d.padC, base = (size_t)blockIdx.y * pc, k = base + c;
    double lo = lmin0[k], hi = lmax0[k];
    if (c < d.nCellsSolve && mask11[k] == 1) {
        const int hc = hostIndex[c], n = d.nEdgesOnCell[c];
        for (int ie = 0; ie < n; ie++) {
            const int nb = d.coc[(size_t)ie * pc + c];
            if (nb < d.nCells && mask11[base + nb] == 1) {
                const bool earlier = nb < d.nCellsSolve && hostIndex[nb] < hc;
                const double a = earlier ? inMin[base + nb] : lmin0[base + nb];
                const double b = earlier ? inMax[base + nb] : lmax0[base + nb];
                if (a < lo) lo = a;
                if (b > hi) hi = b;
            }
        }
    }
    if (lo != inMin[k] || hi != inMax[k]) *changed = 1;
    outMin[k] = lo;
    outMax[k] = hi;
}

// the comparison itself :8552-8601 / :8682-8736
__global__ void k_ir_mono_check(IrDev d, const double *__restrict__ val, const int *__restrict__ mask11,
                                const double *__restrict__ emin, const double *__restrict__ emax)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCellsSolve) return;
    const size_t k = (size_t)blockIdx.y * d.padC + c;
    if (mask11[k] != 1) return;
    const double tolMin = EPS11 * fmax(1.0, fabs(emin[k])), tolMax = EPS11 * fmax(1.0, fabs(emax[k]));
    if (val[k] < emin[k] - tolMin || val[k] > emax[k] + tolMax) atomicOr(d.flag, FLAG_MONO);
}

// All tracers of one (cell, category): compute_gradient :4204/:4429, limit_tracer_gradient :4802/:4999,
// centre value :3785-3791, compute_barycenter_coordinates :4658-4787 (tracers with children) and the
// mass*tracer products (compute_mass_tracer_products :6982-7109).  Everything a child needs from its
// ancestors (barycentre, centre value, gradient, product) belongs to the same cell and category, so one
// thread walks the tracer list in order; the cell's geometry stays in registers for all of them.
template <int M>
__global__ void __launch_bounds__(256) k_ir_reconstruct(IrDev d, const IrTabEntry *__restrict__ tab, int nT)
{
    // the categories of one block of cells share its geometry and neighbour lists: one XCD for all of them
    unsigned cb, ucat;
    if (!xcdTile((d.nCells + 255u) / 256u, (unsigned)d.nCat, cb, ucat)) return;
    // The records of a block's cells are contiguous in memory but a thread's own share is 24 L bytes apart
    // from its neighbour's: they go through LDS and leave as one coalesced copy per chunk of layers.  All
    // threads of the block therefore stay to the end; the ones past the last cell repeat the last cell's
    // work (same values to the same addresses) and their staging rows are not copied.
    __shared__ double stage[256 * 3 * IR_LCHUNK];
    const int c0 = cb * blockDim.x;
    const int nValid = min((int)blockDim.x, d.nCells - c0);
    const int c = min(c0 + (int)threadIdx.x, d.nCells - 1);
    const int cat = (int)ucat;
    const size_t pc = d.padC;
    const bool ice = d.maskCell[c] == 1;
    const int n = d.nEdgesOnCell[c];
    int nb[M];
    double rdc[M], cf0[M], cf1[M], cf2[M], xv[M], yv[M];
    unsigned plus = 0, valid = 0;  // per edge slot: the cell is the edge's first cell; the slot has a neighbour cell
    double t11 = 0, t12 = 0, t13 = 0, t21 = 0, t22 = 0, t23 = 0;
    if (ice) {
        // (in passes: everything that depends on the cell alone is requested in the first, what hangs on an edge number
        // in the second -- in one pass over the edge slots every slot waited for its own chain of loads)
        int eds[M];
#pragma unroll
        for (int ie = 0; ie < M; ie++) {
            nb[ie] = d.nCells;
            eds[ie] = 0;
            rdc[ie] = 1.0;
            cf0[ie] = cf1[ie] = cf2[ie] = xv[ie] = yv[ie] = 0.0;
            if (ie < n) {
                nb[ie] = d.coc[(size_t)ie * pc + c];
                eds[ie] = d.eoc[(size_t)ie * pc + c];
                cf0[ie] = d.coeffs[(size_t)(ie * 3 + 0) * pc + c];
                cf1[ie] = d.coeffs[(size_t)(ie * 3 + 1) * pc + c];
                cf2[ie] = d.coeffs[(size_t)(ie * 3 + 2) * pc + c];
                xv[ie] = d.xVOC[(size_t)ie * pc + c];
                yv[ie] = d.yVOC[(size_t)ie * pc + c];
            }
        }
        int first[M];
#pragma unroll
        for (int ie = 0; ie < M; ie++) {
            first[ie] = -1;
            if (ie < n) {
                rdc[ie] = d.dcEdge[eds[ie]];
                first[ie] = d.coe[eds[ie]];
            }
        }
#pragma unroll
        for (int ie = 0; ie < M; ie++) {
            if (ie < n) {
                if (nb[ie] < d.nCells) valid |= 1u << ie;
                if (c == first[ie]) plus |= 1u << ie;
            }
        }
        if (d.onSphere) {
            t11 = d.g2c[c];
            t12 = d.g2c[pc + c];
            t13 = d.g2c[2 * pc + c];
            t21 = d.g2c[3 * pc + c];
            t22 = d.g2c[4 * pc + c];
            t23 = d.g2c[5 * pc + c];
        }
    }
    const double xc0 = d.geom[c], yc0 = d.geom[pc + c];
    for (int ti = 0; ti < nT; ti++) {
        const IrTabEntry T = tab[ti];
        const bool hasParent = T.parent >= 0;
        const IrTabEntry *P = hasParent ? &tab[T.parent] : nullptr;
        // masks of make_masks (:3523-3553): the parent value decides; a parent without layers gives the same
        // answer for every layer of the child, so it is looked up once per tracer
        const bool sharedMask = hasParent && P->L == 1;
        bool mcS = true;
        unsigned mnbS = 0xffffffffu;
        double pprodS = 1.0, xbS = 0.0, ybS = 0.0;  // likewise the parent's product and barycentre
        if (sharedMask) {
            pprodS = P->prod[(size_t)cat * pc + c];
            xbS = P->xB[(size_t)cat * pc + c];
            ybS = P->yB[(size_t)cat * pc + c];
        }
        if (ice && sharedMask) {
            // (the parent's values at the cell and its neighbours: all seven requested before the first is looked at;
            // with look-up and test in one pass per neighbour every pass waited for its own gather.  A slot without a
            // neighbour reads the cell itself and is left out by its `valid` bit.)
            const size_t lp1 = (size_t)cat * pc;
            const double pvc = P->val[lp1 + c];
            double pvn[M];
#pragma unroll
            for (int ie = 0; ie < M; ie++) pvn[ie] = P->val[lp1 + (((valid >> ie) & 1u) ? nb[ie] : c)];
            mcS = pvc > EPS11;
            mnbS = 0;
#pragma unroll
            for (int ie = 0; ie < M; ie++)
                if (((valid >> ie) & 1u) && pvn[ie] > EPS11) mnbS |= 1u << ie;
        }
        // the values of a layer (the cell's and its neighbours') are requested one layer ahead, so that their way
        // through memory overlaps the arithmetic of the layer before (2 waves a SIMD hide nothing of it: 0.70 ->
        // 0.52 ms at QU60; asking for a tracer's first layer during the tracer before it gained nothing more)
        double fNext = 0.0, fnbNext[M];
        {
            const size_t k0 = (size_t)(cat * T.L) * pc + c;
            fNext = T.val[k0];
#pragma unroll
            for (int ie = 0; ie < M; ie++) fnbNext[ie] = ice ? T.val[k0 - c + (((valid >> ie) & 1u) ? nb[ie] : c)] : 0.0;
        }
        for (int l0 = 0; l0 < T.L; l0 += IR_LCHUNK) {
        const int nlc = min(IR_LCHUNK, T.L - l0);
        for (int layer = l0; layer < l0 + nlc; layer++) {
            const size_t k = (size_t)(cat * T.L + layer) * pc + c;
            const size_t lp = hasParent ? (size_t)(P->L == 1 ? cat : cat * T.L + layer) * pc : 0;
            const double f = fNext;
            double fnbCur[M];
#pragma unroll
            for (int ie = 0; ie < M; ie++) fnbCur[ie] = fnbNext[ie];
            if (layer + 1 < T.L) {  // the tracers' values are inputs here: nothing writes them
                const size_t k1 = k + pc;
                fNext = T.val[k1];
#pragma unroll
                for (int ie = 0; ie < M; ie++) fnbNext[ie] = ice ? T.val[k1 - c + (((valid >> ie) & 1u) ? nb[ie] : c)] : 0.0;
            }
            T.prod[k] = (hasParent ? (sharedMask ? pprodS : P->prod[lp + c]) : 1.0) * f;
            double xb, yb;
            if (sharedMask) {
                xb = xbS;
                yb = ybS;
            } else if (hasParent) {
                xb = P->xB[lp + c];
                yb = P->yB[lp + c];
            } else {
                xb = xc0;
                yb = yc0;
            }
            double gxv = 0.0, gyv = 0.0;
            if (ice) {
                const bool mc = !hasParent ? true : (sharedMask ? mcS : (P->val[lp + c] > EPS11));
                double g0 = 0.0, g1 = 0.0, g2 = 0.0, mx = f, mn = f;
                // No branches per neighbour: a slot without a neighbour (past the cell's last edge, or a boundary)
                // reads the cell itself, counts as masked out and adds +0 to sums that cannot be -0.
#pragma unroll
                for (int ie = 0; ie < M; ie++) {
                    const bool has = ((valid >> ie) & 1u) != 0;
                    const int nbi = has ? nb[ie] : c;
                    const double fnb = fnbCur[ie];
                    bool mnb = has;
                    if (hasParent) mnb = mnb && (sharedMask ? ((mnbS >> ie) & 1u) != 0 : (P->val[lp + nbi] > EPS11));
                    // sg * (fnb - f) with sg = +-1 is the difference taken one way or the other
                    const double diff = (plus >> ie) & 1u ? fnb - f : -(fnb - f);
                    const double ng = (mc && mnb) ? diff / rdc[ie] : 0.0;
                    mx = mnb ? fmax(mx, fnb) : mx;
                    mn = mnb ? fmin(mn, fnb) : mn;
                    g0 = g0 + cf0[ie] * ng;
                    g1 = g1 + cf1[ie] * ng;
                    g2 = g2 + cf2[ie] * ng;
                }
                if (d.rotate && d.onSphere) {  // :4353-4359
                    double t = g0;
                    g0 = -g2;
                    g2 = t;
                }
                if (d.onSphere) {
                    gxv = t11 * g0 + t12 * g1 + t13 * g2;
                    gyv = t21 * g0 + t22 * g1 + t23 * g2;
                } else {
                    gxv = g0;
                    gyv = g1;
                }
                mx = mx - f;
                mn = mn - f;
                double lmx = 0.0, lmn = 0.0;
#pragma unroll
                for (int iv = 0; iv < M; iv++) {
                    if (iv < n) {
                        double dev = gxv * (xv[iv] - xb) + gyv * (yv[iv] - yb);
                        lmx = fmax(lmx, dev);
                        lmn = fmin(lmn, dev);
                    }
                }
                double f1 = fabs(lmx) > fabs(mx) ? fmax(0.0, mx / lmx) : 1.0;
                double f2 = fabs(lmn) > fabs(mn) ? fmax(0.0, mn / lmn) : 1.0;
                double fac = fmin(f1, f2);
                fac = fmax(0.0, fac - EPS11);
                gxv = gxv * fac;
                gyv = gyv * fac;
            }
            const double cen = f - gxv * xb - gyv * yb;
            {
                double *r = stage + ((size_t)threadIdx.x * nlc + (layer - l0)) * 3;
                r[0] = cen;
                r[1] = gxv;
                r[2] = gyv;
            }
            if (T.hasChild) {
                double bx = 0.0, by = 0.0;
                if (ice) {
#define G(j) d.geom[(size_t)(j) * pc + c]
                    double mean[3], ce[3], gx[3], gy[3];
#pragma unroll
                    for (int j = 0; j < 3; j++) {  // constant trip count keeps the arrays in registers
                        mean[j] = f;
                        ce[j] = cen;
                        gx[j] = gxv;
                        gy[j] = gyv;
                        if (j < T.nLev - 1) {
                            const IrTabEntry &A = tab[T.chain[j]];
                            size_t kj = (size_t)(A.L == 1 ? cat : cat * A.L + layer) * pc + c;
                            const IrG *r = A.rec + IR_REC(A, cat, pc, c, A.L == 1 ? 0 : layer);
                            mean[j] = A.val[kj];
                            ce[j] = r[0];
                            gx[j] = r[1];
                            gy[j] = r[2];
                        }
                    }
                    if (T.nLev == 1) {
                        double c0 = ce[0], cx = gx[0], cy = gy[0];
                        double rcp = fabs(mean[0]) > 0.0 ? 1.0 / mean[0] : 0.0;
                        bx = (c0 * G(0) + cx * G(2) + cy * G(3)) * rcp;
                        by = (c0 * G(1) + cx * G(3) + cy * G(4)) * rcp;
                    } else if (T.nLev == 2) {
                        double c0 = ce[0] * ce[1];
                        double cx = ce[0] * gx[1] + gx[0] * ce[1];
                        double cy = ce[0] * gy[1] + gy[0] * ce[1];
                        double cxx = gx[0] * gx[1];
                        double cxy = gx[0] * gy[1] + gy[0] * gx[1];
                        double cyy = gy[0] * gy[1];
                        double pr = mean[0] * mean[1];
                        double rcp = fabs(pr) > 0.0 ? 1.0 / pr : 0.0;
                        bx = (c0 * G(0) + cx * G(2) + cy * G(3) + cxx * G(5) + cxy * G(6) + cyy * G(7)) * rcp;
                        by = (c0 * G(1) + cx * G(3) + cy * G(4) + cxx * G(6) + cxy * G(7) + cyy * G(8)) * rcp;
                    } else {
                        double c0 = ce[0] * ce[1] * ce[2];
                        double cx = ce[0] * ce[1] * gx[2] + ce[0] * gx[1] * ce[2] + gx[0] * ce[1] * ce[2];
                        double cy = ce[0] * ce[1] * gy[2] + ce[0] * gy[1] * ce[2] + gy[0] * ce[1] * ce[2];
                        double cxx = ce[0] * gx[1] * gx[2] + gx[0] * ce[1] * gx[2] + gx[0] * gx[1] * ce[2];
                        double cxy = ce[0] * gx[1] * gy[2] + gx[0] * gy[1] * ce[2] + gy[0] * ce[1] * gx[2] +
                                     ce[0] * gy[1] * gx[2] + gx[0] * ce[1] * gy[2] + gy[0] * gx[1] * ce[2];
                        double cyy = ce[0] * gy[1] * gy[2] + gy[0] * ce[1] * gy[2] + gy[0] * gy[1] * ce[2];
                        double cxxx = gx[0] * gx[1] * gx[2];
                        double cxxy = gx[0] * gx[1] * gy[2] + gx[0] * gy[1] * gx[2] + gy[0] * gx[1] * gx[2];
                        double cxyy = gy[0] * gy[1] * gx[2] + gy[0] * gx[1] * gy[2] + gx[0] * gy[1] * gy[2];
                        double cyyy = gy[0] * gy[1] * gy[2];
                        double pr = mean[0] * mean[1] * mean[2];
                        double rcp = fabs(pr) > 0.0 ? 1.0 / pr : 0.0;
                        bx = (c0 * G(0) + cx * G(2) + cy * G(3) + cxx * G(5) + cxy * G(6) + cyy * G(7) + cxxx * G(9) +
                              cxxy * G(10) + cxyy * G(11) + cyyy * G(12)) *
                             rcp;
                        by = (c0 * G(1) + cx * G(3) + cy * G(4) + cxx * G(6) + cxy * G(7) + cyy * G(8) + cxxx * G(10) +
                              cxxy * G(11) + cxyy * G(12) + cyyy * G(13)) *
                             rcp;
                    }
#undef G
                }
                T.xB[k] = bx;
                T.yB[k] = by;
            }
        }
        __syncthreads();
        {  // staged records -> memory: per cell nlc * 3 consecutive doubles starting at layer l0
            const int row = nlc * 3, total = nValid * row;
            if (nlc == T.L) {
                // the whole tracer in one chunk: the rows of the block's cells follow each other in memory as they do
                // in the staging area, so this is one linear copy (16 bytes a lane; the block's first record sits at a
                // multiple of 64 L records)
                IrG *dst = T.rec + IR_REC(T, cat, pc, c0, 0);
                typedef double IrPair __attribute__((ext_vector_type(2)));
                const int pairs = total >> 1;
                for (int i = threadIdx.x; i < pairs; i += blockDim.x)
                    ((__attribute__((address_space(1))) IrPair *)dst)[i] = reinterpret_cast<const IrPair *>(stage)[i];
                if ((total & 1) && threadIdx.x == 0) dst[total - 1] = stage[total - 1];
            } else {
                for (int i = threadIdx.x; i < total; i += blockDim.x) {
                    const int cl = i / row;
                    T.rec[IR_REC(T, cat, pc, c0 + cl, l0) + (i - cl * row)] = stage[i];
                }
            }
        }
        __syncthreads();  // the next chunk reuses the staging rows; later tracers read these records
        }
    }
}

// find_departure_points :5255-5349
__global__ void k_ir_departure_points(IrDev d, double dt)
{
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= d.nVertices) return;
    double dx = -d.u[v] * dt, dy = -d.v[v] * dt;
    d.dp[v] = dx;
    d.dp[(size_t)d.padV + v] = dy;
    if (sqrt(dx * dx + dy * dy) > d.minLen[v]) atomicAdd(d.flag + 1, 1);
}

// shift_vertices_of_departure_triangle :6270-6500 followed by the signed area (:6231), then stored
// Set-up: for each of an edge's four (six) remap cells and each of its two vertices, the place of that vertex among the
// cell's vertices (0 when it is not one of them, as the search of shift_vertices_of_departure_triangle :6270 leaves
// it) and the places before and after it, packed place | before << 8 | after << 16.
__global__ void k_ir_init_tri_slots(IrDev d)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d.nEdges) return;
    const size_t pe = d.padE, pc = d.padC;
    for (int j = 0; j < d.nRemap; j++) {
        const int cell = d.coer[(size_t)j * pe + e];
        for (int tvoe = 0; tvoe < 2; tvoe++) {
            int packed = 0;
            if (cell < d.nCells) {
                const int n = d.nEdgesOnCell[cell];
                const int vShared = d.voe[(size_t)tvoe * pe + e];
                int tvoc = 0;
                for (int iv = 0; iv < n; iv++)
                    if (d.voc[(size_t)iv * pc + cell] == vShared) tvoc = iv;
                const int im1 = tvoc - 1 < 0 ? tvoc - 1 + n : tvoc - 1;
                const int ip1 = tvoc + 1 >= n ? tvoc + 1 - n : tvoc + 1;
                packed = tvoc | (im1 << 8) | (ip1 << 16);
            }
            d.triSlot[(size_t)(j * 2 + tvoe) * pe + e] = packed;
        }
    }
}

// One departure triangle: its source cell, its vertices moved to that cell's frame (shift_vertices :6270), its signed
// area.  `slot` = d.triSlot of (source cell, edge vertex tvoe); xe, ye = that vertex in the edge's frame.
__device__ inline void emitTriangle(const IrDev &d, int e, int t, double xt[3], double yt[3], int slot, double xe, double ye,
                                    int cell, double e1x, double e1y, double e2x, double e2y, int sign)
{
    const size_t pe = d.padE, pc = d.padC;
    d.triCell[(size_t)t * pe + e] = cell;
    double area = 0.0;
    if (cell < d.nCells) {
        const int tvoc = slot & 255;
        const double xc = d.xVOC[(size_t)tvoc * pc + cell], yc = d.yVOC[(size_t)tvoc * pc + cell];
        if (d.onSphere) {
            const int im1 = (slot >> 8) & 255, ip1 = (slot >> 16) & 255;
            double c1x = d.xVOC[(size_t)im1 * pc + cell] - xc, c1y = d.yVOC[(size_t)im1 * pc + cell] - yc;
            double c2x = d.xVOC[(size_t)ip1 * pc + cell] - xc, c2y = d.yVOC[(size_t)ip1 * pc + cell] - yc;
            double cp = cross2(e1x, e1y, e2x, e2y);
            if (fabs(cp) < EPS11) atomicOr(d.flag, FLAG_PARALLEL);
            if (cp > 0.0) {
                double tx = e1x, ty = e1y;
                e1x = e2x;
                e1y = e2y;
                e2x = tx;
                e2y = ty;
            }
            double denom = e1x * e2y - e2x * e1y;
            for (int k = 0; k < 3; k++) {
                xt[k] = xt[k] - xe;
                yt[k] = yt[k] - ye;
                double ca = (xt[k] * e2y - yt[k] * e2x) / denom;
                double cb = (yt[k] * e1x - xt[k] * e1y) / denom;
                xt[k] = xc + ca * c1x + cb * c2x;
                yt[k] = yc + ca * c1y + cb * c2y;
            }
        } else {
            for (int k = 0; k < 3; k++) {
                xt[k] = xt[k] - xe + xc;
                yt[k] = yt[k] - ye + yc;
            }
        }
        area = fabs(0.5 * ((xt[1] - xt[0]) * (yt[2] - yt[0]) - (yt[1] - yt[0]) * (xt[2] - xt[0])));
        area = area * sign;
    }
    for (int k = 0; k < 3; k++) {
        d.xTri[(size_t)(t * 3 + k) * pe + e] = xt[k];
        d.yTri[(size_t)(t * 3 + k) * pe + e] = yt[k];
    }
    d.triArea[(size_t)t * pe + e] = area;
}

template <int VDEG>
__global__ void __launch_bounds__(128) k_ir_triangles(IrDev d)
{
    constexpr int NR = VDEG == 4 ? 6 : 4;  // side edges / remap cells of an edge
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d.nEdges) return;
    const size_t pe = d.padE, pv = d.padV;
    int masked = 0;
    const int v1 = d.voe[e], v2 = d.voe[pe + e];
    const double d1x = d.dp[v1], d1y = d.dp[pv + v1], d2x = d.dp[v2], d2y = d.dp[pv + v2];
    if (d.remapEdge[e] == 1) {
        if (d1x * d1x + d1y * d1y > 0.0) masked = 1;
        if (d2x * d2x + d2y * d2y > 0.0) masked = 1;
    }
    if (!masked) {
        d.triCount[e] = 0;
        return;
    }
    // everything the branches below may read of this edge is requested here, in one go: read where it is needed,
    // each of these loads was a memory round trip of its own behind a branch (the kernel spent 84 % of its
    // wave-cycles waiting)
    double xo[NR + 2], yo[NR + 2];  // the edge's two vertices, then the far vertices of E1..E4 (E1..E6)
    int en4[NR], cell4[NR], slot8[2 * NR];
#pragma unroll
    for (int j = 0; j < 2 * NR; j++) slot8[j] = d.triSlot[(size_t)j * pe + e];
#pragma unroll
    for (int j = 0; j < NR + 2; j++) {
        xo[j] = d.xVOE[(size_t)j * pe + e];
        yo[j] = d.yVOE[(size_t)j * pe + e];
    }
#pragma unroll
    for (int j = 0; j < NR; j++) {
        en4[j] = d.eoer[(size_t)j * pe + e];
        cell4[j] = d.coer[(size_t)j * pe + e];
    }
    double ev[2][2], dp[2][2];
    ev[0][0] = xo[0];
    ev[0][1] = yo[0];
    ev[1][0] = xo[1];
    ev[1][1] = yo[1];
    dp[0][0] = ev[0][0] + d1x;
    dp[0][1] = ev[0][1] + d1y;
    dp[1][0] = ev[1][0] + d2x;
    dp[1][1] = ev[1][1] + d2y;
    int tc = 0;
    // side triangles :5711-5968
#pragma unroll
    for (int k = 0; k < 2; k++) {
#pragma unroll
        for (int side = 0; side < 2; side++) {
            int ieoe = k + 2 * side;  // 0-based E index: E1,E3 at V1; E2,E4 at V2
            int en = en4[ieoe];
            bool hit = false;
            double ix = 0.0, iy = 0.0, nx = 0.0, ny = 0.0;
            if (en < d.nEdges) {
                nx = xo[ieoe + 2];
                ny = yo[ieoe + 2];
                hit = lineIntersect(dp[0][0], dp[0][1], dp[1][0], dp[1][1], ev[k][0], ev[k][1], nx, ny, ix, iy);
            }
            if (hit) {
                double xt[3] = {ev[k][0], dp[k][0], ix}, yt[3] = {ev[k][1], dp[k][1], iy};
                const int sign = side == 0 ? 1 : -1;
                if (VDEG == 3) {
                    int cell = cell4[k + 2];
                    int other = ieoe + 2;
                    if (other > 3) other = other - 4;
                    double ox = xo[other + 2], oy = yo[other + 2];
                    if (tc < NTRI)
                        emitTriangle(d, e, tc, xt, yt, slot8[(k + 2) * 2 + k], ev[k][0], ev[k][1], cell, nx - ev[k][0],
                                     ny - ev[k][1], ox - ev[k][0], oy - ev[k][1], sign);
                    tc = tc + 1;
                } else {
                    // quadrilateral mesh, :5785-5958: the second basis vector runs along E5 / E6; where the line D1-D2
                    // also crosses E5 / E6 the side triangle is cut in two, one part on each side of the main edge's line
                    const double e2x = xo[NR + k] - ev[k][0], e2y = yo[NR + k] - ev[k][1];  // towards V7 / V8
                    double e1x = nx - ev[k][0], e1y = ny - ev[k][1];
                    int cIdx = k + 2;  // C3 / C4
                    bool hit56 = false;
                    double jx = 0.0, jy = 0.0;
                    if (en4[NR - 2 + k] < d.nEdges)
                        hit56 = lineIntersect(dp[0][0], dp[0][1], dp[1][0], dp[1][1], ev[k][0], ev[k][1], xo[NR + k], yo[NR + k], jx,
                                              jy);
                    if (hit56) {
                        xt[2] = jx;
                        yt[2] = jy;
                        double f1x, f1y;
                        if (side == 0) {  // IP on the left, this part on the right: C5 / C6, first basis vector along E3 / E4
                            cIdx = k + 4;
                            f1x = xo[ieoe + 4] - ev[k][0];
                            f1y = yo[ieoe + 4] - ev[k][1];
                        } else {  // IP on the right, this part on the left: C3 / C4, first basis vector along E1 / E2
                            f1x = xo[ieoe] - ev[k][0];
                            f1y = yo[ieoe] - ev[k][1];
                        }
                        if (tc < NTRI)
                            emitTriangle(d, e, tc, xt, yt, slot8[cIdx * 2 + k], ev[k][0], ev[k][1], cell4[cIdx], f1x, f1y, e2x, e2y,
                                         sign);
                        tc = tc + 1;
                        // V1 / V2, IP0, IP: the part on the side where IP lies
                        double xs[3] = {ev[k][0], jx, ix}, ys[3] = {ev[k][1], jy, iy};
                        const int c2 = side == 0 ? k + 2 : k + 4;
                        if (tc < NTRI)
                            emitTriangle(d, e, tc, xs, ys, slot8[c2 * 2 + k], ev[k][0], ev[k][1], cell4[c2], e1x, e1y, e2x, e2y, sign);
                        tc = tc + 1;
                    } else {
                        if (side == 1) cIdx = k + 4;  // IP and triangle in the right half-plane: C5 / C6
                        if (tc < NTRI)
                            emitTriangle(d, e, tc, xt, yt, slot8[cIdx * 2 + k], ev[k][0], ev[k][1], cell4[cIdx], e1x, e1y, e2x, e2y,
                                         sign);
                        tc = tc + 1;
                    }
                }
                dp[k][0] = ix;
                dp[k][1] = iy;
            }
        }
    }
    // central triangles :5970-6183
    double mx, my;
    bool hitMain = lineIntersect(dp[0][0], dp[0][1], dp[1][0], dp[1][1], ev[0][0], ev[0][1], ev[1][0], ev[1][1], mx, my);
    bool doCentral = hitMain;
    if (!hitMain) {
        // quadrilateral_area(V1, V2, D2, D1) :9137-9158
        double t1 = triArea2(ev[0][0], ev[0][1], ev[1][0], ev[1][1], dp[1][0], dp[1][1]);
        double t2 = triArea2(ev[0][0], ev[0][1], dp[1][0], dp[1][1], dp[0][0], dp[0][1]);
        doCentral = (t1 + t2) > 0.0;
    }
    if (doCentral) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int o = 1 - k;
            double xt[3], yt[3];
            if (hitMain) {  // V, D, IP0
                xt[0] = ev[k][0];  yt[0] = ev[k][1];
                xt[1] = dp[k][0];  yt[1] = dp[k][1];
                xt[2] = mx;        yt[2] = my;
            } else if (k == 0) {  // V1, V2, D1
                xt[0] = ev[0][0];  yt[0] = ev[0][1];
                xt[1] = ev[1][0];  yt[1] = ev[1][1];
                xt[2] = dp[0][0];  yt[2] = dp[0][1];
            } else {  // V2, D1, D2
                xt[0] = ev[1][0];  yt[0] = ev[1][1];
                xt[1] = dp[0][0];  yt[1] = dp[0][1];
                xt[2] = dp[1][0];  yt[2] = dp[1][1];
            }
            bool inHP = inHalfPlane(ev[0][0], ev[0][1], ev[1][0], ev[1][1], dp[k][0], dp[k][1]);
            int cell = inHP ? cell4[0] : cell4[1];
            int other = inHP ? k : k + 2;
            double ox = xo[other + 2], oy = yo[other + 2];
            if (tc < NTRI)
                emitTriangle(d, e, tc, xt, yt, inHP ? slot8[k] : slot8[2 + k], ev[k][0], ev[k][1], cell, ev[o][0] - ev[k][0],
                             ev[o][1] - ev[k][1], ox - ev[k][0], oy - ev[k][1], inHP ? 1 : -1);
            tc = tc + 1;
        }
    }
    d.triCount[e] = tc < NTRI ? tc : NTRI;
}

// integrate_fluxes_over_triangles :6667-6967, all tracers of one (edge, category).  The loop over the
// tracers is outermost, like the reference's, so each flux is the same sum over triangles in the same
// order; the edge's few triangles are re-read from cache for every tracer plane.
__global__ void __launch_bounds__(256) k_ir_flux(IrDev d, const IrTabEntry *__restrict__ tab, int nT)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d.nEdges) return;
    const int cat = blockIdx.y;
    const size_t pe = d.padE, pc = d.padC;
    const int cnt = d.triCount[e];
    for (int ti = 0; ti < nT; ti++) {
        const IrTabEntry T = tab[ti];
        for (int layer = 0; layer < T.L; layer++) {
            double flux = 0.0;
            for (int t = 0; t < cnt; t++) {
                const double area = d.triArea[(size_t)t * pe + e];
                if (area == 0.0) continue;
                const int c = d.triCell[(size_t)t * pe + e];
                const double x1 = d.xTri[(size_t)(t * 3) * pe + e], x2 = d.xTri[(size_t)(t * 3 + 1) * pe + e],
                             x3 = d.xTri[(size_t)(t * 3 + 2) * pe + e];
                const double y1 = d.yTri[(size_t)(t * 3) * pe + e], y2 = d.yTri[(size_t)(t * 3 + 1) * pe + e],
                             y3 = d.yTri[(size_t)(t * 3 + 2) * pe + e];
                double xq[6], yq[6];  // get_triangle_quadrature_points :6605-6644
                xq[0] = Q1QP * x1 + Q1QP * x2 + Q2QP * x3;  yq[0] = Q1QP * y1 + Q1QP * y2 + Q2QP * y3;
                xq[1] = Q1QP * x1 + Q2QP * x2 + Q1QP * x3;  yq[1] = Q1QP * y1 + Q2QP * y2 + Q1QP * y3;
                xq[2] = Q2QP * x1 + Q1QP * x2 + Q1QP * x3;  yq[2] = Q2QP * y1 + Q1QP * y2 + Q1QP * y3;
                xq[3] = Q3QP * x1 + Q4QP * x2 + Q4QP * x3;  yq[3] = Q3QP * y1 + Q4QP * y2 + Q4QP * y3;
                xq[4] = Q4QP * x1 + Q3QP * x2 + Q4QP * x3;  yq[4] = Q4QP * y1 + Q3QP * y2 + Q4QP * y3;
                xq[5] = Q4QP * x1 + Q4QP * x2 + Q3QP * x3;  yq[5] = Q4QP * y1 + Q4QP * y2 + Q3QP * y3;
                double val[6] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (j < T.nLev) {
                        const IrTabEntry &A = tab[T.chain[j]];
                        const IrG *r = A.rec + IR_REC(A, cat, pc, c, A.L == 1 ? 0 : layer);
                        const double cj = r[0], gxj = r[1], gyj = r[2];
#pragma unroll
                        for (int q = 0; q < 6; q++) val[q] = val[q] * (cj + gxj * xq[q] + gyj * yq[q]);
                    }
                }
                double integral = 0.0;
#pragma unroll
                for (int q = 0; q < 6; q++) integral = integral + (q < 3 ? W1QP : W2QP) * val[q];
                if (ti == 0) {  // "IR negative reconstructed ice area", :6903-6951
                    bool neg = false;
#pragma unroll
                    for (int q = 0; q < 6; q++) neg = neg || (val[q] < 0.0);
                    if (neg) atomicOr(d.flag, FLAG_NEGAREA);
                }
                flux = flux + area * integral;
            }
            T.flux[(size_t)(cat * T.L + layer) * pe + e] = flux;
        }
    }
}

// The same fluxes with one thread per (edge, tracer, category, chunk of up to 8 layers): blockIdx.x runs
// over these groups (fastest in dispatch order, so the blocks that share an edge chunk's triangles run
// together), blockIdx.y over chunks of edges.  The quadrature points of a triangle and the product of
// the single-layer ancestors are evaluated once for all layers of the group; every flux is still the
// reference's sum over triangles of area * (sum over quadrature points), in the same order.
struct IrPlane {
    int ti, layer0, cat;
};

__global__ void __launch_bounds__(256) k_ir_flux_planes(IrDev d, const IrTabEntry *__restrict__ tab,
                                                        const IrPlane *__restrict__ planes, int nPlanes)
{
    // all plane groups of one block of edges read the same triangles: keep them on one XCD
    unsigned eb, pi;
    if (!xcdTile((d.nEdges + 255u) / 256u, (unsigned)nPlanes, eb, pi)) return;
    const int e = eb * blockDim.x + threadIdx.x;
    if (e >= d.nEdges) return;
    const IrPlane pl = planes[pi];
    const IrTabEntry T = tab[pl.ti];
    const int cat = pl.cat, layer0 = pl.layer0;
    const int nl = min(IR_LCHUNK, T.L - layer0);
    const size_t pe = d.padE, pc = d.padC;
    const int cnt = d.triCount[e];
    // ancestors (and the tracer itself if it has one layer) whose value does not depend on the layer
    int nShared = 0;
    const double *recp[4];  // records of category cat, starting at layer0 for a multi-layer chain member
    int Lk[4], Sk[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        recp[j] = nullptr;
        Lk[j] = 1;
        Sk[j] = 1;
        if (j < T.nLev) {
            const IrTabEntry &A = tab[T.chain[j]];
            Lk[j] = A.L;
            Sk[j] = A.S;
            recp[j] = A.rec + IR_REC(A, cat, pc, 0, A.L == 1 ? 0 : layer0);
            if (A.L == 1 && nShared == j) nShared = j + 1;
        }
    }
    double flux[IR_LCHUNK];
#pragma unroll
    for (int ly = 0; ly < IR_LCHUNK; ly++) flux[ly] = 0.0;
    for (int t = 0; t < cnt; t++) {
        const double area = d.triArea[(size_t)t * pe + e];
        if (area == 0.0) continue;
        const int c = d.triCell[(size_t)t * pe + e];
        const double x1 = d.xTri[(size_t)(t * 3) * pe + e], x2 = d.xTri[(size_t)(t * 3 + 1) * pe + e],
                     x3 = d.xTri[(size_t)(t * 3 + 2) * pe + e];
        const double y1 = d.yTri[(size_t)(t * 3) * pe + e], y2 = d.yTri[(size_t)(t * 3 + 1) * pe + e],
                     y3 = d.yTri[(size_t)(t * 3 + 2) * pe + e];
        double xq[6], yq[6];  // get_triangle_quadrature_points :6605-6644
        xq[0] = Q1QP * x1 + Q1QP * x2 + Q2QP * x3;  yq[0] = Q1QP * y1 + Q1QP * y2 + Q2QP * y3;
        xq[1] = Q1QP * x1 + Q2QP * x2 + Q1QP * x3;  yq[1] = Q1QP * y1 + Q2QP * y2 + Q1QP * y3;
        xq[2] = Q2QP * x1 + Q1QP * x2 + Q1QP * x3;  yq[2] = Q2QP * y1 + Q1QP * y2 + Q1QP * y3;
        xq[3] = Q3QP * x1 + Q4QP * x2 + Q4QP * x3;  yq[3] = Q3QP * y1 + Q4QP * y2 + Q4QP * y3;
        xq[4] = Q4QP * x1 + Q3QP * x2 + Q4QP * x3;  yq[4] = Q4QP * y1 + Q3QP * y2 + Q4QP * y3;
        xq[5] = Q4QP * x1 + Q4QP * x2 + Q3QP * x3;  yq[5] = Q4QP * y1 + Q4QP * y2 + Q3QP * y3;
        double pv[6] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j < nShared) {
                const double *r = recp[j] + (size_t)c * Sk[j] * 3;  // nShared members have one layer
                const double cj = r[0], gxj = r[1], gyj = r[2];
#pragma unroll
                for (int q = 0; q < 6; q++) pv[q] = pv[q] * (cj + gxj * xq[q] + gyj * yq[q]);
            }
        }
        if (pl.ti == 0) {  // "IR negative reconstructed ice area", :6903-6951
            bool neg = false;
#pragma unroll
            for (int q = 0; q < 6; q++) neg = neg || (pv[q] < 0.0);
            if (neg) atomicOr(d.flag, FLAG_NEGAREA);
        }
#pragma unroll
        for (int ly = 0; ly < IR_LCHUNK; ly++) {
            if (ly < nl) {
                double val[6];
#pragma unroll
                for (int q = 0; q < 6; q++) val[q] = pv[q];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (j >= nShared && j < T.nLev) {
                        const double *r = recp[j] + ((size_t)c * Sk[j] + (Lk[j] == 1 ? 0 : ly)) * 3;
                        const double cj = r[0], gxj = r[1], gyj = r[2];
#pragma unroll
                        for (int q = 0; q < 6; q++) val[q] = val[q] * (cj + gxj * xq[q] + gyj * yq[q]);
                    }
                }
                double integral = 0.0;
#pragma unroll
                for (int q = 0; q < 6; q++) integral = integral + (q < 3 ? W1QP : W2QP) * val[q];
                flux[ly] = flux[ly] + area * integral;
            }
        }
    }
#pragma unroll
    for (int ly = 0; ly < IR_LCHUNK; ly++)
        if (ly < nl) T.flux[(size_t)(cat * T.L + layer0 + ly) * pe + e] = flux[ly];
}

// The same fluxes with one thread per (edge, category) that walks the tracer tree depth first, triangle by
// triangle: the quadrature points of a triangle are formed once, and the value of every tracer with children
// at those points is kept (one set of six per level of the tree) for its children to multiply on, as the
// reference keeps triangleValue for them (:6820-6900) - here in registers instead of memory.  A tracer with
// children must have one layer for that.  The partial sums of all fluxes sit in LDS, one row of 256 per
// layer field; each is still the sum over triangles, in their order, of area * (sum over quadrature points).
struct IrStep {
    int ti, level, keep, slot;  // tracer, depth in the tree, has children, first row of its fluxes in LDS
};
#define IR_WALK_MAXSLOTS 32

// one tracer (all its layers) on one triangle: value at the quadrature points times the parent's (level > 0),
// the integral added to the tracer's rows in LDS, the values left for the children if it has any
template <int LEVEL>
__device__ __forceinline__ void irWalkStep(const IrDev &d, const IrG *__restrict__ rec, int L,
                                           const IrG *__restrict__ nextRec, double (&cur)[3], double *f, double area,
                                           const double (&xq)[6], const double (&yq)[6], const double (&par)[6],
                                           double (&own)[6], bool keep, bool first)
{
    for (int ly = 0; ly < L; ly++) {
        // `cur` holds this layer's record; the next one (the next layer's, or the first of the next tracer) is
        // requested now, so that it travels while this layer is worked on
        const double cj = cur[0], gxj = cur[1], gyj = cur[2];
        const IrG *nx = ly + 1 < L ? rec + (ly + 1) * 3 : nextRec;
        if (nx) {
            cur[0] = nx[0];
            cur[1] = nx[1];
            cur[2] = nx[2];
        }
        double val[6];
#pragma unroll
        for (int q = 0; q < 6; q++) {
            val[q] = cj + gxj * xq[q] + gyj * yq[q];
            if (LEVEL > 0) val[q] = par[q] * val[q];
        }
        if (first) {  // "IR negative reconstructed ice area", :6903-6951
            bool neg = false;
#pragma unroll
            for (int q = 0; q < 6; q++) neg = neg || (val[q] < 0.0);
            if (neg) atomicOr(d.flag, FLAG_NEGAREA);
        }
        double integral = 0.0;
#pragma unroll
        for (int q = 0; q < 6; q++) integral = integral + (q < 3 ? W1QP : W2QP) * val[q];
        f[ly * 256] = f[ly * 256] + area * integral;
        if (LEVEL < 3 && keep) {  // one layer (checked when the steps were made)
#pragma unroll
            for (int q = 0; q < 6; q++) own[q] = val[q];
        }
    }
}

__global__ void __launch_bounds__(256, 4) k_ir_flux_walk(IrDev d, const IrTabEntry *__restrict__ tab,
                                                      const IrStep *__restrict__ steps, int nSteps, int nSlots)
{
    extern __shared__ double fl[];  // [nSlots][256]
    unsigned eb, ucat;
    if (!xcdTile((d.nEdges + 255u) / 256u, (unsigned)d.nCat, eb, ucat)) return;
    const int tid = threadIdx.x;
    const int e = eb * blockDim.x + tid;
    const int cat = (int)ucat;
    const size_t pe = d.padE, pc = d.padC;
    for (int sl = 0; sl < nSlots; sl++) fl[sl * 256 + tid] = 0.0;
    const int eSafe = e < d.nEdges ? e : 0;
    const int cnt = e < d.nEdges ? d.triCount[e] : 0;
    // area and source cell of a triangle are requested one triangle ahead, and the first record of the next
    // triangle with the last step of this one: a triangle then starts with one memory round trip in front of it
    // (its vertices) instead of two in a row (its source cell, then that cell's first record)
    // (the first triangle's area and cell are requested together with the count, not after it)
    const double area0 = d.triArea[eSafe];
    const int cell0 = d.triCell[eSafe];
    double areaN = cnt > 0 ? area0 : 0.0;
    int cellN = cnt > 0 ? cell0 : 0;
    double cur[3] = {0.0, 0.0, 0.0};
    bool curReady = false;  // `cur` holds the first record of the triangle at hand
    const IrTabEntry T0 = tab[steps[0].ti];
    for (int t = 0; t < cnt; t++) {
        const double area = areaN;
        const int c = cellN;
        if (t + 1 < cnt) {
            areaN = d.triArea[(size_t)(t + 1) * pe + e];
            cellN = d.triCell[(size_t)(t + 1) * pe + e];
        }
        if (area == 0.0) continue;  // (nothing was requested for it: see the last step below)
        const double x1 = d.xTri[(size_t)(t * 3) * pe + e], x2 = d.xTri[(size_t)(t * 3 + 1) * pe + e],
                     x3 = d.xTri[(size_t)(t * 3 + 2) * pe + e];
        const double y1 = d.yTri[(size_t)(t * 3) * pe + e], y2 = d.yTri[(size_t)(t * 3 + 1) * pe + e],
                     y3 = d.yTri[(size_t)(t * 3 + 2) * pe + e];
        double xq[6], yq[6];  // get_triangle_quadrature_points :6605-6644
        xq[0] = Q1QP * x1 + Q1QP * x2 + Q2QP * x3;  yq[0] = Q1QP * y1 + Q1QP * y2 + Q2QP * y3;
        xq[1] = Q1QP * x1 + Q2QP * x2 + Q1QP * x3;  yq[1] = Q1QP * y1 + Q2QP * y2 + Q1QP * y3;
        xq[2] = Q2QP * x1 + Q1QP * x2 + Q1QP * x3;  yq[2] = Q2QP * y1 + Q1QP * y2 + Q1QP * y3;
        xq[3] = Q3QP * x1 + Q4QP * x2 + Q4QP * x3;  yq[3] = Q3QP * y1 + Q4QP * y2 + Q4QP * y3;
        xq[4] = Q4QP * x1 + Q3QP * x2 + Q4QP * x3;  yq[4] = Q4QP * y1 + Q3QP * y2 + Q4QP * y3;
        xq[5] = Q4QP * x1 + Q4QP * x2 + Q3QP * x3;  yq[5] = Q4QP * y1 + Q4QP * y2 + Q3QP * y3;
        double lv0[6], lv1[6], lv2[6];  // values of the current ancestors at the quadrature points, by level
#pragma unroll
        for (int q = 0; q < 6; q++) lv0[q] = lv1[q] = lv2[q] = 1.0;
        if (!curReady) {
            const IrG *r0 = T0.rec + IR_REC(T0, cat, pc, c, 0);
            cur[0] = r0[0];
            cur[1] = r0[1];
            cur[2] = r0[2];
        }
        for (int si = 0; si < nSteps; si++) {
            const IrStep S = steps[si];  // the same for every lane: the branches below are scalar
            const IrTabEntry T = tab[S.ti];
            const IrG *rec = T.rec + IR_REC(T, cat, pc, c, 0);
            const IrG *nextRec = nullptr;
            if (si + 1 < nSteps) {
                const IrTabEntry &N = tab[steps[si + 1].ti];
                nextRec = N.rec + IR_REC(N, cat, pc, c, 0);
            } else {
                curReady = t + 1 < cnt && areaN != 0.0;
                if (curReady) nextRec = T0.rec + IR_REC(T0, cat, pc, cellN, 0);
            }
            double *f = fl + S.slot * 256 + tid;
            const bool first = S.ti == 0;
            switch (S.level) {
            case 0:
                irWalkStep<0>(d, rec, T.L, nextRec, cur, f, area, xq, yq, lv0, lv0, S.keep != 0, first);
                break;
            case 1:
                irWalkStep<1>(d, rec, T.L, nextRec, cur, f, area, xq, yq, lv0, lv1, S.keep != 0, first);
                break;
            case 2:
                irWalkStep<2>(d, rec, T.L, nextRec, cur, f, area, xq, yq, lv1, lv2, S.keep != 0, first);
                break;
            default:
                irWalkStep<3>(d, rec, T.L, nextRec, cur, f, area, xq, yq, lv2, lv2, false, first);
            }
        }
    }
    if (e >= d.nEdges) return;
    for (int si = 0; si < nSteps; si++) {
        const IrStep S = steps[si];
        const IrTabEntry T = tab[S.ti];
        for (int ly = 0; ly < T.L; ly++) T.flux[(size_t)(cat * T.L + ly) * pe + e] = fl[(S.slot + ly) * 256 + tid];
    }
}

// update_mass_and_tracers :7125-7523, all tracers of one (owned cell, category).  A block is 256 consecutive owned
// cells.  Edges are numbered after the lower-numbered of their two cells, so the edges of a block's cells are,
// but for those shared with earlier blocks, one run of about 3 x 256 consecutive edges: per layer field the block
// reads that run (IR_UPD_WIN fluxes from `updWin[block]` on, chosen at set-up to hold as many of the block's edges
// as possible) with coalesced loads into LDS, one layer field ahead of its use, and a thread takes its six fluxes
// from there; only the edges outside the run are gathered from memory.  (With all six gathered, the kernel took
// 0.30 ms at QU60, of which 0.10 ms were the gathers' own.)
#ifndef IR_UPD_BLOCK
#define IR_UPD_BLOCK 256
#endif
#ifndef IR_UPD_PER
#define IR_UPD_PER 3
#endif
#define IR_UPD_WIN (IR_UPD_PER * IR_UPD_BLOCK)
// the layer field after (ti, layer) in the order (tracer, layer); ti == number of tracers after the last one
__device__ inline void irNextField(const IrTabEntry *__restrict__ tab, int &ti, int &layer)
{
    if (layer + 1 < tab[ti].L)
        layer++;
    else {
        ti++;
        layer = 0;
    }
}
__device__ inline const IrG *irFluxPlane(const IrTabEntry *__restrict__ tab, int ti, int layer, int cat, size_t pe)
{
    return tab[ti].flux + (size_t)(cat * tab[ti].L + layer) * pe;
}
template <int M>
__global__ void __launch_bounds__(IR_UPD_BLOCK) k_ir_update(IrDev d, const IrTabEntry *__restrict__ tab, int nT)
{
    __shared__ double winStore[2 * IR_UPD_WIN];
    // addressed as LDS explicitly: with generic pointers the compiler merged the LDS read and the gather below into
    // one access through a pointer of either kind and then failed to select instructions for it
    typedef __attribute__((address_space(3))) double LdsDouble;
    LdsDouble *win = (LdsDouble *)winStore;
    const int tid = threadIdx.x;
    const int cRaw = blockIdx.x * blockDim.x + tid;
    const bool live = cRaw < d.nCellsSolve;
    const int c = live ? cRaw : d.nCellsSolve - 1;  // every thread stays for the barriers; the spare ones store nothing
    const int cat = blockIdx.y;
    const size_t pc = d.padC, pe = d.padE;
    const int n = d.nEdgesOnCell[c];
    const int eLo = d.updWin[blockIdx.x];
    int rel[M];  // edge minus eLo; inside the window: taken from LDS
    int ed[M];
    unsigned plus = 0;
#pragma unroll
    for (int ie = 0; ie < M; ie++) {
        ed[ie] = d.nEdges;
        if (ie < n) ed[ie] = d.eoc[(size_t)ie * pc + c];
    }
    const double area = d.areaCell[c];
    {
        // (the edges' first cells in a pass of their own: asked for slot by slot, each waited for its edge number)
        int first[M];
#pragma unroll
        for (int ie = 0; ie < M; ie++) {
            first[ie] = -1;
            if (ie < n) first[ie] = d.coe[ed[ie]];
        }
#pragma unroll
        for (int ie = 0; ie < M; ie++) {
            rel[ie] = -1;
            if (ie < n) {
                rel[ie] = ed[ie] - eLo;
                if (rel[ie] >= IR_UPD_WIN) rel[ie] = -1;
                if (c == first[ie]) plus |= 1u << ie;
            }
        }
    }
    bool negMass = false;
    // what a thread copies of a window: IR_UPD_PER entries, 256 apart
    int wi[IR_UPD_PER];
#pragma unroll
    for (int j = 0; j < IR_UPD_PER; j++) wi[j] = min(eLo + j * IR_UPD_BLOCK + tid, (int)pe - 1);
    // Layer fields are taken in the order (tracer, layer).  The window of field i + 2 and the old product of field
    // i + 1 are requested while field i is worked on; the window of field i + 1, requested one field earlier, goes
    // to LDS at the end of field i: every load has a whole field's work to arrive in.
    int ti1 = 0, ly1 = 0;  // field i + 1
    irNextField(tab, ti1, ly1);
    int ti2 = ti1, ly2 = ly1;  // field i + 2
    if (ti2 < nT) irNextField(tab, ti2, ly2);
    double pendW[IR_UPD_PER], prodNext = 0.0;
    {
        const IrG *f0 = irFluxPlane(tab, 0, 0, cat, pe);
#pragma unroll
        for (int j = 0; j < IR_UPD_PER; j++) win[j * IR_UPD_BLOCK + tid] = f0[wi[j]];
        if (ti1 < nT) {
            const IrG *f1 = irFluxPlane(tab, ti1, ly1, cat, pe);
#pragma unroll
            for (int j = 0; j < IR_UPD_PER; j++) pendW[j] = f1[wi[j]];
        }
        prodNext = tab[0].prod[(size_t)cat * tab[0].L * pc + c];
    }
    __syncthreads();
    int buf = 0;
    for (int ti = 0; ti < nT; ti++) {
        const IrTabEntry T = tab[ti];
        const bool hasParent = T.parent >= 0;
        const IrTabEntry *P = hasParent ? &tab[T.parent] : nullptr;
        for (int layer = 0; layer < T.L; layer++) {
            const int l = cat * T.L + layer;
            const size_t k = (size_t)l * pc + c;
            const bool more1 = ti1 < nT, more2 = ti2 < nT;
            double newW[IR_UPD_PER];
            if (more2) {
                const IrG *f2 = irFluxPlane(tab, ti2, ly2, cat, pe);
#pragma unroll
                for (int j = 0; j < IR_UPD_PER; j++) newW[j] = f2[wi[j]];
            }
            const double prodOld = prodNext;
            if (more1) prodNext = tab[ti1].prod[(size_t)(cat * tab[ti1].L + ly1) * pc + c];
            const IrG *fp = T.flux + (size_t)l * pe;
            // (requested before the gathers below, not after them where it is used)
            const double pp = hasParent ? P->prod[(size_t)(P->L == 1 ? cat : l) * pc + c] : 1.0;
            // (the six fluxes first, then their sum: with the gather and the addition in one pass of the loop, every
            // pass with a lane outside the window waited for its own load)
            double fe[M];
#pragma unroll
            for (int ie = 0; ie < M; ie++) fe[ie] = win[buf * IR_UPD_WIN + (rel[ie] >= 0 ? rel[ie] : 0)];
#pragma unroll
            for (int ie = 0; ie < M; ie++)
                if (ie < n && rel[ie] < 0) fe[ie] = fp[ed[ie]];
            double flux = 0.0;
#pragma unroll
            for (int ie = 0; ie < M; ie++)
                if (ie < n) flux = flux + fe[ie] * ((plus >> ie) & 1u ? 1.0 : -1.0);
            double nv;
            if (pp > 0.0)
                nv = (prodOld - (flux / area)) / pp;
            else
                nv = 0.0;
            if (live) {
                if (T.hasChild) T.prod[k] = pp * nv;  // read by the children below, by nothing after this kernel
                if (!hasParent) {  // :7465-7507
                    if (nv < -EPS11 * EPS11)
                        negMass = true;
                    else if (nv < 0.0)
                        nv = 0.0;
                }
                T.val[k] = nv;
            }
            if (more1) {
#pragma unroll
                for (int j = 0; j < IR_UPD_PER; j++) win[(buf ^ 1) * IR_UPD_WIN + j * IR_UPD_BLOCK + tid] = pendW[j];
            }
#pragma unroll
            for (int j = 0; j < IR_UPD_PER; j++) pendW[j] = newW[j];
            ti1 = ti2;
            ly1 = ly2;
            if (more2) irNextField(tab, ti2, ly2);
            // one barrier per layer field: the buffer written above was last read one field ago, before the barrier
            // that ended that field
            __syncthreads();
            buf ^= 1;
        }
    }
    if (negMass)
        __hip_atomic_fetch_or((__attribute__((address_space(1))) int *)d.flag, FLAG_NEGMASS, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT);  // "IR: Negative mass in IR", reported once the launch has ended
}

// zap_small_mass :8764-8886: thread = (owned cell, category).  A separate launch: the conservation sums of
// the reference are taken between the update and the zap.
__global__ void k_ir_zap(IrDev d, const IrTabEntry *__restrict__ tab, int nT)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCellsSolve) return;
    const int cat = blockIdx.y;
    IrG *mass = tab[0].val;
    double m = mass[(size_t)cat * d.padC + c];
    if (m > 0.0 && m < 1.0e-22) {
        mass[(size_t)cat * d.padC + c] = 0.0;
        for (int i = 1; i < nT; i++)
            for (int ly = 0; ly < tab[i].L; ly++) tab[i].val[(size_t)(cat * tab[i].L + ly) * d.padC + c] = 0.0;
    }
}

// recover_tracer_means :7541-7982 (config_recover_tracer_means_check): the product mass * tracer ... integrated
// over the cell from the reconstructions, sub-triangle (centre, vertex i, vertex i+1) by sub-triangle with the
// 6-point rule and the area fractions 0.25 dcEdge dvEdge / sum, against the product of the means.  One tracer per
// launch; a diagnostic, so every thread recomputes what it needs.
template <int M>
__global__ void __launch_bounds__(256) k_ir_recover_means(IrDev d, const IrTabEntry *__restrict__ tab, int ti,
                                                          double *recovered, int *count)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    const int cat = blockIdx.y;
    const size_t pc = d.padC;
    const IrTabEntry T = tab[ti];
    const int n = d.nEdgesOnCell[c];
    double frac[M], xv[M], yv[M];
    double sumArea = 0.0;
#pragma unroll
    for (int ie = 0; ie < M; ie++) {
        frac[ie] = 0.0;
        xv[ie] = yv[ie] = 0.0;
        if (ie < n) {
            const int e = d.eoc[(size_t)ie * pc + c];
            frac[ie] = 0.25 * d.dcEdge[e] * d.dvEdge[e];
            sumArea = sumArea + frac[ie];
            xv[ie] = d.xVOC[(size_t)ie * pc + c];
            yv[ie] = d.yVOC[(size_t)ie * pc + c];
        }
    }
    for (int layer = 0; layer < T.L; layer++) {
        double cell = 0.0;
#pragma unroll
        for (int ie = 0; ie < M; ie++) {
            if (ie < n) {
                const double fr = frac[ie] / sumArea;
                const int ip = ie + 1 < n ? ie + 1 : 0;
                const double x1 = 0.0, y1 = 0.0, x2 = xv[ie], y2 = yv[ie];
                double x3 = 0.0, y3 = 0.0;
#pragma unroll
                for (int k = 0; k < M; k++)
                    if (k == ip) {
                        x3 = xv[k];
                        y3 = yv[k];
                    }
                double xq[6], yq[6];
                xq[0] = Q1QP * x1 + Q1QP * x2 + Q2QP * x3;  yq[0] = Q1QP * y1 + Q1QP * y2 + Q2QP * y3;
                xq[1] = Q1QP * x1 + Q2QP * x2 + Q1QP * x3;  yq[1] = Q1QP * y1 + Q2QP * y2 + Q1QP * y3;
                xq[2] = Q2QP * x1 + Q1QP * x2 + Q1QP * x3;  yq[2] = Q2QP * y1 + Q1QP * y2 + Q1QP * y3;
                xq[3] = Q3QP * x1 + Q4QP * x2 + Q4QP * x3;  yq[3] = Q3QP * y1 + Q4QP * y2 + Q4QP * y3;
                xq[4] = Q4QP * x1 + Q3QP * x2 + Q4QP * x3;  yq[4] = Q4QP * y1 + Q3QP * y2 + Q4QP * y3;
                xq[5] = Q4QP * x1 + Q4QP * x2 + Q3QP * x3;  yq[5] = Q4QP * y1 + Q4QP * y2 + Q3QP * y3;
                double val[6] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (j < T.nLev) {
                        const IrTabEntry &A = tab[T.chain[j]];
                        const IrG *r = A.rec + IR_REC(A, cat, pc, c, A.L == 1 ? 0 : layer);
                        const double cj = r[0], gxj = r[1], gyj = r[2];
#pragma unroll
                        for (int q = 0; q < 6; q++) val[q] = (cj + gxj * xq[q] + gyj * yq[q]) * val[q];
                    }
                }
                double tri = 0.0;
#pragma unroll
                for (int q = 0; q < 6; q++) tri = tri + (q < 3 ? W1QP : W2QP) * val[q];
                cell = cell + fr * tri;
            }
        }
        const size_t k = (size_t)(cat * T.L + layer) * pc + c;
        recovered[k] = cell;
        const double mean = T.prod[k], difference = cell - mean;
        if (fabs(difference) > EPS11 * fabs(mean) && fabs(difference) > EPS11) atomicAdd(count, 1);
    }
}

// sum_tracers :7998-8111: one block per plane of tracer ti, sum over the owned cells of areaCell * (product
// of the chain's values); the order of additions differs from the reference's serial loop (tree reduction)
__global__ void k_ir_sum(IrDev d, const IrTabEntry *__restrict__ tab, int ti, double *out)
{
    __shared__ double sh[256];
    const IrTabEntry T = tab[ti];
    const int l = blockIdx.x, cat = l / T.L, layer = l - cat * T.L;
    double s = 0.0;
    for (int c = threadIdx.x; c < d.nCellsSolve; c += blockDim.x) {
        double p = 1.0;
        for (int j = 0; j < T.nLev; j++) {
            const IrTabEntry &A = tab[T.chain[j]];
            p = p * A.val[(size_t)(A.L == 1 ? cat : cat * A.L + layer) * d.padC + c];
        }
        s = s + d.areaCell[c] * p;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = blockDim.x / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[l] = sh[0];
}

__global__ void k_ir_zero_flags(int *p)
{
    if (threadIdx.x < 16) p[threadIdx.x] = 0;
}

__global__ void k_ir_zero_one(int *p)
{
    if (threadIdx.x == 0) *p = 0;
}

// ------------------------------------------------------------------------------------------ host side
static int uploadPlaneInts(sih_ctx *c, int *dst, const std::vector<int> &v)
{
    HIPCHK(hipMemcpyAsync(dst, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SIH_OK;
}

// The edge description of the mesh on the device, shared by the incremental remap and the upwind advection:
// validation of the host's index arrays, the internal edge numbering, the connectivity tables _eoc / _coe /
// _voe / _eov in internal numbering and the edge coordinates and lengths.
int sih_setup_edges(sih_ctx *c, const sih_ir_desc *m, const char *who)
{
    HIPCHK(hipSetDevice(c->device));
    const int nC = c->nCells, nV = c->nVertices, nE = c->nEdges, M = c->M, D = c->D;
    // validate connectivity before the device dereferences it
    for (int e = 0; e < nE; e++)
        for (int k = 0; k < 2; k++) {
            int cc = m->cellsOnEdge[(size_t)e * 2 + k], vv = m->verticesOnEdge[(size_t)e * 2 + k];
            if (cc < 0 || cc > nC + 1 || vv < 1 || vv > nV) {
                sih_set_error(std::string(who) + ": cellsOnEdge / verticesOnEdge index out of range");
                return SIH_ERR_ARG;
            }
        }
    for (int i = 0; i < nC; i++)
        for (int s = 0; s < M; s++) {
            int e = m->edgesOnCell[(size_t)i * M + s];
            if (e < 0 || e > nE + 1) {
                sih_set_error(std::string(who) + ": edgesOnCell index out of range");
                return SIH_ERR_ARG;
            }
        }
    for (size_t k = 0; k < (size_t)nV * D; k++)
        if (m->edgesOnVertex[k] < 0 || m->edgesOnVertex[k] > nE + 1) {
            sih_set_error(std::string(who) + ": edgesOnVertex index out of range");
            return SIH_ERR_ARG;
        }
    // internal edge numbering: edges follow the lower-numbered (internal) of their two cells
    c->edgePerm.resize(nE);
    for (int e = 0; e < nE; e++) c->edgePerm[e] = e;
    if (c->renumber) {
        std::vector<int> key(nE, nC);
        for (int e = 0; e < nE; e++)
            for (int k = 0; k < 2; k++) {
                int cc = m->cellsOnEdge[(size_t)e * 2 + k];
                if (cc >= 1 && cc <= nC) key[e] = std::min(key[e], c->cellInv[cc - 1]);
            }
        std::stable_sort(c->edgePerm.begin(), c->edgePerm.end(), [&](int a, int b) { return key[a] < key[b]; });
    }
    c->edgeInv.assign(nE + 1, nE);
    for (int i = 0; i < nE; i++) c->edgeInv[c->edgePerm[i]] = i;
    int rc;
    if (!c->d_edgePerm && (rc = sih_dev_alloc(c, (void **)&c->d_edgePerm, sizeof(int) * c->padE))) return rc;
    HIPCHK(hipMemcpyAsync(c->d_edgePerm, c->edgePerm.data(), sizeof(int) * nE, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));

    const struct {
        const char *name;
        SihLoc loc;
        int lead;
        bool isInt;
    } specs[] = {
        {"_eoc", LOC_CELL, M, true}, {"_coe", LOC_EDGE, 2, true}, {"_voe", LOC_EDGE, 2, true},
        {"_eov", LOC_VERTEX, D, true}, {"xEdge", LOC_EDGE, 1, false}, {"yEdge", LOC_EDGE, 1, false},
        {"zEdge", LOC_EDGE, 1, false}, {"dcEdge", LOC_EDGE, 1, false}, {"dvEdge", LOC_EDGE, 1, false},
    };
    for (const auto &s : specs)
        if ((rc = sih_add_field(c, s.name, s.loc, s.lead, s.isInt))) return rc;
    auto fi = [&](const char *n) { return (int *)sih_find_field(c, n)->d; };
    // connectivity in internal numbering
    {
        std::vector<int> eoc((size_t)M * c->padC, nE), coe((size_t)2 * c->padE, nC), voe((size_t)2 * c->padE, nV),
            eov((size_t)D * c->padV, nE);
        for (int i = 0; i < nC; i++) {
            int h = c->cellPerm[i];
            for (int s = 0; s < M; s++) {
                int e = m->edgesOnCell[(size_t)h * M + s];
                eoc[(size_t)s * c->padC + i] = (e >= 1 && e <= nE) ? c->edgeInv[e - 1] : nE;
            }
        }
        for (int i = 0; i < nE; i++) {
            int h = c->edgePerm[i];
            for (int k = 0; k < 2; k++) {
                int cc = m->cellsOnEdge[(size_t)h * 2 + k], vv = m->verticesOnEdge[(size_t)h * 2 + k];
                coe[(size_t)k * c->padE + i] = (cc >= 1 && cc <= nC) ? c->cellInv[cc - 1] : nC;
                voe[(size_t)k * c->padE + i] = c->vertInv[vv - 1];
            }
        }
        for (int i = 0; i < nV; i++) {
            int h = c->vertPerm[i];
            for (int k = 0; k < D; k++) {
                int e = m->edgesOnVertex[(size_t)h * D + k];
                eov[(size_t)k * c->padV + i] = (e >= 1 && e <= nE) ? c->edgeInv[e - 1] : nE;
            }
        }
        if ((rc = uploadPlaneInts(c, fi("_eoc"), eoc))) return rc;
        if ((rc = uploadPlaneInts(c, fi("_coe"), coe))) return rc;
        if ((rc = uploadPlaneInts(c, fi("_voe"), voe))) return rc;
        if ((rc = uploadPlaneInts(c, fi("_eov"), eov))) return rc;
    }
    if ((rc = sih_field_put(c, "xEdge", m->xEdge, nE))) return rc;
    if ((rc = sih_field_put(c, "yEdge", m->yEdge, nE))) return rc;
    if ((rc = sih_field_put(c, "zEdge", m->zEdge, nE))) return rc;
    if ((rc = sih_field_put(c, "dcEdge", m->dcEdge, nE))) return rc;
    if ((rc = sih_field_put(c, "dvEdge", m->dvEdge, nE))) return rc;
    return SIH_OK;
}

extern "C" int sih_ir_init(sih_ctx *c, const sih_ir_desc *m)
{
    if (!c || !m) {
        sih_set_error("sih_ir_init: null argument");
        return SIH_ERR_ARG;
    }
    if (!m->edgesOnCell || !m->cellsOnEdge || !m->verticesOnEdge || !m->edgesOnVertex || !m->xEdge || !m->yEdge ||
        !m->zEdge || !m->dcEdge || !m->dvEdge || !m->coeffsReconstruct || m->nCategories < 1 || c->nEdges < 1) {
        sih_set_error("sih_ir_init: incomplete edge description of the mesh");
        return SIH_ERR_ARG;
    }
    if (c->D != 3 && c->D != 4) {
        sih_set_error("incremental remap: vertexDegree must be 3 (hexagonal meshes) or 4 (quadrilateral meshes)");
        return SIH_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(c->device));
    const int nC = c->nCells, nV = c->nVertices, nE = c->nEdges, M = c->M, D = c->D;
    const int NR = D == 4 ? 6 : 4;  // side edges / cells around an edge, :1587-1591
    int rc;
    if ((rc = sih_setup_edges(c, m, "sih_ir_init"))) return rc;
    if (c->ir) sih_ir_destroy(c);
    sih_ir_state *ir = new sih_ir_state();
    c->ir = ir;
    ir->nCat = m->nCategories;
    {
        // per block of 256 owned cells: the run of IR_UPD_WIN consecutive edges that holds most of its cells' edges
        const int nB = std::max(1, (c->nCellsSolve + IR_UPD_BLOCK - 1) / IR_UPD_BLOCK);
        std::vector<int> winStart(nB, 0);
        for (int b = 0; b < nB; b++) {
            std::vector<int> es;
            for (int i = b * IR_UPD_BLOCK; i < std::min(c->nCellsSolve, (b + 1) * IR_UPD_BLOCK); i++) {
                const int h = c->cellPerm[i];
                for (int sIdx = 0; sIdx < c->hNe[i]; sIdx++) {
                    const int e = m->edgesOnCell[(size_t)h * M + sIdx];
                    if (e >= 1 && e <= nE) es.push_back(c->edgeInv[e - 1]);
                }
            }
            std::sort(es.begin(), es.end());
            size_t best = 0, lo = 0;
            int bestStart = es.empty() ? 0 : es[0];
            for (size_t hi = 0; hi < es.size(); hi++) {
                while (es[hi] - es[lo] >= IR_UPD_WIN) lo++;
                if (hi - lo + 1 > best) {
                    best = hi - lo + 1;
                    bestStart = es[lo];
                }
            }
            winStart[b] = bestStart;
        }
        HIPCHK(hipMalloc((void **)&ir->d_updWin, sizeof(int) * nB));
        HIPCHK(hipMemcpy(ir->d_updWin, winStart.data(), sizeof(int) * nB, hipMemcpyHostToDevice));
    }

    struct Spec {
        const char *name;
        SihLoc loc;
        int lead;
        bool isInt;
    };
    const Spec specs[] = {
        {"_eoc", LOC_CELL, M, true}, {"_coe", LOC_EDGE, 2, true}, {"_voe", LOC_EDGE, 2, true},
        {"_eov", LOC_VERTEX, D, true}, {"remapEdge", LOC_EDGE, 1, true}, {"_eoer", LOC_EDGE, NR, true},
        {"_coer", LOC_EDGE, NR, true}, {"_triSlot", LOC_EDGE, 2 * NR, true}, {"_g2c", LOC_CELL, 6, false},
        {"xVertexOnCell", LOC_CELL, M, false}, {"yVertexOnCell", LOC_CELL, M, false},
        {"xVertexOnEdge", LOC_EDGE, NR + 2, false}, {"yVertexOnEdge", LOC_EDGE, NR + 2, false},
        {"geomAvgCell", LOC_CELL, 14, false}, {"minLengthEdgesOnVertex", LOC_VERTEX, 1, false},
        {"xEdge", LOC_EDGE, 1, false}, {"yEdge", LOC_EDGE, 1, false}, {"zEdge", LOC_EDGE, 1, false},
        {"dcEdge", LOC_EDGE, 1, false}, {"dvEdge", LOC_EDGE, 1, false},
        {"coeffs_reconstruct", LOC_CELL, 3 * M, false},
        {"departurePoint", LOC_VERTEX, 2, false}, {"_triCount", LOC_EDGE, 1, true},
        {"_triCell", LOC_EDGE, NTRI, true}, {"xTriangle", LOC_EDGE, 3 * NTRI, false},
        {"yTriangle", LOC_EDGE, 3 * NTRI, false}, {"triangleArea", LOC_EDGE, NTRI, false},
        {"maskCell", LOC_CELL, 1, true},
    };
    for (const Spec &s : specs)
        if ((rc = sih_add_field(c, s.name, s.loc, s.lead, s.isInt))) return rc;
    auto fi = [&](const char *n) { return (int *)sih_find_field(c, n)->d; };
    auto fd = [&](const char *n) { return (double *)sih_find_field(c, n)->d; };

    if ((rc = sih_field_put(c, "coeffs_reconstruct", m->coeffsReconstruct, (size_t)3 * M * nC))) return rc;

    IrDev &d = ir->d;
    d.nCells = nC;
    d.nCellsSolve = c->nCellsSolve;
    d.nVertices = nV;
    d.nEdges = nE;
    d.M = M;
    d.vdeg = D;
    d.nRemap = NR;
    d.nCat = ir->nCat;
    d.padC = c->padC;
    d.padV = c->padV;
    d.padE = c->padE;
    d.onSphere = c->onASphere;
    d.rotate = m->rotateCartesianGrid;
    d.nEdgesOnCell = fi("_nEdgesOnCell");
    d.voc = fi("_voc");
    d.coc = fi("_coc");
    d.eoc = fi("_eoc");
    d.coe = fi("_coe");
    d.voe = fi("_voe");
    d.eov = fi("_eov");
    d.remapEdge = fi("remapEdge");
    d.eoer = fi("_eoer");
    d.coer = fi("_coer");
    d.triSlot = fi("_triSlot");
    d.g2c = fd("_g2c");
    d.xVOC = fd("xVertexOnCell");
    d.yVOC = fd("yVertexOnCell");
    d.xVOE = fd("xVertexOnEdge");
    d.yVOE = fd("yVertexOnEdge");
    d.geom = fd("geomAvgCell");
    d.minLen = fd("minLengthEdgesOnVertex");
    d.dcEdge = fd("dcEdge");
    d.dvEdge = fd("dvEdge");
    d.coeffs = fd("coeffs_reconstruct");
    d.areaCell = fd("areaCell");
    d.u = fd("uVelocity");
    d.v = fd("vVelocity");
    d.dp = fd("departurePoint");
    d.triCount = fi("_triCount");
    d.triCell = fi("_triCell");
    d.xTri = fd("xTriangle");
    d.yTri = fd("yTriangle");
    d.triArea = fd("triangleArea");
    d.maskCell = fi("maskCell");
    d.updWin = ir->d_updWin;
    d.flag = c->d_flag;

    IrInitArgs a{fd("xCell"), fd("yCell"), fd("zCell"), fd("xVertex"), fd("yVertex"), fd("zVertex"),
                 fd("xEdge"), fd("yEdge"),  fd("zEdge")};
    dim3 blk(256);
    hipLaunchKernelGGL(k_ir_zero_flags, dim3(1), dim3(64), 0, c->stream, c->d_flag);
    hipLaunchKernelGGL(k_ir_init_cells, dim3((nC + 255) / 256), blk, 0, c->stream, d, a);
    hipLaunchKernelGGL(k_ir_init_edges, dim3((nE + 255) / 256), blk, 0, c->stream, d, a);
    hipLaunchKernelGGL(k_ir_init_edge_vertices, dim3((nE + 255) / 256), blk, 0, c->stream, d, a);
    hipLaunchKernelGGL(k_ir_init_vertices, dim3((nV + 255) / 256), blk, 0, c->stream, d, a);
    hipLaunchKernelGGL(k_ir_init_geom, dim3((nC + 255) / 256), blk, 0, c->stream, d);
    hipLaunchKernelGGL(k_ir_init_tri_slots, dim3((nE + 255) / 256), blk, 0, c->stream, d);
    HIPCHK(hipGetLastError());
    int flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (flag & FLAG_CCW) {
        sih_set_error("IR requires convex cells with vertices arranged CCW around the cell");
        return SIH_ERR_ARG;
    }
    if (flag & FLAG_HALFPLANE) {
        sih_set_error("IR: cell 1 is not in the left half-plane of an edge");
        return SIH_ERR_ARG;
    }
    if (flag & FLAG_AREA) {
        sih_set_error("IR: Bad cell area");
        return SIH_ERR_ARG;
    }
    for (hipEvent_t &e : ir->ev) HIPCHK(hipEventCreate(&e));
    ir->inited = true;
    return SIH_OK;
}

extern "C" int sih_ir_set_tracers(sih_ctx *c, int nTracers, const char *const *names, const char *const *parents,
                                  const int32_t *nLayers)
{
    if (!c || !c->ir || !c->ir->inited) {
        sih_set_error("sih_ir_set_tracers: call sih_ir_init first");
        return SIH_ERR_STATE;
    }
    if (nTracers < 1 || !names || !parents || !nLayers) {
        sih_set_error("IR tracers: No active tracers for IR");
        return SIH_ERR_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    sih_ir_state *ir = c->ir;
    std::vector<IrTracer> list(nTracers);
    for (int i = 0; i < nTracers; i++) {
        IrTracer &t = list[i];
        t.name = names[i];
        t.L = nLayers[i];
        if (t.L < 1) {
            sih_set_error("IR tracers: nLayers must be positive");
            return SIH_ERR_ARG;
        }
        std::string p = parents[i] ? parents[i] : "None";
        if (i == 0) {
            if (p != "None" || t.L != 1) {
                sih_set_error("IR: mass-like field must come first, have no parent and be 2D");
                return SIH_ERR_ARG;
            }
        } else {
            for (int j = 0; j < i; j++)
                if (list[j].name == p) t.parent = j;
            if (t.parent < 0) {
                sih_set_error("IR tracers: parent must precede its child in the list: " + t.name);
                return SIH_ERR_ARG;
            }
            IrTracer &pt = list[t.parent];
            pt.hasChild = true;
            t.nParents = pt.nParents + 1;
            if (t.nParents > 3) {
                sih_set_error("Tracer has > 3 parents: " + t.name);
                return SIH_ERR_ARG;
            }
            if (pt.L != 1 && pt.L != t.L) {
                sih_set_error("IR tracers: a layered parent must have the layer count of its child: " + t.name);
                return SIH_ERR_ARG;
            }
        }
    }
    for (IrTracer &t : list)
        if (t.hasChild && t.nParents == 3) {
            sih_set_error("IR, construct_linear_tracer_fields: Tracers with 3 parents should not have children");
            return SIH_ERR_ARG;
        }
    int rc;
    int maxLead = 0;
    for (IrTracer &t : list) {
        const int lead = t.L * ir->nCat;
        maxLead = std::max(maxLead, lead);
        if ((rc = sih_add_field(c, t.name.c_str(), LOC_CELL, lead, false))) return rc;
        t.val = (double *)sih_find_field(c, t.name.c_str())->d;
        struct {
            const char *suffix;
            double **p;
            SihLoc loc;
            bool need;
        } sc[] = {{"Product", &t.prod, LOC_CELL, true},
                  {"Barycenterx", &t.xB, LOC_CELL, t.hasChild}, {"Barycentery", &t.yB, LOC_CELL, t.hasChild},
                  {"EdgeFlux", &t.flux, LOC_EDGE, true}};
        for (auto &s : sc) {
            if (!s.need) continue;
            std::string fn = t.name + s.suffix;
            if ((rc = sih_add_field(c, fn.c_str(), s.loc, lead, false))) return rc;
            *s.p = (double *)sih_find_field(c, fn.c_str())->d;
        }
        {  // the records: not a plane field, only its storage comes from the field table
            std::string fn = "_" + t.name + "Records";
            if ((rc = sih_add_field(c, fn.c_str(), LOC_CELL, 3 * lead, false))) return rc;
            t.rec = (double *)sih_find_field(c, fn.c_str())->d;
        }
        t.sumInit.assign(lead, 0.0);
        t.sumFinal.assign(lead, 0.0);
    }
    ir->tracers = list;
    ir->iVol = ir->iSnow = -1;
    for (int i = 0; i < nTracers; i++) {
        if (list[i].name == "iceVolumeCategory") ir->iVol = i;
        if (list[i].name == "snowVolumeCategory") ir->iSnow = i;
    }
    std::vector<IrTabEntry> tab(nTracers);
    for (int i = 0; i < nTracers; i++) {
        const IrTracer &t = list[i];
        IrTabEntry &q = tab[i];
        q = IrTabEntry{(IrG *)t.val, (IrG *)t.rec, (IrG *)t.xB, (IrG *)t.yB, (IrG *)t.prod, (IrG *)t.flux, t.L, t.parent, t.nParents + 1,
                       t.hasChild ? 1 : 0, {0, 0, 0, 0}, t.L};
        int j = i;
        for (int k = t.nParents; k >= 0; k--) {
            q.chain[k] = j;
            j = list[j].parent;
        }
    }
    if (ir->d_tab) HIPCHK(hipFree(ir->d_tab));
    HIPCHK(hipMalloc((void **)&ir->d_tab, sizeof(IrTabEntry) * nTracers));
    HIPCHK(hipMemcpy(ir->d_tab, tab.data(), sizeof(IrTabEntry) * nTracers, hipMemcpyHostToDevice));
    {
        std::vector<IrPlane> pls;
        for (int i = 0; i < nTracers; i++)
            for (int cat = 0; cat < ir->nCat; cat++)
                for (int ly = 0; ly < list[i].L; ly += IR_LCHUNK) pls.push_back(IrPlane{i, ly, cat});
        if (ir->d_planes) HIPCHK(hipFree(ir->d_planes));
        HIPCHK(hipMalloc((void **)&ir->d_planes, sizeof(IrPlane) * pls.size()));
        HIPCHK(hipMemcpy(ir->d_planes, pls.data(), sizeof(IrPlane) * pls.size(), hipMemcpyHostToDevice));
        ir->nPlanes = (int)pls.size();
        // the tracer tree depth first, siblings in list order (for k_ir_flux_walk)
        std::vector<IrStep> steps;
        int slots = 0;
        bool ok = true;
        std::vector<std::pair<int, int>> stack;  // (tracer, level), visited last-in first-out
        for (int i = nTracers - 1; i >= 0; i--)
            if (list[i].parent < 0) stack.push_back({i, 0});
        while (!stack.empty()) {
            const int i = stack.back().first, lev = stack.back().second;
            stack.pop_back();
            if (lev > 3 || (list[i].hasChild && (list[i].L != 1 || lev > 2))) ok = false;
            steps.push_back(IrStep{i, lev, list[i].hasChild ? 1 : 0, slots});
            slots += list[i].L;
            for (int j = nTracers - 1; j >= 0; j--)
                if (list[j].parent == i) stack.push_back({j, lev + 1});
        }
        if ((int)steps.size() != nTracers || slots > IR_WALK_MAXSLOTS || steps.empty() || steps[0].ti != 0) ok = false;
        if (ir->d_steps) HIPCHK(hipFree(ir->d_steps));
        ir->d_steps = nullptr;
        HIPCHK(hipMalloc((void **)&ir->d_steps, sizeof(IrStep) * steps.size()));
        HIPCHK(hipMemcpy(ir->d_steps, steps.data(), sizeof(IrStep) * steps.size(), hipMemcpyHostToDevice));
        ir->nSteps = (int)steps.size();
        ir->nSlots = slots;
        ir->walkOk = ok;
        const char *fm = getenv("SIH_IR_FLUX_MODE");
        ir->fluxMode = fm ? atoi(fm) : 2;
    }
    if (ir->d_sums) HIPCHK(hipFree(ir->d_sums));
    HIPCHK(hipMalloc((void **)&ir->d_sums, sizeof(double) * maxLead));
    return SIH_OK;
}

template <int M>
static void launchCellKernels(sih_ir_state *ir, hipStream_t s, bool update)
{
    IrDev &d = ir->d;
    const int nT = (int)ir->tracers.size();
    if (!update)
        hipLaunchKernelGGL(k_ir_reconstruct<M>, dim3(xcdGrid((d.nCells + 255) / 256, d.nCat)), dim3(256), 0, s, d,
                           ir->d_tab, nT);
    else
        hipLaunchKernelGGL(k_ir_update<M>, dim3((d.nCellsSolve + IR_UPD_BLOCK - 1) / IR_UPD_BLOCK, d.nCat), dim3(IR_UPD_BLOCK), 0, s, d, ir->d_tab,
                           nT);
}

static void launchCells(sih_ir_state *ir, hipStream_t s, bool update)
{
    const int M = ir->d.M;
    if (M <= 6)
        launchCellKernels<6>(ir, s, update);
    else if (M <= 8)
        launchCellKernels<8>(ir, s, update);
    else
        launchCellKernels<SIH_MAXE>(ir, s, update);
}

static int irSums(sih_ctx *c, bool init)
{
    sih_ir_state *ir = c->ir;
    for (size_t i = 0; i < ir->tracers.size(); i++) {
        IrTracer &t = ir->tracers[i];
        const int lead = t.L * ir->nCat;
        hipLaunchKernelGGL(k_ir_sum, dim3(lead), dim3(256), 0, c->stream, ir->d, ir->d_tab, (int)i, ir->d_sums);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync((init ? t.sumInit : t.sumFinal).data(), ir->d_sums, sizeof(double) * lead,
                              hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return SIH_OK;
}

// fields of the monotonicity check, created when it is first asked for: <tracer>LocalMin / LocalMax (the
// reference's tracer_monotonicity pool; a multi-block host exchanges their halos between sih_ir_run and
// sih_ir_check_monotonicity), the 1e-11 mask of the old values, and the sweep buffers
static int monoSetup(sih_ctx *c)
{
    sih_ir_state *ir = c->ir;
    if (ir->monoReady) return SIH_OK;
    int rc, maxLead = 0;
    for (IrTracer &t : ir->tracers) {
        if (t.parent < 0) continue;
        const int lead = t.L * ir->nCat;
        maxLead = std::max(maxLead, lead);
        const std::string a = t.name + "LocalMin", b = t.name + "LocalMax", m = "_" + t.name + "Mask11";
        if ((rc = sih_add_field(c, a.c_str(), LOC_CELL, lead, false))) return rc;
        if ((rc = sih_add_field(c, b.c_str(), LOC_CELL, lead, false))) return rc;
        if ((rc = sih_add_field(c, m.c_str(), LOC_CELL, lead, true))) return rc;
        t.lmin = (double *)sih_find_field(c, a.c_str())->d;
        t.lmax = (double *)sih_find_field(c, b.c_str())->d;
        t.mask11 = (int *)sih_find_field(c, m.c_str())->d;
    }
    for (int k = 0; k < 4 && maxLead > 0; k++) {
        const std::string n = "_monoSweep" + std::to_string(k);
        if ((rc = sih_add_field(c, n.c_str(), LOC_CELL, maxLead, false))) return rc;
        ir->monoTmp[k] = (double *)sih_find_field(c, n.c_str())->d;
    }
    ir->monoReady = true;
    return SIH_OK;
}

// thickness -> volume :2608-2650
static void irToVolume(sih_ctx *c)
{
    sih_ir_state *ir = c->ir;
    const IrDev &d = ir->d;
    for (int i : {ir->iVol, ir->iSnow})
        if (i >= 0)
            hipLaunchKernelGGL(k_ir_convert, dim3((d.nCells + 255) / 256, d.nCat), dim3(256), 0, c->stream, d,
                               ir->tracers[0].val, ir->tracers[i].val, 0);
}

static int irRun(sih_ctx *c, double dt, int checks, bool timed, double *totalMs, double *phaseMs)
{
    if (!c || !c->ir || !c->ir->inited || c->ir->tracers.empty()) {
        sih_set_error("sih_ir_run: call sih_ir_init and sih_ir_set_tracers first");
        return SIH_ERR_STATE;
    }
    if (c->ir->monoPending) {
        sih_set_error("sih_ir_run: the previous step asked for the monotonicity check; call sih_ir_check_monotonicity");
        return SIH_ERR_STATE;
    }
    const int checkConservation = checks & 1;
    const bool monotonicity = (checks & 2) != 0;
    HIPCHK(hipSetDevice(c->device));
    sih_ir_state *ir = c->ir;
    IrDev &d = ir->d;
    const int nT = (int)ir->tracers.size();
    const dim3 blk(256);
    const unsigned gC = (d.nCells + 255) / 256, gCs = (d.nCellsSolve + 255) / 256, gE = (d.nEdges + 255) / 256,
                   gV = (d.nVertices + 255) / 256;
    hipStream_t s = c->stream;
    auto mark = [&](int i) {
        if (timed) (void)hipEventRecord(ir->ev[i], s);
    };
    mark(0);
    hipLaunchKernelGGL(k_ir_zero_flags, dim3(1), dim3(64), 0, s, c->d_flag);
    // volume -> thickness :2502-2544, masks :3023-3029
    const double *area = ir->tracers[0].val;
    for (int i : {ir->iVol, ir->iSnow})
        if (i >= 0)
            hipLaunchKernelGGL(k_ir_convert, dim3(gC, d.nCat), blk, 0, s, d, area, ir->tracers[i].val, 1);
    if (monotonicity) {
        int rcm = monoSetup(c);
        if (rcm) return rcm;
        for (IrTracer &t : ir->tracers)
            if (t.parent >= 0)
                hipLaunchKernelGGL(k_ir_mono_init, dim3(gC, t.L * d.nCat), blk, 0, s, d, t.val,
                                   ir->tracers[t.parent].val, t.L, ir->tracers[t.parent].L, t.lmin, t.lmax, t.mask11);
    }
    hipLaunchKernelGGL(k_ir_mask_cell, dim3(gC), blk, 0, s, d, area);
    mark(1);
    // construct_linear_tracer_fields :3580-4189, compute_mass_tracer_products :6982
    launchCells(ir, s, false);
    if (checks & SIH_IR_CHECK_RECOVER_MEANS) {
        for (size_t i = 0; i < ir->tracers.size(); i++) {
            IrTracer &t = ir->tracers[i];
            if (!t.recovered) {
                const std::string fn = t.name + "RecoveredProduct";
                int rc0 = sih_add_field(c, fn.c_str(), LOC_CELL, t.L * d.nCat, false);
                if (rc0) return rc0;
                t.recovered = (double *)sih_find_field(c, fn.c_str())->d;
            }
            const dim3 g(gC, d.nCat);
            if (d.M <= 6)
                hipLaunchKernelGGL(k_ir_recover_means<6>, g, blk, 0, s, d, ir->d_tab, (int)i, t.recovered, c->d_flag + 3);
            else if (d.M <= 8)
                hipLaunchKernelGGL(k_ir_recover_means<8>, g, blk, 0, s, d, ir->d_tab, (int)i, t.recovered, c->d_flag + 3);
            else
                hipLaunchKernelGGL(k_ir_recover_means<SIH_MAXE>, g, blk, 0, s, d, ir->d_tab, (int)i, t.recovered,
                                   c->d_flag + 3);
        }
    }
    mark(2);
    hipLaunchKernelGGL(k_ir_departure_points, dim3(gV), blk, 0, s, d, dt);
    if (d.vdeg == 4)
        hipLaunchKernelGGL(k_ir_triangles<4>, dim3((d.nEdges + 127) / 128), dim3(128), 0, s, d);
    else
        hipLaunchKernelGGL(k_ir_triangles<3>, dim3((d.nEdges + 127) / 128), dim3(128), 0, s, d);
    mark(3);
    if (ir->fluxMode >= 2 && ir->walkOk)
        hipLaunchKernelGGL(k_ir_flux_walk, dim3(xcdGrid(gE, d.nCat)), blk, sizeof(double) * 256 * ir->nSlots, s, d,
                           ir->d_tab, ir->d_steps, ir->nSteps, ir->nSlots);
    else if (ir->fluxMode >= 1)
        hipLaunchKernelGGL(k_ir_flux_planes, dim3(xcdGrid(gE, ir->nPlanes)), blk, 0, s, d, ir->d_tab, ir->d_planes,
                           ir->nPlanes);
    else
        hipLaunchKernelGGL(k_ir_flux, dim3(gE, d.nCat), blk, 0, s, d, ir->d_tab, nT);
    mark(4);
    HIPCHK(hipGetLastError());
    int rc;
    if (checkConservation && (rc = irSums(c, true))) return rc;
    launchCells(ir, s, true);
    if (checkConservation && (rc = irSums(c, false))) return rc;
    hipLaunchKernelGGL(k_ir_zap, dim3(gCs, d.nCat), blk, 0, s, d, ir->d_tab, nT);
    // thickness -> volume :2608-2650 (after the monotonicity check, which compares thicknesses)
    ir->monoPending = monotonicity;
    if (!monotonicity) irToVolume(c);
    mark(5);
    HIPCHK(hipGetLastError());
    int flags[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(flags, c->d_flag, sizeof(flags), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    ir->cflWarnings = flags[1];
    ir->recoverWarnings = flags[3];
    if (timed) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ir->ev[0], ir->ev[5]));
        if (totalMs) *totalMs = ms;
        for (int i = 0; i < 5 && phaseMs; i++) {
            HIPCHK(hipEventElapsedTime(&ms, ir->ev[i], ir->ev[i + 1]));
            phaseMs[i] = ms;
        }
    }
    if (flags[0] & FLAG_PARALLEL) {
        sih_set_error("IR: basis vectors for IR coordinate transformations must not be parallel");
        return SIH_ERR_IR;
    }
    if (flags[0] & FLAG_NEGAREA) {
        sih_set_error("MPAS-seaice: IR negative reconstructed ice area");
        return SIH_ERR_IR;
    }
    if (flags[0] & FLAG_NEGMASS) {
        sih_set_error("IR: Negative mass in IR");
        return SIH_ERR_IR;
    }
    return SIH_OK;
}

extern "C" int sih_ir_run(sih_ctx *c, double dt, int checks)
{
    return irRun(c, dt, checks, false, nullptr, nullptr);
}

// check_tracer_monotonicity :8416-8749 and the thickness -> volume conversion that follows it
extern "C" int sih_ir_check_monotonicity(sih_ctx *c)
{
    if (!c || !c->ir || !c->ir->monoPending) {
        sih_set_error("sih_ir_check_monotonicity: no step with the monotonicity check is pending");
        return SIH_ERR_STATE;
    }
    HIPCHK(hipSetDevice(c->device));
    sih_ir_state *ir = c->ir;
    const IrDev &d = ir->d;
    hipStream_t s = c->stream;
    const dim3 blk(256);
    const unsigned gC = (d.nCells + 255) / 256, gCs = (d.nCellsSolve + 255) / 256;
    hipLaunchKernelGGL(k_ir_zero_flags, dim3(1), dim3(64), 0, s, c->d_flag);
    std::string failed;
    for (IrTracer &t : ir->tracers) {
        if (t.parent < 0) continue;  // monotonicity holds for tracers, not for the mass-like field
        const int lead = t.L * d.nCat;
        const size_t bytes = sizeof(double) * (size_t)lead * d.padC;
        // sweep 0 reads the unextended values
        HIPCHK(hipMemcpyAsync(ir->monoTmp[0], t.lmin, bytes, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(ir->monoTmp[1], t.lmax, bytes, hipMemcpyDeviceToDevice, s));
        int cur = 0, sweeps = 0;
        for (;;) {
            double *inMin = ir->monoTmp[2 * cur], *inMax = ir->monoTmp[2 * cur + 1];
            double *outMin = ir->monoTmp[2 * (1 - cur)], *outMax = ir->monoTmp[2 * (1 - cur) + 1];
            hipLaunchKernelGGL(k_ir_zero_one, dim3(1), dim3(64), 0, s, c->d_flag + 2);
            hipLaunchKernelGGL(k_ir_mono_extend, dim3(gC, lead), blk, 0, s, d, c->d_cellPerm, t.mask11, t.lmin, t.lmax,
                               inMin, inMax, outMin, outMax, c->d_flag + 2);
            int changed = 0;
            HIPCHK(hipMemcpyAsync(&changed, c->d_flag + 2, sizeof(int), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            cur = 1 - cur;
            if (!changed) break;
            if (++sweeps > d.nCells + 1) {  // cannot happen: every sweep settles at least the next cell in order
                sih_set_error("sih_ir_check_monotonicity: extension did not settle");
                return SIH_ERR_IR;
            }
        }
        // the extended values replace the fields, as the reference's in-place loop leaves them
        HIPCHK(hipMemcpyAsync(t.lmin, ir->monoTmp[2 * cur], bytes, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(t.lmax, ir->monoTmp[2 * cur + 1], bytes, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_ir_mono_check, dim3(gCs, lead), blk, 0, s, d, t.val, t.mask11, t.lmin, t.lmax);
        int flag = 0;
        HIPCHK(hipMemcpyAsync(&flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (flag & FLAG_MONO) {
            failed = t.name;
            break;  // the reference returns at the first violation
        }
    }
    irToVolume(c);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    ir->monoPending = false;
    if (!failed.empty()) {
        sih_set_error("IR advection, monotonicity violation: tracer = " + failed);
        return SIH_ERR_IR;
    }
    return SIH_OK;
}

extern "C" int sih_ir_run_timed(sih_ctx *c, double dt, double *totalMs, double *phaseMs5)
{
    return irRun(c, dt, 0, true, totalMs, phaseMs5);
}

// records -> planes, for inspection
__global__ void k_ir_unpack_records(IrDev d, const double *rec, int L, int S, double *center, double *gx, double *gy)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nCells) return;
    const size_t pc = d.padC;
    for (int cat = 0; cat < d.nCat; cat++)
        for (int layer = 0; layer < L; layer++) {
            const double *r = rec + (((size_t)cat * pc + c) * S + layer) * 3;
            const size_t k = (size_t)(cat * L + layer) * pc + c;
            center[k] = r[0];
            gx[k] = r[1];
            gy[k] = r[2];
        }
}

extern "C" int sih_ir_unpack_reconstruction(sih_ctx *c)
{
    if (!c || !c->ir || c->ir->tracers.empty()) {
        sih_set_error("sih_ir_unpack_reconstruction: sih_ir_set_tracers has not run");
        return SIH_ERR_STATE;
    }
    HIPCHK(hipSetDevice(c->device));
    sih_ir_state *ir = c->ir;
    int rc;
    for (IrTracer &t : ir->tracers) {
        const int lead = t.L * ir->nCat;
        double **dst[3] = {&t.center, &t.gx, &t.gy};
        const char *suffix[3] = {"Center", "Gradx", "Grady"};
        for (int k = 0; k < 3; k++) {
            const std::string fn = t.name + suffix[k];
            if ((rc = sih_add_field(c, fn.c_str(), LOC_CELL, lead, false))) return rc;
            *dst[k] = (double *)sih_find_field(c, fn.c_str())->d;
        }
        hipLaunchKernelGGL(k_ir_unpack_records, dim3((ir->d.nCells + 255) / 256), dim3(256), 0, c->stream, ir->d, t.rec, t.L,
                           t.L, t.center, t.gx, t.gy);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return SIH_OK;
}

extern "C" int sih_ir_get_sums(sih_ctx *c, const char *tracer, double *initSums, double *finalSums)
{
    if (!c || !c->ir || !tracer || !initSums || !finalSums) {
        sih_set_error("sih_ir_get_sums: bad argument");
        return SIH_ERR_ARG;
    }
    for (IrTracer &t : c->ir->tracers)
        if (t.name == tracer) {
            memcpy(initSums, t.sumInit.data(), sizeof(double) * t.sumInit.size());
            memcpy(finalSums, t.sumFinal.data(), sizeof(double) * t.sumFinal.size());
            return SIH_OK;
        }
    sih_set_error(std::string("sih_ir_get_sums: unknown tracer ") + tracer);
    return SIH_ERR_ARG;
}

int sih_ir_cfl_warnings(sih_ctx *c) { return c && c->ir ? c->ir->cflWarnings : 0; }
int sih_ir_recover_warnings(sih_ctx *c) { return c && c->ir ? c->ir->recoverWarnings : 0; }
int sih_ir_flux_walk(sih_ctx *c) { return c && c->ir && c->ir->fluxMode == 2 && c->ir->walkOk ? 1 : 0; }

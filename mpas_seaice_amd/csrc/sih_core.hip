// Context, device field registry, layout conversion and halo pack/unpack of libseaice_hip.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>

#define SIH_DEBUG_ALLOC_IMPL
#include "sih_internal.h"

#ifdef SIH_DEBUG_ALLOC
#include <mutex>
static std::map<char *, std::pair<size_t, std::string>> g_dbgLive;
static std::mutex g_dbgMutex;
hipError_t sih_dbg_malloc(void **p, size_t n, const char *file, int line)
{
    hipError_t e = hipMalloc(p, n);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_dbgMutex);
    char *b = (char *)*p;
    for (auto &kv : g_dbgLive)
        if (b < kv.first + kv.second.first && kv.first < b + n)
            fprintf(stderr, "SIH_DEBUG_ALLOC: %s:%d got %p+%zu overlapping live %p+%zu from %s\n", file, line, (void *)b, n,
                    (void *)kv.first, kv.second.first, kv.second.second.c_str());
    g_dbgLive[b] = {n, std::string(file) + ":" + std::to_string(line)};
    return e;
}
hipError_t sih_dbg_free(void *p, const char *file, int line)
{
    if (p) {
        std::lock_guard<std::mutex> lock(g_dbgMutex);
        auto it = g_dbgLive.find((char *)p);
        if (it == g_dbgLive.end())
            fprintf(stderr, "SIH_DEBUG_ALLOC: %s:%d frees %p which is not live (double free or foreign pointer)\n", file, line, p);
        else
            g_dbgLive.erase(it);
    }
    return hipFree(p);
}
#define hipMalloc(p, n) sih_dbg_malloc((void **)(p), (n), __FILE__, __LINE__)
#define hipFree(p) sih_dbg_free((void *)(p), __FILE__, __LINE__)
#endif

static thread_local std::string g_err;

void sih_set_error(const std::string &s) { g_err = s; }

int sih_hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    g_err = buf;
    return SIH_ERR_HIP;
}

extern "C" const char *sih_last_error(void) { return g_err.c_str(); }

static int roundUp(int n, int m) { return ((n + m - 1) / m) * m; }

// ------------------------------------------------------------------------------------------
// layout conversion kernels: host layout (lead fastest, entity slowest, host numbering)
//   <-> device SoA (entity fastest, internal numbering)
template <typename T>
__global__ void k_put(T *__restrict__ dst, const T *__restrict__ src, const int *__restrict__ perm, int n, int L,
                      int pad)
{
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)n * L) return;
    int l = (int)(t / n), i = (int)(t % n);
    int h = perm ? perm[i] : i;
    dst[(size_t)l * pad + i] = src[(size_t)h * L + l];
}

template <typename T>
__global__ void k_get(T *__restrict__ dst, const T *__restrict__ src, const int *__restrict__ perm, int n, int L,
                      int pad)
{
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)n * L) return;
    int l = (int)(t / n), i = (int)(t % n);
    int h = perm ? perm[i] : i;
    dst[(size_t)h * L + l] = src[(size_t)l * pad + i];
}

// ------------------------------------------------------------------------------------------
static int locCount(const sih_ctx *c, SihLoc loc)
{
    return loc == LOC_CELL ? c->nCells : loc == LOC_VERTEX ? c->nVertices : c->nEdges;
}
static int locPad(const sih_ctx *c, SihLoc loc) { return loc == LOC_CELL ? c->padC : loc == LOC_VERTEX ? c->padV : c->padE; }
static const int *locPerm(const sih_ctx *c, SihLoc loc)
{
    return loc == LOC_CELL ? c->d_cellPerm : loc == LOC_VERTEX ? c->d_vertPerm : c->d_edgePerm;
}

// Device memory of a context comes from a few large chunks (one hipMalloc each) that are carved up
// here: hundreds of megabyte-sized arrays allocated one by one end up on small page fragments, and the
// gather kernels then pay for translation misses.  SIH_ARENA_MB sets the chunk size (0: one hipMalloc per
// array), SIH_ALLOC_ALIGN the alignment of the pieces in bytes.
static int devAlloc(sih_ctx *c, void **p, size_t bytes)
{
    static const size_t chunkMB = getenv("SIH_ARENA_MB") ? (size_t)atol(getenv("SIH_ARENA_MB")) : 512;
    static const size_t align = getenv("SIH_ALLOC_ALIGN") ? (size_t)atol(getenv("SIH_ALLOC_ALIGN")) : 65536;
    if (chunkMB == 0) {
        HIPCHK(hipMalloc(p, bytes));
        HIPCHK(hipMemsetAsync(*p, 0, bytes, c->stream));
        c->allocs.push_back(*p);
        return SIH_OK;
    }
    size_t need = (bytes + align - 1) / align * align;
    if (!c->arenaBase || c->arenaUsed + need > c->arenaSize) {
        size_t sz = std::max(chunkMB << 20, need);
        void *base = nullptr;
        HIPCHK(hipMalloc(&base, sz));
        c->allocs.push_back(base);
        c->arenaBase = (char *)base;
        c->arenaSize = sz;
        c->arenaUsed = 0;
    }
    *p = c->arenaBase + c->arenaUsed;
    c->arenaUsed += need;
    HIPCHK(hipMemsetAsync(*p, 0, bytes, c->stream));
    return SIH_OK;
}

static int addField(sih_ctx *c, const char *name, SihLoc loc, int lead, bool isInt)
{
    SihField f;
    f.name = name;
    f.loc = loc;
    f.lead = lead;
    f.isInt = isInt;
    f.d = nullptr;
    size_t bytes = (size_t)lead * locPad(c, loc) * (isInt ? 4 : 8);
    int rc = devAlloc(c, &f.d, bytes);
    if (rc) return rc;
    c->fields[name] = f;
    return SIH_OK;
}

int sih_dev_alloc(sih_ctx *c, void **p, size_t bytes) { return devAlloc(c, p, bytes); }

int sih_add_field(sih_ctx *c, const char *name, SihLoc loc, int lead, bool isInt)
{
    if (c->fields.count(name)) {
        SihField &f = c->fields[name];
        if (f.loc == loc && f.lead == lead && f.isInt == isInt) return SIH_OK;
        sih_set_error(std::string("field exists with another shape: ") + name);
        return SIH_ERR_STATE;
    }
    return addField(c, name, loc, lead, isInt);
}

SihField *sih_find_field(sih_ctx *c, const char *name)
{
    auto it = c->fields.find(name);
    if (it == c->fields.end()) return nullptr;
    return &it->second;
}

static int ensureStage(sih_ctx *c, size_t bytes)
{
    if (c->stageBytes >= bytes) return SIH_OK;
    if (c->d_stage) {
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipFree(c->d_stage));
        c->d_stage = nullptr;
        c->stageBytes = 0;
    }
    size_t nb = std::max(bytes, (size_t)1 << 20);
    HIPCHK(hipMalloc(&c->d_stage, nb));
    c->stageBytes = nb;
    return SIH_OK;
}

// Morton key of a point in the unit cube
static uint64_t mortonKey(double x, double y, double z)
{
    auto spread = [](uint64_t v) {
        v &= 0x1fffff;
        v = (v | v << 32) & 0x1f00000000ffffULL;
        v = (v | v << 16) & 0x1f0000ff0000ffULL;
        v = (v | v << 8) & 0x100f00f00f00f00fULL;
        v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
        v = (v | v << 2) & 0x1249249249249249ULL;
        return v;
    };
    auto q = [](double t) {
        double s = std::min(std::max(t, 0.0), 1.0) * 2097151.0;
        return (uint64_t)s;
    };
    return spread(q(x)) | (spread(q(y)) << 1) | (spread(q(z)) << 2);
}

extern "C" int sih_block_destroy(sih_ctx *c);
static int resolveScratch(int device);

// contexts alive per device in this process: a kernel that needs the whole device to itself (evp_persist.hip) is
// only chosen while its context is the only one
static std::atomic<int> g_liveCtx[SIH_MAX_DEVICES];
int sih_live_contexts(int device) { return device >= 0 && device < SIH_MAX_DEVICES ? g_liveCtx[device].load() : 2; }

extern "C" int sih_block_create(const sih_mesh_desc *m, int device, sih_ctx **out)
{
    if (!m || !out) {
        sih_set_error("sih_block_create: null argument");
        return SIH_ERR_ARG;
    }
    *out = nullptr;  // the caller's handle is defined on every return
    // vertexDegree 4 (quadrilateral meshes): the velocity solver runs in its two-kernel form, the incremental remap
    // with six side edges per edge; the weak scheme and the upwind advection refuse such a block at their own set-up
    if (m->maxEdges < 3 || m->maxEdges > SIH_MAXE || (m->vertexDegree != 3 && m->vertexDegree != 4)) {
        sih_set_error("sih_block_create: maxEdges must be in [3,10] and vertexDegree 3 or 4");
        return SIH_ERR_UNSUPPORTED;
    }
    if (m->nCells < 1 || m->nVertices < 1 || m->nCellsSolve > m->nCells || m->nVerticesSolve > m->nVertices ||
        m->nCellsSolve < 0 || m->nVerticesSolve < 0 || m->nEdges < 0 || m->nEdgesSolve < 0 ||
        m->nEdgesSolve > m->nEdges || !m->nEdgesOnCell || !m->verticesOnCell || !m->cellsOnCell || !m->cellsOnVertex || !m->xCell || !m->yCell ||
        !m->zCell || !m->xVertex || !m->yVertex || !m->zVertex || !m->areaCell || !m->areaTriangle || !m->fVertex) {
        sih_set_error("sih_block_create: inconsistent mesh description");
        return SIH_ERR_ARG;
    }
    HIPCHK(hipSetDevice(device));
    {
        int rcs = resolveScratch(device);
        if (rcs) return rcs;
    }
    sih_ctx *c = new sih_ctx();
    struct Guard {  // any early return below releases what has been created so far
        sih_ctx *c;
        ~Guard()
        {
            if (c) sih_block_destroy(c);
        }
    } guard{c};
    c->device = device;
    if (device >= 0 && device < SIH_MAX_DEVICES) {
        g_liveCtx[device]++;
        c->counted = true;
    }
    HIPCHK(hipStreamCreate(&c->ownStream));
    c->stream = c->ownStream;
    c->nCells = m->nCells;
    c->nCellsSolve = m->nCellsSolve;
    c->nVertices = m->nVertices;
    c->nVerticesSolve = m->nVerticesSolve;
    c->nEdges = m->nEdges;
    c->nEdgesSolve = m->nEdgesSolve;
    c->M = m->maxEdges;
    c->D = m->vertexDegree;
    c->onASphere = m->onASphere;
    c->R = m->sphereRadius;
    c->padC = roundUp(c->nCells + 1, 64);
    c->padV = roundUp(c->nVertices + 1, 64);
    c->padE = roundUp(std::max(c->nEdges, 0) + 1, 64);
    c->renumber = m->renumber != 0;
    const int nC = c->nCells, nV = c->nVertices, M = c->M, D = c->D;

    // validate connectivity before anything is dereferenced on the device
    for (int i = 0; i < nC; i++) {
        int E = m->nEdgesOnCell[i];
        if (E < 3 || E > M) {
            sih_set_error("sih_block_create: nEdgesOnCell out of range");
            return SIH_ERR_ARG;
        }
        for (int s = 0; s < E; s++) {
            int v = m->verticesOnCell[(size_t)i * M + s];
            int nb = m->cellsOnCell[(size_t)i * M + s];
            if (v < 1 || v > nV + 1 || nb < 0 || nb > nC + 1) {
                sih_set_error("sih_block_create: verticesOnCell / cellsOnCell index out of range");
                    return SIH_ERR_ARG;
            }
        }
    }
    for (size_t k = 0; k < (size_t)nV * D; k++) {
        int cc = m->cellsOnVertex[k];
        if (cc < 0 || cc > nC + 1) {
            sih_set_error("sih_block_create: cellsOnVertex index out of range");
            return SIH_ERR_ARG;
        }
    }
    if (m->dvEdge && m->nEdgesSolve > 0) {
        double mn = 1.0e30;
        for (int e = 0; e < m->nEdgesSolve; e++) mn = std::min(mn, m->dvEdge[e]);
        c->minDvEdge = mn;
    }

    // ---- internal numbering: owned entities first (as in MPAS), each group along a Morton curve;
    //      vertices follow the cell they are first met at.
    c->cellPerm.resize(nC);
    std::iota(c->cellPerm.begin(), c->cellPerm.end(), 0);
    c->vertPerm.resize(nV);
    std::iota(c->vertPerm.begin(), c->vertPerm.end(), 0);
    if (c->renumber) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        const double *xyz[3] = {m->xCell, m->yCell, m->zCell};
        for (int k = 0; k < 3; k++)
            for (int i = 0; i < nC; i++) {
                lo[k] = std::min(lo[k], xyz[k][i]);
                hi[k] = std::max(hi[k], xyz[k][i]);
            }
        std::vector<uint64_t> key(nC);
        for (int i = 0; i < nC; i++) {
            double q[3];
            for (int k = 0; k < 3; k++) q[k] = hi[k] > lo[k] ? (xyz[k][i] - lo[k]) / (hi[k] - lo[k]) : 0.0;
            key[i] = mortonKey(q[0], q[1], q[2]);
        }
        auto cmp = [&](int a, int b) { return key[a] != key[b] ? key[a] < key[b] : a < b; };
        std::sort(c->cellPerm.begin(), c->cellPerm.begin() + c->nCellsSolve, cmp);
        std::sort(c->cellPerm.begin() + c->nCellsSolve, c->cellPerm.end(), cmp);
    }
    c->cellInv.assign(nC + 1, nC);
    for (int i = 0; i < nC; i++) c->cellInv[c->cellPerm[i]] = i;
    if (c->renumber) {
        std::vector<int> vkey(nV, nC);
        for (int v = 0; v < nV; v++)
            for (int k = 0; k < D; k++) {
                int cc = m->cellsOnVertex[(size_t)v * D + k];
                if (cc >= 1 && cc <= nC) vkey[v] = std::min(vkey[v], c->cellInv[cc - 1]);
            }
        auto cmp = [&](int a, int b) { return vkey[a] != vkey[b] ? vkey[a] < vkey[b] : a < b; };
        std::sort(c->vertPerm.begin(), c->vertPerm.begin() + c->nVerticesSolve, cmp);
        std::sort(c->vertPerm.begin() + c->nVerticesSolve, c->vertPerm.end(), cmp);
    }
    c->vertInv.assign(nV + 1, nV);
    for (int i = 0; i < nV; i++) c->vertInv[c->vertPerm[i]] = i;

    int rc;
    if ((rc = devAlloc(c, (void **)&c->d_cellPerm, sizeof(int) * c->padC))) return rc;
    if ((rc = devAlloc(c, (void **)&c->d_vertPerm, sizeof(int) * c->padV))) return rc;
    HIPCHK(hipMemcpyAsync(c->d_cellPerm, c->cellPerm.data(), sizeof(int) * nC, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_vertPerm, c->vertPerm.data(), sizeof(int) * nV, hipMemcpyHostToDevice, c->stream));
    if ((rc = devAlloc(c, (void **)&c->d_flag, 64))) return rc;

    // ---- field registry
    struct Spec {
        const char *name;
        SihLoc loc;
        int lead;
        bool isInt;
    };
    const int MM = M * M;
    const Spec specs[] = {
        // connectivity in internal numbering
        {"_nEdgesOnCell", LOC_CELL, 1, true}, {"_voc", LOC_CELL, M, true}, {"_coc", LOC_CELL, M, true},
        {"_cov", LOC_VERTEX, D, true}, {"_gidx", LOC_VERTEX, D, true},
        {"_vertexActive", LOC_VERTEX, 1, true},
        // geometry
        {"xCell", LOC_CELL, 1, false}, {"yCell", LOC_CELL, 1, false}, {"zCell", LOC_CELL, 1, false},
        {"xVertex", LOC_VERTEX, 1, false}, {"yVertex", LOC_VERTEX, 1, false}, {"zVertex", LOC_VERTEX, 1, false},
        {"areaCell", LOC_CELL, 1, false}, {"areaTriangle", LOC_VERTEX, 1, false}, {"fVertex", LOC_VERTEX, 1, false},
        // boundary
        {"interiorVertex", LOC_VERTEX, 1, true},
        // velocity_solver pool
        {"uVelocity", LOC_VERTEX, 1, false}, {"vVelocity", LOC_VERTEX, 1, false},
        {"uVelocityInitial", LOC_VERTEX, 1, false}, {"vVelocityInitial", LOC_VERTEX, 1, false},
        {"solveStress", LOC_CELL, 1, true}, {"solveVelocity", LOC_VERTEX, 1, true},
        {"solveVelocityPrevious", LOC_VERTEX, 1, true}, {"icePressure", LOC_CELL, 1, false},
        {"stressDivergenceU", LOC_VERTEX, 1, false}, {"stressDivergenceV", LOC_VERTEX, 1, false},
        {"airStressCellU", LOC_CELL, 1, false}, {"airStressCellV", LOC_CELL, 1, false},
        {"airStressVertexU", LOC_VERTEX, 1, false}, {"airStressVertexV", LOC_VERTEX, 1, false},
        {"uOceanVelocityVertex", LOC_VERTEX, 1, false}, {"vOceanVelocityVertex", LOC_VERTEX, 1, false},
        {"oceanStressU", LOC_VERTEX, 1, false}, {"oceanStressV", LOC_VERTEX, 1, false},
        {"oceanStressCoeff", LOC_VERTEX, 1, false}, {"oceanStressCellU", LOC_CELL, 1, false},
        {"oceanStressCellV", LOC_CELL, 1, false}, {"seaSurfaceTiltVertexU", LOC_VERTEX, 1, false},
        {"seaSurfaceTiltVertexV", LOC_VERTEX, 1, false}, {"surfaceTiltForceU", LOC_VERTEX, 1, false},
        {"surfaceTiltForceV", LOC_VERTEX, 1, false}, {"totalMassVertexfVertex", LOC_VERTEX, 1, false},
        {"divergence", LOC_CELL, 1, false}, {"shear", LOC_CELL, 1, false},
        // velocity_variational pool
        {"strain11", LOC_CELL, M, false}, {"strain22", LOC_CELL, M, false}, {"strain12", LOC_CELL, M, false},
        {"stress11", LOC_CELL, M, false}, {"stress22", LOC_CELL, M, false}, {"stress12", LOC_CELL, M, false},
        {"replacementPressure", LOC_CELL, M, false}, {"principalStress1", LOC_CELL, M, false},
        {"principalStress2", LOC_CELL, M, false}, {"basisGradientU", LOC_CELL, MM, false},
        {"basisGradientV", LOC_CELL, MM, false}, {"basisIntegralsU", LOC_CELL, MM, false},
        {"basisIntegralsV", LOC_CELL, MM, false}, {"basisIntegralsMetric", LOC_CELL, MM, false},
        {"tanLatVertexRotatedOverRadius", LOC_VERTEX, 1, false}, {"variationalDenominator", LOC_VERTEX, 1, false},
        {"cellVerticesAtVertex", LOC_VERTEX, D, true},
        // compact gradients and per-corner divergence partial sums (library-internal)
        {"_gradU3", LOC_CELL, 3 * M, false}, {"_gradV3", LOC_CELL, 3 * M, false},
        {"_partU", LOC_CELL, M, false}, {"_partV", LOC_CELL, M, false}, {"_nEdgesDerive", LOC_CELL, 1, true},
        {"_airTiltU", LOC_VERTEX, 1, false}, {"_airTiltV", LOC_VERTEX, 1, false},
        // icestate, aggregates, coupling
        {"iceAreaVertex", LOC_VERTEX, 1, false}, {"totalMassVertex", LOC_VERTEX, 1, false},
        {"iceAreaCellInitial", LOC_CELL, 1, false}, {"totalMassCell", LOC_CELL, 1, false},
        {"iceAreaCell", LOC_CELL, 1, false}, {"iceVolumeCell", LOC_CELL, 1, false},
        {"snowVolumeCell", LOC_CELL, 1, false}, {"uOceanVelocity", LOC_CELL, 1, false},
        {"vOceanVelocity", LOC_CELL, 1, false}, {"seaSurfaceTiltU", LOC_CELL, 1, false},
        {"seaSurfaceTiltV", LOC_CELL, 1, false}, {"landIceMask", LOC_CELL, 1, true},
        {"landIceMaskVertex", LOC_VERTEX, 1, true}, {"uAirVelocity", LOC_CELL, 1, false},
        {"vAirVelocity", LOC_CELL, 1, false}, {"airDensity", LOC_CELL, 1, false},
        {"ridgeConvergence", LOC_CELL, 1, false}, {"ridgeShear", LOC_CELL, 1, false},
    };
    for (const Spec &s : specs)
        if ((rc = addField(c, s.name, s.loc, s.lead, s.isInt))) return rc;

    // ---- connectivity in internal numbering (built on the host, uploaded as SoA)
    {
        std::vector<int> ne(c->padC, 0), voc((size_t)M * c->padC, nV), coc((size_t)M * c->padC, nC);
        for (int i = 0; i < nC; i++) {
            int h = c->cellPerm[i];
            int E = m->nEdgesOnCell[h];
            ne[i] = E;
            for (int s = 0; s < E; s++) {
                int v = m->verticesOnCell[(size_t)h * M + s];
                int nb = m->cellsOnCell[(size_t)h * M + s];
                voc[(size_t)s * c->padC + i] = (v >= 1 && v <= nV) ? c->vertInv[v - 1] : nV;
                coc[(size_t)s * c->padC + i] = (nb >= 1 && nb <= nC) ? c->cellInv[nb - 1] : nC;
            }
        }
        std::vector<int> cov((size_t)D * c->padV, nC);
        for (int i = 0; i < nV; i++) {
            int h = c->vertPerm[i];
            for (int k = 0; k < D; k++) {
                int cc = m->cellsOnVertex[(size_t)h * D + k];
                cov[(size_t)k * c->padV + i] = (cc >= 1 && cc <= nC) ? c->cellInv[cc - 1] : nC;
            }
        }
        c->hVoc = voc;
        c->hNe = ne;
        // dvEdge of edgesOnCell(s, cell) per cell and slot s; the
        // piecewise-linear basis needs it (velocity_solver_pwl.F:163-167)
        if (m->edgesOnCell && m->dvEdge && m->nEdges > 0) {
            int rc2 = addField(c, "_dvEdgeOnCell", LOC_CELL, M, false);
            if (rc2) return rc2;
            std::vector<double> dvoc((size_t)M * c->padC, 0.0);
            for (int i = 0; i < nC; i++) {
                int h = c->cellPerm[i];
                for (int s = 0; s < m->nEdgesOnCell[h]; s++) {
                    int e = m->edgesOnCell[(size_t)h * M + s];
                    if (e >= 1 && e <= m->nEdges) dvoc[(size_t)s * c->padC + i] = m->dvEdge[e - 1];
                }
            }
            HIPCHK(hipMemcpyAsync(c->fields["_dvEdgeOnCell"].d, dvoc.data(), dvoc.size() * 8, hipMemcpyHostToDevice,
                                  c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        HIPCHK(hipMemcpyAsync(c->fields["_nEdgesOnCell"].d, ne.data(), ne.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->fields["_voc"].d, voc.data(), voc.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->fields["_coc"].d, coc.data(), coc.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->fields["_cov"].d, cov.data(), cov.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    struct G {
        const char *n;
        const double *p;
        size_t cnt;
    };
    const G geo[] = {{"xCell", m->xCell, (size_t)nC},       {"yCell", m->yCell, (size_t)nC},
                     {"zCell", m->zCell, (size_t)nC},       {"xVertex", m->xVertex, (size_t)nV},
                     {"yVertex", m->yVertex, (size_t)nV},   {"zVertex", m->zVertex, (size_t)nV},
                     {"areaCell", m->areaCell, (size_t)nC}, {"areaTriangle", m->areaTriangle, (size_t)nV},
                     {"fVertex", m->fVertex, (size_t)nV}};
    for (const G &g : geo)
        if ((rc = sih_field_put(c, g.n, g.p, g.cnt))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    guard.c = nullptr;
    *out = c;
    return SIH_OK;
}

extern "C" int sih_block_destroy(sih_ctx *c)
{
    if (!c) return SIH_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->graphExec) (void)hipGraphExecDestroy(c->graphExec);
    sih_comm_free(c);
    sih_persist_destroy(c);
    sih_ridge_destroy(c);
    sih_ir_destroy(c);
    sih_weak_destroy(c);
    sih_upwind_destroy(c);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    for (void *p : c->allocs) (void)hipFree(p);
    for (int k = 0; k < 2; k++) {
        if (c->d_send[k]) (void)hipFree(c->d_send[k]);
        if (c->d_recv[k]) (void)hipFree(c->d_recv[k]);
        for (int j = 0; j < 2; j++) {
            if (c->d_segStart[k][j]) (void)hipFree(c->d_segStart[k][j]);
            if (c->d_segCnt[k][j]) (void)hipFree(c->d_segCnt[k][j]);
        }
    }
    if (c->d_stage) (void)hipFree(c->d_stage);
    for (int k = 0; k < 4; k++)
        if (c->d_part[k]) (void)hipFree(c->d_part[k]);
    for (void *p : {(void *)c->d_sbIdx, (void *)c->d_sbOrigin, (void *)c->d_sbSign, (void *)c->d_sbMaskV,
                    (void *)c->d_sbMaskC, (void *)c->d_sbTmp})
        if (p) (void)hipFree(p);
    for (auto &kv : c->partGraphs) (void)hipGraphExecDestroy(kv.second);
    for (auto &kv : c->haloTables) {
        if (kv.second.d_planes) (void)hipFree(kv.second.d_planes);
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++)
                if (kv.second.d_base[a][b]) (void)hipFree(kv.second.d_base[a][b]);
    }
    if (c->copyStream) {
        (void)hipStreamSynchronize(c->copyStream);
        (void)hipStreamDestroy(c->copyStream);
    }
    if (c->evCopyFrom) (void)hipEventDestroy(c->evCopyFrom);
    if (c->evCopyGot) (void)hipEventDestroy(c->evCopyGot);
    if (c->d_stageCopy) (void)hipFree(c->d_stageCopy);
    if (c->h_stageCopy) (void)hipHostFree(c->h_stageCopy);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (hipEvent_t e : c->stageEvents) (void)hipEventDestroy(e);
    if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
    if (c->counted) g_liveCtx[c->device]--;
    delete c;
    return SIH_OK;
}

extern "C" int sih_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int sih_set_stream(sih_ctx *c, void *s)
{
    if (!c) return SIH_ERR_ARG;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->ownStream;
    if (c->graphExec) {  // a captured graph is tied to nothing stream-specific, but rebuild to be safe
        (void)hipGraphExecDestroy(c->graphExec);
        c->graphExec = nullptr;
        c->graphN = 0;
    }
    for (auto &kv : c->partGraphs) (void)hipGraphExecDestroy(kv.second);
    c->partGraphs.clear();
    return SIH_OK;
}

extern "C" int sih_sync(sih_ctx *c)
{
    if (!c) return SIH_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return sih_persist_status(c);  // a persistent launch that gave up is reported here
}

// Page-locked host memory for the arrays a host exchanges with the library every step.  Transfers from and
// to such memory are plain DMA (measured 57 GB/s either way on an MI355X box, against 30 GB/s up and 18 GB/s
// down for pageable memory, which the runtime has to stage).
extern "C" int sih_host_alloc(void **p, size_t bytes)
{
    if (!p) {
        sih_set_error("sih_host_alloc: null argument");
        return SIH_ERR_ARG;
    }
    *p = nullptr;
    HIPCHK(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocPortable));
    return SIH_OK;
}

extern "C" int sih_host_free(void *p)
{
    if (p) HIPCHK(hipHostFree(p));
    return SIH_OK;
}

extern "C" size_t sih_field_count(sih_ctx *c, const char *name, int *elemSize)
{
    SihField *f = c ? sih_find_field(c, name) : nullptr;
    if (!f) return 0;
    if (elemSize) *elemSize = f->isInt ? 4 : 8;
    return (size_t)f->lead * locCount(c, f->loc);
}

// whether the runtime knows the host address as page-locked memory (hipHostMalloc, hipHostRegister)
static bool isPageLocked(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // an address the runtime has never seen is no error here
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// dst[k] <- base + off[k] (toUser) or the other way, len[k] bytes each, in pieces of 2 MiB dealt to up to eight
// threads (one core moves about 10 GB/s, the DMA beside it 50)
static void hostCopyParallel(const std::vector<void *> &dst, const char *base, const std::vector<size_t> &off,
                             const std::vector<size_t> &len, bool toUser = true)
{
    struct Piece { char *d; const char *s; size_t n; };
    std::vector<Piece> pieces;
    const size_t chunk = (size_t)2 << 20;
    for (size_t k = 0; k < dst.size(); k++)
        for (size_t o = 0; o < len[k]; o += chunk) {
            char *u = (char *)dst[k] + o, *l = const_cast<char *>(base) + off[k] + o;
            pieces.push_back({toUser ? u : l, toUser ? l : u, std::min(chunk, len[k] - o)});
        }
    unsigned nThreads = std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("SIH_COPY_THREADS")) nThreads = (unsigned)std::max(1, atoi(e));
    nThreads = (unsigned)std::min<size_t>(nThreads, std::max<size_t>(pieces.size(), 1));
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (size_t i = next.fetch_add(1); i < pieces.size(); i = next.fetch_add(1)) memcpy(pieces[i].d, pieces[i].s, pieces[i].n);
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nThreads; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
}

// true if host array k begins off[k] bytes after host array 0 for every k: the batch is one piece of host memory laid
// out like the staging buffer (the padding between two fields then belongs to the host's arena as well)
static bool hostsFollowStageLayout(int nf, void *const *hosts, const std::vector<size_t> &off)
{
    if (nf < 2) return false;
    for (int k = 1; k < nf; k++)
        if ((const char *)hosts[k] != (const char *)hosts[0] + off[k]) return false;
    return true;
}

static void launchLayout(sih_ctx *c, SihField *f, char *stage, bool put, hipStream_t st)
{
    const int n = locCount(c, f->loc), pad = locPad(c, f->loc), L = f->lead;
    const int *perm = c->renumber ? locPerm(c, f->loc) : nullptr;
    const long total = (long)n * L;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (put) {
        if (f->isInt)
            hipLaunchKernelGGL(k_put<int>, grid, block, 0, st, (int *)f->d, (const int *)stage, perm, n, L, pad);
        else
            hipLaunchKernelGGL(k_put<double>, grid, block, 0, st, (double *)f->d, (const double *)stage, perm, n, L, pad);
    } else {
        if (f->isInt)
            hipLaunchKernelGGL(k_get<int>, grid, block, 0, st, (int *)stage, (const int *)f->d, perm, n, L, pad);
        else
            hipLaunchKernelGGL(k_get<double>, grid, block, 0, st, (double *)stage, (const double *)f->d, perm, n, L, pad);
    }
}

// putgetMany for host arrays that are not page-locked (see there)
static int putgetStaged(sih_ctx *c, int nf, const std::vector<SihField *> &fs, void *const *hosts, const size_t *counts,
                        const std::vector<size_t> &off, bool put)
{
    if (c->hStageBytes < off[nf]) {
        if (c->h_stage) HIPCHK(hipHostFree(c->h_stage));
        c->h_stage = nullptr;
        c->hStageBytes = 0;
        HIPCHK(hipHostMalloc(&c->h_stage, off[nf], hipHostMallocPortable));
        c->hStageBytes = off[nf];
    }
    // groups of consecutive fields of about 16 MiB
    std::vector<int> first{0};
    for (int k = 0; k < nf; k++)
        if (off[k + 1] - off[first.back()] >= ((size_t)16 << 20) && k + 1 < nf) first.push_back(k + 1);
    first.push_back(nf);
    const int nG = (int)first.size() - 1;
    while ((int)c->stageEvents.size() < nG) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->stageEvents.push_back(e);
    }
    auto group = [&](int g, std::vector<void *> &dst, std::vector<size_t> &o, std::vector<size_t> &len) {
        dst.clear();
        o.clear();
        len.clear();
        for (int k = first[g]; k < first[g + 1]; k++) {
            dst.push_back(hosts[k]);
            o.push_back(off[k]);
            len.push_back(counts[k] * (fs[k]->isInt ? 4 : 8));
        }
    };
    std::vector<void *> dst;
    std::vector<size_t> o, len;
    if (put) {
        for (int g = 0; g < nG; g++) {
            group(g, dst, o, len);
            hostCopyParallel(dst, (const char *)c->h_stage, o, len, false);  // host arrays -> page-locked buffer
            const size_t b0 = off[first[g]], b1 = off[first[g + 1]];
            HIPCHK(hipMemcpyAsync((char *)c->d_stage + b0, (char *)c->h_stage + b0, b1 - b0, hipMemcpyHostToDevice, c->stream));
            for (int k = first[g]; k < first[g + 1]; k++) launchLayout(c, fs[k], (char *)c->d_stage + off[k], true, c->stream);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        return SIH_OK;
    }
    for (int g = 0; g < nG; g++) {
        for (int k = first[g]; k < first[g + 1]; k++) launchLayout(c, fs[k], (char *)c->d_stage + off[k], false, c->stream);
        HIPCHK(hipGetLastError());
        const size_t b0 = off[first[g]], b1 = off[first[g + 1]];
        HIPCHK(hipMemcpyAsync((char *)c->h_stage + b0, (char *)c->d_stage + b0, b1 - b0, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipEventRecord(c->stageEvents[g], c->stream));
    }
    for (int g = 0; g < nG; g++) {
        HIPCHK(hipEventSynchronize(c->stageEvents[g]));
        group(g, dst, o, len);
        hostCopyParallel(dst, (const char *)c->h_stage, o, len, true);  // page-locked buffer -> host arrays
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return SIH_OK;
}

// One or many fields between host arrays and the device planes.  Every field gets its own 256-byte
// aligned region of the staging buffer, so all copies and all layout kernels of a batch are queued
// back to back and the stream is waited for once at the end.
static int putgetMany(sih_ctx *c, int nf, const char *const *names, void *const *hosts, const size_t *counts, bool put)
{
    if (!c || nf < 0 || (nf > 0 && (!names || !hosts || !counts))) {
        sih_set_error("sih_field_put/get: null argument");
        return SIH_ERR_ARG;
    }
    std::vector<SihField *> fs(nf);
    std::vector<size_t> off(nf + 1, 0);
    for (int k = 0; k < nf; k++) {
        if (!names[k] || !hosts[k]) {
            sih_set_error("sih_field_put/get: null argument");
            return SIH_ERR_ARG;
        }
        SihField *f = sih_find_field(c, names[k]);
        if (!f) {
            sih_set_error(std::string("unknown field: ") + names[k]);
            return SIH_ERR_ARG;
        }
        if (counts[k] != (size_t)locCount(c, f->loc) * f->lead) {
            sih_set_error(std::string("field ") + names[k] + ": element count mismatch");
            return SIH_ERR_ARG;
        }
        fs[k] = f;
        size_t bytes = counts[k] * (f->isInt ? 4 : 8);
        off[k + 1] = off[k] + ((bytes + 255) & ~(size_t)255);
    }
    if (nf == 0) return SIH_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc = ensureStage(c, off[nf]);
    if (rc) return rc;
    {
        // Ordinary (pageable) host arrays, as a host model's framework allocates them: the runtime would stage every copy
        // itself, at a third of the link's rate on the way down.  A batch of some size goes through the library's own
        // page-locked buffer instead, in groups of fields, with the DMA of one group beside the threads that move the
        // group before (down) or after (up) it between that buffer and the host arrays.
        static const bool staged = [] {
            const char *e = getenv("SIH_STAGE_PAGEABLE");
            return !(e && atoi(e) == 0);
        }();
        bool pageable = false;
        for (int k = 0; k < nf && !pageable; k++) pageable = !isPageLocked(hosts[k]);
        if (staged && pageable && off[nf] >= ((size_t)4 << 20)) return putgetStaged(c, nf, fs, hosts, counts, off, put);
    }
    // A host that keeps the fields of a batch in one arena, each at the next multiple of 256 bytes -- the layout of
    // the staging buffer -- gets one copy for the whole batch instead of one per field (a copy costs 10-20 us before
    // its first byte moves).
    const bool oneCopy = hostsFollowStageLayout(nf, hosts, off);
    if (put && oneCopy) HIPCHK(hipMemcpyAsync(c->d_stage, hosts[0], off[nf - 1] + counts[nf - 1] * (fs[nf - 1]->isInt ? 4 : 8),
                                              hipMemcpyHostToDevice, c->stream));
    for (int k = 0; k < nf; k++) {
        SihField *f = fs[k];
        int n = locCount(c, f->loc), pad = locPad(c, f->loc), L = f->lead;
        size_t bytes = counts[k] * (f->isInt ? 4 : 8);
        char *stage = (char *)c->d_stage + off[k];
        const int *perm = c->renumber ? locPerm(c, f->loc) : nullptr;
        long total = (long)n * L;
        dim3 grid((unsigned)((total + 255) / 256)), block(256);
        if (put) {
            if (!oneCopy) HIPCHK(hipMemcpyAsync(stage, hosts[k], bytes, hipMemcpyHostToDevice, c->stream));
            if (f->isInt)
                hipLaunchKernelGGL(k_put<int>, grid, block, 0, c->stream, (int *)f->d, (const int *)stage, perm, n, L, pad);
            else
                hipLaunchKernelGGL(k_put<double>, grid, block, 0, c->stream, (double *)f->d, (const double *)stage, perm, n, L,
                                   pad);
        } else {
            if (f->isInt)
                hipLaunchKernelGGL(k_get<int>, grid, block, 0, c->stream, (int *)stage, (const int *)f->d, perm, n, L, pad);
            else
                hipLaunchKernelGGL(k_get<double>, grid, block, 0, c->stream, (double *)stage, (const double *)f->d, perm, n, L,
                                   pad);
        }
        HIPCHK(hipGetLastError());
    }
    if (!put) {
        if (oneCopy)
            HIPCHK(hipMemcpyAsync(hosts[0], c->d_stage, off[nf - 1] + counts[nf - 1] * (fs[nf - 1]->isInt ? 4 : 8),
                                  hipMemcpyDeviceToHost, c->stream));
        else
            for (int k = 0; k < nf; k++)
                HIPCHK(hipMemcpyAsync(hosts[k], (char *)c->d_stage + off[k], counts[k] * (fs[k]->isInt ? 4 : 8),
                                      hipMemcpyDeviceToHost, c->stream));
    }
    // the staging buffer is reused by the next call, and the caller owns the host arrays again
    HIPCHK(hipStreamSynchronize(c->stream));
    return SIH_OK;
}

// A download that drains behind the work the host queues next (the advection after the velocity solver, the next
// step's uploads).  The fields are read by layout kernels on a second stream once everything queued on the context's
// stream so far is done; the context's stream then waits only for those short kernels, not for the copies to the
// host, which run by DMA beside whatever is launched next.  The host arrays belong to the library until
// sih_fields_get_finish returns.  Page-locked host arrays (sih_host_alloc) make the copies truly asynchronous; with
// ordinary memory the runtime stages them and this call may return only when they are done (still correct).
extern "C" int sih_fields_get_start(sih_ctx *c, int nf, const char *const *names, void *const *hosts, const size_t *counts)
{
    if (!c || nf < 0 || (nf > 0 && (!names || !hosts || !counts))) {
        sih_set_error("sih_fields_get_start: null argument");
        return SIH_ERR_ARG;
    }
    std::vector<SihField *> fs(nf);
    std::vector<size_t> off(nf + 1, 0);
    for (int k = 0; k < nf; k++) {
        SihField *f = (names[k] && hosts[k]) ? sih_find_field(c, names[k]) : nullptr;
        if (!f) {
            sih_set_error(std::string("sih_fields_get_start: unknown field or null array: ") + (names[k] ? names[k] : "(null)"));
            return SIH_ERR_ARG;
        }
        if (counts[k] != (size_t)locCount(c, f->loc) * f->lead) {
            sih_set_error(std::string("field ") + names[k] + ": element count mismatch");
            return SIH_ERR_ARG;
        }
        fs[k] = f;
        off[k + 1] = off[k] + ((counts[k] * (f->isInt ? 4 : 8) + 255) & ~(size_t)255);
    }
    if (nf == 0) return SIH_OK;
    HIPCHK(hipSetDevice(c->device));
    if (!c->copyStream) {
        HIPCHK(hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->evCopyFrom, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->evCopyGot, hipEventDisableTiming));
    }
    // one download at a time: the staging buffer and the host arrays of the previous one must be free
    HIPCHK(hipStreamSynchronize(c->copyStream));
    c->copyPending = false;
    if (c->stageCopyBytes < off[nf]) {
        if (c->d_stageCopy) HIPCHK(hipFree(c->d_stageCopy));
        c->d_stageCopy = nullptr;
        c->stageCopyBytes = 0;
        HIPCHK(hipMalloc(&c->d_stageCopy, off[nf]));
        c->stageCopyBytes = off[nf];
    }
    HIPCHK(hipEventRecord(c->evCopyFrom, c->stream));
    HIPCHK(hipStreamWaitEvent(c->copyStream, c->evCopyFrom, 0));
    for (int k = 0; k < nf; k++) {
        SihField *f = fs[k];
        const int n = locCount(c, f->loc), pad = locPad(c, f->loc), L = f->lead;
        char *stage = (char *)c->d_stageCopy + off[k];
        const int *perm = c->renumber ? locPerm(c, f->loc) : nullptr;
        const long total = (long)n * L;
        dim3 grid((unsigned)((total + 255) / 256)), block(256);
        if (f->isInt)
            hipLaunchKernelGGL(k_get<int>, grid, block, 0, c->copyStream, (int *)stage, (const int *)f->d, perm, n, L, pad);
        else
            hipLaunchKernelGGL(k_get<double>, grid, block, 0, c->copyStream, (double *)stage, (const double *)f->d, perm, n, L,
                               pad);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(c->evCopyGot, c->copyStream));
    HIPCHK(hipStreamWaitEvent(c->stream, c->evCopyGot, 0));  // later work may overwrite the fields: not before they are read
    c->copyPending = true;
    c->copyDst.clear();
    c->copyOff.clear();
    c->copyLen.clear();
    bool pageable = false;
    for (int k = 0; k < nf && !pageable; k++) pageable = !isPageLocked(hosts[k]);
    if (pageable) {
        // ordinary host memory: a copy into it is staged by the runtime and holds up the caller.  One DMA into the
        // library's own page-locked buffer instead; the last hop is made by sih_fields_get_finish.
        if (c->hStageCopyBytes < off[nf]) {
            if (c->h_stageCopy) HIPCHK(hipHostFree(c->h_stageCopy));
            c->h_stageCopy = nullptr;
            c->hStageCopyBytes = 0;
            HIPCHK(hipHostMalloc(&c->h_stageCopy, off[nf], hipHostMallocPortable));
            c->hStageCopyBytes = off[nf];
        }
        HIPCHK(hipMemcpyAsync(c->h_stageCopy, c->d_stageCopy, off[nf - 1] + counts[nf - 1] * (fs[nf - 1]->isInt ? 4 : 8),
                              hipMemcpyDeviceToHost, c->copyStream));
        for (int k = 0; k < nf; k++) {
            c->copyDst.push_back(hosts[k]);
            c->copyOff.push_back(off[k]);
            c->copyLen.push_back(counts[k] * (fs[k]->isInt ? 4 : 8));
        }
        return SIH_OK;
    }
    if (hostsFollowStageLayout(nf, hosts, off))
        HIPCHK(hipMemcpyAsync(hosts[0], c->d_stageCopy, off[nf - 1] + counts[nf - 1] * (fs[nf - 1]->isInt ? 4 : 8),
                              hipMemcpyDeviceToHost, c->copyStream));
    else
        for (int k = 0; k < nf; k++)
            HIPCHK(hipMemcpyAsync(hosts[k], (char *)c->d_stageCopy + off[k], counts[k] * (fs[k]->isInt ? 4 : 8),
                                  hipMemcpyDeviceToHost, c->copyStream));
    return SIH_OK;
}

// waits for the download sih_fields_get_start began; the host arrays are the caller's again.  No-op without one.
extern "C" int sih_fields_get_finish(sih_ctx *c)
{
    if (!c) return SIH_ERR_ARG;
    if (!c->copyStream || !c->copyPending) return SIH_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->copyStream));
    c->copyPending = false;
    if (!c->copyDst.empty()) {
        hostCopyParallel(c->copyDst, (const char *)c->h_stageCopy, c->copyOff, c->copyLen);
        c->copyDst.clear();
        c->copyOff.clear();
        c->copyLen.clear();
    }
    return SIH_OK;
}

// ---- hand-off to the column package (ridging) ----------------------------------------------------------------
// One record per owned cell in the layout colpkg_step_ridge takes its arguments in (column.F:3465-3530): for cell
// h, category cat, the listed fields' layers one after the other -- trcrn(nt, ncat) of that cell is then one
// contiguous piece of the host buffer.  Only cells [firstCell, firstCell + nCells) in host numbering travel.
__global__ void k_column_get(double *__restrict__ dst, const double *__restrict__ src, const int *__restrict__ perm,
                             int nAll, int L, int nCat, int pad, int S, int slot0, int cell0, int n)
{
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)nAll * L * nCat) return;
    const int l = (int)(t / nAll), i = (int)(t % nAll);
    const int h = (perm ? perm[i] : i) - cell0;
    if (h < 0 || h >= n) return;
    const int cat = l / L, layer = l - cat * L;
    dst[((size_t)h * nCat + cat) * S + slot0 + layer] = src[(size_t)l * pad + i];
}

__global__ void k_column_put(double *__restrict__ dst, const double *__restrict__ src, const int *__restrict__ perm,
                             int nAll, int L, int nCat, int pad, int S, int slot0, int cell0, int n)
{
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)nAll * L * nCat) return;
    const int l = (int)(t / nAll), i = (int)(t % nAll);
    const int h = (perm ? perm[i] : i) - cell0;
    if (h < 0 || h >= n) return;
    const int cat = l / L, layer = l - cat * L;
    dst[(size_t)l * pad + i] = src[((size_t)h * nCat + cat) * S + slot0 + layer];
}

static int columnMove(sih_ctx *c, int nf, const char *const *names, int nCat, int firstCell, int nCells, double *host,
                      size_t count, bool put)
{
    const char *who = put ? "sih_column_put" : "sih_column_get";
    if (!c || nf < 1 || !names || !host || nCat < 1 || firstCell < 0 || nCells < 0 || firstCell + nCells > c->nCells) {
        sih_set_error(std::string(who) + ": bad argument");
        return SIH_ERR_ARG;
    }
    std::vector<SihField *> fs(nf);
    std::vector<int> slot0(nf, 0);
    int S = 0;
    for (int k = 0; k < nf; k++) {
        SihField *f = names[k] ? sih_find_field(c, names[k]) : nullptr;
        if (!f) {
            sih_set_error(std::string(who) + ": unknown field: " + (names[k] ? names[k] : "(null)"));
            return SIH_ERR_ARG;
        }
        if (f->loc != LOC_CELL || f->isInt || f->lead % nCat != 0) {
            sih_set_error(std::string(who) + ": field " + names[k] +
                          " must be a real cell field whose leading size is a multiple of nCategories");
            return SIH_ERR_ARG;
        }
        fs[k] = f;
        slot0[k] = S;
        S += f->lead / nCat;
    }
    if (count != (size_t)nCells * nCat * S) {
        sih_set_error(std::string(who) + ": element count mismatch");
        return SIH_ERR_ARG;
    }
    if (nCells == 0) return SIH_OK;
    HIPCHK(hipSetDevice(c->device));
    const size_t bytes = count * sizeof(double);
    int rc = ensureStage(c, bytes);
    if (rc) return rc;
    double *stage = (double *)c->d_stage;
    const int *perm = c->renumber ? c->d_cellPerm : nullptr;
    if (put) HIPCHK(hipMemcpyAsync(stage, host, bytes, hipMemcpyHostToDevice, c->stream));
    for (int k = 0; k < nf; k++) {
        SihField *f = fs[k];
        const int L = f->lead / nCat;
        const long total = (long)c->nCells * f->lead;
        dim3 grid((unsigned)((total + 255) / 256)), block(256);
        if (put)
            hipLaunchKernelGGL(k_column_put, grid, block, 0, c->stream, (double *)f->d, (const double *)stage, perm, c->nCells,
                               L, nCat, c->padC, S, slot0[k], firstCell, nCells);
        else
            hipLaunchKernelGGL(k_column_get, grid, block, 0, c->stream, stage, (const double *)f->d, perm, c->nCells, L, nCat,
                               c->padC, S, slot0[k], firstCell, nCells);
        HIPCHK(hipGetLastError());
    }
    if (!put) HIPCHK(hipMemcpyAsync(host, stage, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SIH_OK;
}

extern "C" int sih_column_get(sih_ctx *c, int nFields, const char *const *names, int nCategories, int firstCell,
                              int nCells, double *host, size_t count)
{
    return columnMove(c, nFields, names, nCategories, firstCell, nCells, host, count, false);
}

extern "C" int sih_column_put(sih_ctx *c, int nFields, const char *const *names, int nCategories, int firstCell,
                              int nCells, const double *host, size_t count)
{
    return columnMove(c, nFields, names, nCategories, firstCell, nCells, (double *)host, count, true);
}

static int putget(sih_ctx *c, const char *name, void *host, size_t count, bool put)
{
    return putgetMany(c, 1, &name, &host, &count, put);
}

void sih_drop_graphs(sih_ctx *c)
{
    if (c->graphExec) {
        (void)hipGraphExecDestroy(c->graphExec);
        c->graphExec = nullptr;
        c->graphN = 0;
    }
    for (auto &kv : c->partGraphs) (void)hipGraphExecDestroy(kv.second);
    c->partGraphs.clear();
}

// ------------------------------------------------------------------------------------------
// Staging memory that exists without any allocation call: one array in the code object, 16 MiB on every GPU the
// library runs on, for the entry point that must not allocate (sih_field_get_noalloc).  Host code never uses
// the array's name as an address: the per-device address comes from hipGetSymbolAddress, once per device.
#define SIH_STATIC_STAGE_BYTES (16u << 20)
#define SIH_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[SIH_STATIC_STAGE_BYTES];
static void *g_scratchPtr[SIH_MAX_DEVICES];

// with `device` current; called from sih_block_create, so every context's device has its address before the
// first sih_field_get_noalloc
static int resolveScratch(int device)
{
    if (device < 0 || device >= SIH_MAX_DEVICES) {
        sih_set_error("device ordinal out of range");
        return SIH_ERR_ARG;
    }
    if (g_scratchPtr[device]) return SIH_OK;
    void *p = nullptr;
    HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    g_scratchPtr[device] = p;
    return SIH_OK;
}

extern "C" int sih_field_get_noalloc(sih_ctx *c, const char *name, void *host, size_t count)
{
    if (!c || !name || !host) {
        sih_set_error("sih_field_get_noalloc: null argument");
        return SIH_ERR_ARG;
    }
    SihField *f = nullptr;
    for (auto &kv : c->fields)  // no temporary string: this path allocates nothing, on the host either
        if (!strcmp(kv.first.c_str(), name)) {
            f = &kv.second;
            break;
        }
    if (!f) {
        sih_set_error("sih_field_get_noalloc: unknown field");
        return SIH_ERR_ARG;
    }
    const int n = locCount(c, f->loc), pad = locPad(c, f->loc), L = f->lead;
    if (count != (size_t)n * L) {
        sih_set_error("sih_field_get_noalloc: element count mismatch");
        return SIH_ERR_ARG;
    }
    const size_t bytes = count * (f->isInt ? 4 : 8);
    if (bytes > SIH_STATIC_STAGE_BYTES) {
        sih_set_error("sih_field_get_noalloc: the field is larger than the static staging buffer (16 MiB); use sih_field_get");
        return SIH_ERR_ARG;
    }
    void *stage = g_scratchPtr[c->device];
    if (!stage) {
        sih_set_error("sih_field_get_noalloc: the staging buffer of this device was not resolved");
        return SIH_ERR_STATE;
    }
    HIPCHK(hipSetDevice(c->device));
    if (count == 0) return SIH_OK;
    const int *perm = c->renumber ? locPerm(c, f->loc) : nullptr;
    dim3 grid((unsigned)((count + 255) / 256)), block(256);
    if (f->isInt)
        hipLaunchKernelGGL(k_get<int>, grid, block, 0, c->stream, (int *)stage, (const int *)f->d, perm, n, L, pad);
    else
        hipLaunchKernelGGL(k_get<double>, grid, block, 0, c->stream, (double *)stage, (const double *)f->d, perm, n, L, pad);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host, stage, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));  // the staging buffer is free for the next call, the host array is the caller's
    return SIH_OK;
}

static void notePut(sih_ctx *c, const char *name)
{
    if (c->inited && (!strncmp(name, "basisGradient", 13) || !strncmp(name, "basisIntegrals", 14)) &&
        (c->sparseGrad || c->symMetric || c->deriveGrad || c->deriveRow)) {
        // the compact gradients are stale until sih_evp_static_changed(), and the metric integrals may no longer be
        // symmetric; the dense kernel variant reads what the host has just put.  Captured graphs hold launches of the
        // compact variant: they go too.
        c->sparseGrad = false;
        c->symMetric = false;
        c->dev.symMetric = 0;
        c->deriveGrad = false;
        c->dev.deriveGrad = 0;
        c->deriveRow = false;
        c->dev.deriveRow = 0;
        sih_drop_graphs(c);
    }
}

extern "C" int sih_field_put(sih_ctx *c, const char *name, const void *host, size_t count)
{
    if (name && name[0] == '_') {
        sih_set_error("fields starting with '_' are library-internal");
        return SIH_ERR_ARG;
    }
    int rc = putget(c, name, const_cast<void *>(host), count, true);
    if (rc == SIH_OK) notePut(c, name);
    return rc;
}

extern "C" int sih_field_get(sih_ctx *c, const char *name, void *host, size_t count)
{
    return putget(c, name, host, count, false);
}

extern "C" int sih_fields_put(sih_ctx *c, int nFields, const char *const *names, const void *const *hosts,
                              const size_t *counts)
{
    for (int k = 0; names && k < nFields; k++)
        if (names[k] && names[k][0] == '_') {
            sih_set_error("fields starting with '_' are library-internal");
            return SIH_ERR_ARG;
        }
    int rc = putgetMany(c, nFields, names, const_cast<void *const *>(hosts), counts, true);
    for (int k = 0; rc == SIH_OK && k < nFields; k++) notePut(c, names[k]);
    return rc;
}

extern "C" int sih_fields_get(sih_ctx *c, int nFields, const char *const *names, void *const *hosts, const size_t *counts)
{
    return putgetMany(c, nFields, names, hosts, counts, false);
}

// ------------------------------------------------------------------------------------------ halo
static int uploadInts(int **dst, const std::vector<int> &v)
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

extern "C" int sih_halo_set_lists(sih_ctx *c, int location, int nSend, const int32_t *sendIdx, int nRecv,
                                  const int32_t *recvIdx, int nPeers, const int32_t *sendCounts,
                                  const int32_t *recvCounts)
{
    if (!c || location < 0 || location > 1 || nSend < 0 || nRecv < 0 || (nSend && !sendIdx) || (nRecv && !recvIdx) ||
        nPeers < 0 || (nPeers && (!sendCounts || !recvCounts))) {
        sih_set_error("sih_halo_set_lists: bad argument");
        return SIH_ERR_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    const std::vector<int> &inv = location == 0 ? c->cellInv : c->vertInv;
    int n = location == 0 ? c->nCells : c->nVertices;
    std::vector<int> s(nSend), r(nRecv);
    for (int i = 0; i < nSend; i++) {
        if (sendIdx[i] < 1 || sendIdx[i] > n) {
            sih_set_error("sih_halo_set_lists: send index out of range");
            return SIH_ERR_ARG;
        }
        s[i] = inv[sendIdx[i] - 1];
    }
    for (int i = 0; i < nRecv; i++) {
        if (recvIdx[i] < 1 || recvIdx[i] > n) {
            sih_set_error("sih_halo_set_lists: recv index out of range");
            return SIH_ERR_ARG;
        }
        r[i] = inv[recvIdx[i] - 1];
    }
    std::vector<int> ss(nSend, 0), sc(nSend, nSend), rs(nRecv, 0), rcn(nRecv, nRecv);
    if (nPeers > 0) {
        long ts = 0, tr = 0;
        for (int p = 0; p < nPeers; p++) {
            if (sendCounts[p] < 0 || recvCounts[p] < 0) {
                sih_set_error("sih_halo_set_lists: negative count");
                return SIH_ERR_ARG;
            }
            ts += sendCounts[p];
            tr += recvCounts[p];
        }
        if (ts != nSend || tr != nRecv) {
            sih_set_error("sih_halo_set_lists: counts do not add up to the list lengths");
            return SIH_ERR_ARG;
        }
        int o = 0;
        for (int p = 0; p < nPeers; p++) {
            for (int i = 0; i < sendCounts[p]; i++) {
                ss[o + i] = o;
                sc[o + i] = sendCounts[p];
            }
            o += sendCounts[p];
        }
        o = 0;
        for (int p = 0; p < nPeers; p++) {
            for (int i = 0; i < recvCounts[p]; i++) {
                rs[o + i] = o;
                rcn[o + i] = recvCounts[p];
            }
            o += recvCounts[p];
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    if (nPeers > 0) {
        c->hSendCounts[location].assign(sendCounts, sendCounts + nPeers);
        c->hRecvCounts[location].assign(recvCounts, recvCounts + nPeers);
    } else {
        c->hSendCounts[location].assign(1, nSend);
        c->hRecvCounts[location].assign(1, nRecv);
    }
    for (auto &kv : c->haloTables) {  // tables built for the previous lists are stale
        if (kv.second.d_planes) (void)hipFree(kv.second.d_planes);
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++)
                if (kv.second.d_base[a][b]) (void)hipFree(kv.second.d_base[a][b]);
    }
    c->haloTables.clear();
    c->hSend[location] = s;
    c->hRecv[location] = r;
    c->partsReady = false;  // subsets derived from the previous lists are stale
    int rc;
    if ((rc = uploadInts(&c->d_send[location], s))) return rc;
    if ((rc = uploadInts(&c->d_recv[location], r))) return rc;
    if ((rc = uploadInts(&c->d_segStart[location][0], ss))) return rc;
    if ((rc = uploadInts(&c->d_segCnt[location][0], sc))) return rc;
    if ((rc = uploadInts(&c->d_segStart[location][1], rs))) return rc;
    if ((rc = uploadInts(&c->d_segCnt[location][1], rcn))) return rc;
    c->nSend[location] = nSend;
    c->nRecv[location] = nRecv;
    return SIH_OK;
}

// all planes of one message group in a single launch: blockIdx.y = plane, planes[] = base pointers
template <typename T>
__global__ void k_pack_all(T *__restrict__ buf, T *const *__restrict__ planes, const int *__restrict__ idx,
                           const int *__restrict__ segStart, const int *__restrict__ segCnt, int n, int nf)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        int s0 = segStart[t], sc = segCnt[t], fi = blockIdx.y;
        buf[(size_t)s0 * nf + (size_t)fi * sc + (t - s0)] = planes[fi][idx[t]];
    }
}

template <typename T>
__global__ void k_unpack_all(const T *__restrict__ buf, T *const *__restrict__ planes, const int *__restrict__ idx,
                             const int *__restrict__ segStart, const int *__restrict__ segCnt, int n, int nf)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        int s0 = segStart[t], sc = segCnt[t], fi = blockIdx.y;
        planes[fi][idx[t]] = buf[(size_t)s0 * nf + (size_t)fi * sc + (t - s0)];
    }
}

// cell and vertex fields of one exchange in a single message and a single launch; per neighbour the
// message holds [cell planes][vertex planes]
template <typename T, bool PACK>
__global__ void k_halo_both(T *__restrict__ buf, T *const *__restrict__ planes, int nfC, const int *__restrict__ idxC,
                            const int *__restrict__ baseC, const int *__restrict__ cntC, int nC,
                            const int *__restrict__ idxV, const int *__restrict__ baseV,
                            const int *__restrict__ cntV, int nV)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int fi = blockIdx.y;
    size_t a;
    T *f;
    if (fi < nfC) {
        if (t >= nC) return;
        a = (size_t)baseC[t] + (size_t)fi * cntC[t];
        f = planes[fi] + idxC[t];
    } else {
        if (t >= nV) return;
        a = (size_t)baseV[t] + (size_t)(fi - nfC) * cntV[t];
        f = planes[fi] + idxV[t];
    }
    if (PACK)
        buf[a] = *f;
    else
        *f = buf[a];
}

static int haloBoth(sih_ctx *c, int nFields, const char *const *names, void *buf, bool pack)
{
    if (c->hSendCounts[0].size() != c->hSendCounts[1].size() || c->hSendCounts[0].empty()) {
        sih_set_error("sih_halo_pack/unpack: combined messages need cell and vertex lists over the same neighbours");
        return SIH_ERR_STATE;
    }
    if (!buf) {
        sih_set_error("sih_halo_pack/unpack: null buffer");
        return SIH_ERR_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    std::string key = "2";
    for (int f = 0; f < nFields; f++) {
        key += '|';
        key += names[f];
    }
    auto it = c->haloTables.find(key);
    if (it == c->haloTables.end()) {
        std::vector<void *> planes[2];
        bool isInt = false;
        for (int f = 0; f < nFields; f++) {
            SihField *fl = sih_find_field(c, names[f]);
            if (!fl || (fl->loc != LOC_CELL && fl->loc != LOC_VERTEX)) {
                sih_set_error(std::string("sih_halo_pack/unpack: not a cell or vertex field: ") + names[f]);
                return SIH_ERR_ARG;
            }
            if (f == 0) isInt = fl->isInt;
            if (fl->isInt != isInt) {
                sih_set_error("sih_halo_pack/unpack: fields of one call must share a type");
                return SIH_ERR_ARG;
            }
            const int pad = fl->loc == LOC_CELL ? c->padC : c->padV;
            for (int l = 0; l < fl->lead; l++)
                planes[fl->loc == LOC_CELL ? 0 : 1].push_back((char *)fl->d + (size_t)l * pad * (isInt ? 4 : 8));
        }
        sih_ctx::HaloTable tb;
        tb.isInt = isInt;
        tb.nPlanesCell = (int)planes[0].size();
        tb.nPlanes = (int)(planes[0].size() + planes[1].size());
        std::vector<void *> all(planes[0]);
        all.insert(all.end(), planes[1].begin(), planes[1].end());
        HIPCHK(hipMalloc((void **)&tb.d_planes, sizeof(void *) * std::max<size_t>(all.size(), 1)));
        HIPCHK(hipMemcpy(tb.d_planes, all.data(), sizeof(void *) * all.size(), hipMemcpyHostToDevice));
        const int nfC = (int)planes[0].size(), nfV = (int)planes[1].size();
        for (int dir = 0; dir < 2; dir++) {
            const std::vector<int> &cc = dir == 0 ? c->hSendCounts[0] : c->hRecvCounts[0];
            const std::vector<int> &cv = dir == 0 ? c->hSendCounts[1] : c->hRecvCounts[1];
            std::vector<int> bC, bV;
            long base = 0;
            for (size_t p = 0; p < cc.size(); p++) {
                for (int i = 0; i < cc[p]; i++) bC.push_back((int)(base + i));
                for (int i = 0; i < cv[p]; i++) bV.push_back((int)(base + (long)nfC * cc[p] + i));
                base += (long)nfC * cc[p] + (long)nfV * cv[p];
            }
            if (base > 2147483647L) {
                sih_set_error("sih_halo_pack/unpack: message too large");
                return SIH_ERR_ARG;
            }
            int rc;
            if ((rc = uploadInts(&tb.d_base[0][dir], bC))) return rc;
            if ((rc = uploadInts(&tb.d_base[1][dir], bV))) return rc;
        }
        it = c->haloTables.emplace(key, tb).first;
    }
    const sih_ctx::HaloTable &tb = it->second;
    const int dir = pack ? 0 : 1;
    const int nC = pack ? c->nSend[0] : c->nRecv[0], nV = pack ? c->nSend[1] : c->nRecv[1];
    if (std::max(nC, nV) == 0 || tb.nPlanes == 0) return SIH_OK;
    const int *idxC = pack ? c->d_send[0] : c->d_recv[0], *idxV = pack ? c->d_send[1] : c->d_recv[1];
    const int *cntC = c->d_segCnt[0][dir], *cntV = c->d_segCnt[1][dir];
    dim3 grid((std::max(nC, nV) + 255) / 256, tb.nPlanes), block(256);
    if (tb.isInt) {
        if (pack)
            hipLaunchKernelGGL((k_halo_both<int, true>), grid, block, 0, c->stream, (int *)buf, (int *const *)tb.d_planes,
                               tb.nPlanesCell, idxC, tb.d_base[0][dir], cntC, nC, idxV, tb.d_base[1][dir], cntV, nV);
        else
            hipLaunchKernelGGL((k_halo_both<int, false>), grid, block, 0, c->stream, (int *)buf,
                               (int *const *)tb.d_planes, tb.nPlanesCell, idxC, tb.d_base[0][dir], cntC, nC, idxV,
                               tb.d_base[1][dir], cntV, nV);
    } else {
        if (pack)
            hipLaunchKernelGGL((k_halo_both<double, true>), grid, block, 0, c->stream, (double *)buf,
                               (double *const *)tb.d_planes, tb.nPlanesCell, idxC, tb.d_base[0][dir], cntC, nC, idxV,
                               tb.d_base[1][dir], cntV, nV);
        else
            hipLaunchKernelGGL((k_halo_both<double, false>), grid, block, 0, c->stream, (double *)buf,
                               (double *const *)tb.d_planes, tb.nPlanesCell, idxC, tb.d_base[0][dir], cntC, nC, idxV,
                               tb.d_base[1][dir], cntV, nV);
    }
    HIPCHK(hipGetLastError());
    return SIH_OK;
}

static int haloPackUnpack(sih_ctx *c, int location, int nFields, const char *const *names, void *buf, bool pack)
{
    if (!c || location < 0 || location > 2 || nFields < 0 || !names) {
        sih_set_error("sih_halo_pack/unpack: bad argument");
        return SIH_ERR_ARG;
    }
    if (location == 2) return nFields ? haloBoth(c, nFields, names, buf, pack) : SIH_OK;
    int n = pack ? c->nSend[location] : c->nRecv[location];
    const int *idx = pack ? c->d_send[location] : c->d_recv[location];
    const int *sg0 = c->d_segStart[location][pack ? 0 : 1], *sgc = c->d_segCnt[location][pack ? 0 : 1];
    if (n == 0 || nFields == 0) return SIH_OK;
    if (!buf) {
        sih_set_error("sih_halo_pack/unpack: null buffer");
        return SIH_ERR_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    // the table of plane pointers of a field group is built once and kept on the device
    std::string key = std::to_string(location);
    for (int f = 0; f < nFields; f++) {
        key += '|';
        key += names[f];
    }
    auto it = c->haloTables.find(key);
    if (it == c->haloTables.end()) {
        std::vector<void *> planes;
        bool isInt = false;
        const int pad = location == 0 ? c->padC : c->padV;
        for (int f = 0; f < nFields; f++) {
            SihField *fl = sih_find_field(c, names[f]);
            if (!fl || (int)fl->loc != location) {
                sih_set_error(std::string("sih_halo_pack/unpack: not a field of this location: ") + names[f]);
                return SIH_ERR_ARG;
            }
            if (f == 0) isInt = fl->isInt;
            if (fl->isInt != isInt) {
                sih_set_error("sih_halo_pack/unpack: fields of one call must share a type");
                return SIH_ERR_ARG;
            }
            for (int l = 0; l < fl->lead; l++)
                planes.push_back((char *)fl->d + (size_t)l * pad * (isInt ? 4 : 8));
        }
        sih_ctx::HaloTable tb;
        tb.nPlanes = (int)planes.size();
        tb.isInt = isInt;
        HIPCHK(hipMalloc((void **)&tb.d_planes, sizeof(void *) * planes.size()));
        HIPCHK(hipMemcpy(tb.d_planes, planes.data(), sizeof(void *) * planes.size(), hipMemcpyHostToDevice));
        it = c->haloTables.emplace(key, tb).first;
    }
    const sih_ctx::HaloTable &tb = it->second;
    dim3 grid((n + 255) / 256, tb.nPlanes), block(256);
    if (tb.isInt) {
        if (pack)
            hipLaunchKernelGGL(k_pack_all<int>, grid, block, 0, c->stream, (int *)buf, (int *const *)tb.d_planes, idx,
                               sg0, sgc, n, tb.nPlanes);
        else
            hipLaunchKernelGGL(k_unpack_all<int>, grid, block, 0, c->stream, (const int *)buf,
                               (int *const *)tb.d_planes, idx, sg0, sgc, n, tb.nPlanes);
    } else {
        if (pack)
            hipLaunchKernelGGL(k_pack_all<double>, grid, block, 0, c->stream, (double *)buf,
                               (double *const *)tb.d_planes, idx, sg0, sgc, n, tb.nPlanes);
        else
            hipLaunchKernelGGL(k_unpack_all<double>, grid, block, 0, c->stream, (const double *)buf,
                               (double *const *)tb.d_planes, idx, sg0, sgc, n, tb.nPlanes);
    }
    HIPCHK(hipGetLastError());
    return SIH_OK;
}

extern "C" int sih_halo_pack(sih_ctx *c, int location, int nFields, const char *const *names, void *deviceBuf)
{
    return haloPackUnpack(c, location, nFields, names, deviceBuf, true);
}

extern "C" int sih_halo_unpack(sih_ctx *c, int location, int nFields, const char *const *names, const void *deviceBuf)
{
    return haloPackUnpack(c, location, nFields, names, const_cast<void *>(deviceBuf), false);
}

extern "C" int sih_get_info(sih_ctx *c, const char *key, double *value)
{
    if (!c || !key || !value) return SIH_ERR_ARG;
    std::string k = key;
    if (k == "sparseGradient") *value = c->sparseGrad ? 1.0 : 0.0;
    else if (k == "symmetricMetric") *value = c->symMetric ? 1.0 : 0.0;
    else if (k == "derivedGradient") *value = c->deriveGrad ? 1.0 : 0.0;
    else if (k == "derivedIntegralRow") *value = c->deriveRow ? 1.0 : 0.0;
    else if (k == "dampingTimescale") *value = c->dev.dampT;
    else if (k == "elasticTimeStep") *value = c->dev.dtE;
    else if (k == "dynamicsTimeStep") *value = c->dev.dtDyn;
    else if (k == "numericalInertiaCoefficient") *value = c->dev.brlx;
    else if (k == "padC") *value = c->padC;
    else if (k == "padV") *value = c->padV;
    else if (k == "minDvEdge") *value = c->minDvEdge;
    else if (k == "irCflWarnings") *value = sih_ir_cfl_warnings(c);
    else if (k == "irRecoverMeansWarnings") *value = sih_ir_recover_warnings(c);
    else if (k == "ridgeIterations") *value = sih_ridge_info(c);  // most iterations of ridge_ice in any cell, last call
    else if (k == "irFluxWalk") *value = sih_ir_flux_walk(c);
    else if (k == "splitCellKernel") *value = c->splitCells ? 1.0 : 0.0;
    else if (k == "persistentKernel") *value = sih_persist_usable(c) ? 1.0 : 0.0;
    else if (k == "persistentPerCU") *value = sih_persist_info(c, 0);
    else if (k == "persistentReason") *value = sih_persist_info(c, 1);
    else if (k == "persistentTiles") *value = sih_persist_info(c, 2);
    else if (k == "persistentArrived") *value = sih_persist_info(c, 3);
    else if (k == "persistentLaunches") *value = sih_persist_info(c, 4);
    else if (k == "persistentExchangeLaunches") *value = sih_persist_info(c, 5);
    else if (k == "persistentTileCells") *value = sih_persist_info(c, 7);
    else if (k == "iterationForm") *value = c->lastForm;
    else if (k == "fusedKernel") *value = sih_fused_usable(c) ? 1.0 : 0.0;
    else if (k == "fusedLaunches") *value = sih_persist_info(c, 6);
    else if (k == "persistentExchangeReady") *value = sih_persist_exchange_usable(c) && sih_live_contexts(c->device) == 1 ? 1.0 : 0.0;
    else {
        sih_set_error("sih_get_info: unknown key " + k);
        return SIH_ERR_ARG;
    }
    return SIH_OK;
}

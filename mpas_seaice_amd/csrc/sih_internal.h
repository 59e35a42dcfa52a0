// Internal definitions of libseaice_hip (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <cstdint>
#include <string>
#include <vector>

#include "seaice_hip.h"

// Allocation checker for hunting double frees and overlapping allocations (build with -DSIH_DEBUG_ALLOC): every
// hipMalloc / hipFree of the library goes through a registry that reports a free of an unknown or already freed
// pointer and an allocation that overlaps a live one, with file and line, on stderr.
#ifdef SIH_DEBUG_ALLOC
hipError_t sih_dbg_malloc(void **p, size_t n, const char *file, int line);
hipError_t sih_dbg_free(void *p, const char *file, int line);
#ifndef SIH_DEBUG_ALLOC_IMPL
#define hipMalloc(p, n) sih_dbg_malloc((void **)(p), (n), __FILE__, __LINE__)
#define hipFree(p) sih_dbg_free((void *)(p), __FILE__, __LINE__)
#endif
#endif

#define SIH_MAXE 10  // largest maxEdges the kernels are instantiated for
#define SIH_MAX_TRAPEZOID 21  // largest trapezoidal integration order: (n+1)(n+2)/2 = 253 points
#define SIH_MAX_RULE 256
#define SIH_MAXCAT 32  // most thickness categories the column package's ice strength handles

enum SihLoc { LOC_CELL = 0, LOC_VERTEX = 1, LOC_EDGE = 2 };

// A device-resident field: structure-of-arrays, element (l, i) of a field with `lead` leading
// entries per mesh entity lives at d[l * pad + i], i being the library's internal entity index.
struct SihField {
    std::string name;
    SihLoc loc;
    int lead;
    bool isInt;
    void *d;
};

// everything the time-step kernels need, by value
struct SihDev {
    int nCells, nCellsSolve, nVertices, nVerticesSolve, M, D, nCat;
    int padC, padV;
    // connectivity (internal numbering, 0-based; "none" = nCells / nVertices, a zero-filled slot)
    const int *nEdgesOnCell, *voc, *coc, *cov, *cvav, *gidx;
    const int *interiorVertex, *landIceMask, *landIceMaskVertex, *vActive;
    int *solveStress, *solveVelocity, *solveVelocityPrevious;
    // geometry
    const double *areaCell, *areaTriangle, *fVertex, *tanLat, *varDenom;
    const double *gradU, *gradV, *gradU3, *gradV3, *SU, *SV, *SM;
    // state
    double *u, *v, *uInit, *vInit;
    double *strain11, *strain22, *strain12, *stress11, *stress22, *stress12, *Prep, *prin1, *prin2;
    double *partU, *partV;
    double *divU, *divV, *icePressure;
    double *airCellU, *airCellV, *airVertU, *airVertV;
    double *uOcnVert, *vOcnVert, *ocnStressU, *ocnStressV, *ocnCoeff, *ocnStressCellU, *ocnStressCellV;
    double *tiltVertU, *tiltVertV, *tiltForceU, *tiltForceV, *massF;
    double *airTiltU, *airTiltV;  // airStressVertex + surfaceTiltForce, constant over the subcycles (formed in the pre-cycle)
    double *divergence, *shear, *ridgeConv, *ridgeShear;
    double *iceAreaVertex, *totalMassVertex, *iceAreaCellInitial, *totalMassCell;
    double *iceAreaCell, *iceVolumeCell, *snowVolumeCell;
    const double *iceAreaCat, *iceVolCat, *snowVolCat;
    const double *uOcn, *vOcn, *tiltU, *tiltV, *uAir, *vAir, *airDensity;
    const double *openWater;  // openWaterArea, read by the column package's ice strength
    // scalars
    double dtDyn, dtE, dampT, brlx;
    int relation, useOcean, quadratic, useAir, tiltMode, useColumn, haloVertices;
    int symMetric;  // basisIntegralsMetric(s, j) == (j, s) bit for bit: the cell kernels read one of each pair
    int deriveGrad; // where the three gradients met at a corner sum to zero to rounding the corner's own is not fetched
    int chunkCells;   // the split cell kernel maps its blocks to contiguous eighths of the cells per XCD
    int streamHints;  // the cell kernel's non-temporal marks: 1 stresses and partial sums (large blocks), 2 the basis arrays too (huge ones)
    int deriveRow;  // basisIntegralsU / V: the first velocity corner's row of a marked cell (bit 9 of nEdgesDerive) is minus the
                    // sum of the other rows, bit for bit, and the split cell kernel forms it on the chip instead of fetching it
    const int *nEdgesDerive;  // nEdgesOnCell | 256 for the cells where that holds at every corner | 512 (see deriveRow)
    int columnStrength, kstrength, krdgPartic, krdgRedist;  // colpkg_ice_strength in place of Hibler's formula
    double muRdg, Cf;
};

struct sih_ctx {
    int device = 0;
    hipStream_t ownStream = nullptr, stream = nullptr;
    int nCells = 0, nCellsSolve = 0, nVertices = 0, nVerticesSolve = 0, nEdges = 0, nEdgesSolve = 0;
    int M = 0, D = 0, onASphere = 0;
    double R = 1.0;
    int padC = 0, padV = 0, padE = 0;
    bool renumber = false;
    std::vector<int> cellPerm, cellInv, vertPerm, vertInv;  // perm: internal -> host (0-based); inv: host -> internal
    std::vector<int> edgePerm, edgeInv;                      // filled by sih_ir_init (identity until then)
    int *d_cellPerm = nullptr, *d_vertPerm = nullptr, *d_edgePerm = nullptr;
    struct sih_ir_state *ir = nullptr;                       // incremental-remap advection state
    struct sih_weak_state *weak = nullptr;                   // weak strain / stress-divergence scheme
    struct sih_upwind_state *upwind = nullptr;               // upwind advection state
    struct sih_persist_state *persist = nullptr;             // persistent subcycle kernel (small blocks)
    // iterations, launches, exchanges and launches-with-exchange the persistent kernel has counted on the device so
    // far: kept across a rebuild of its tables (sih_evp_init again, sih_evp_static_changed), because the tags and the
    // bank parity of the words in the transport's receive region go on counting from them
    unsigned long long persistCarry[4] = {0, 0, 0, 0};
    struct sih_ridge_state *ridge = nullptr;                 // ridging of the column package (column_ridge.hip)
    struct sih_comm_state *comm = nullptr;                   // halo exchange with blocks of other processes / devices
    std::map<std::string, SihField> fields;
    std::vector<void *> allocs;
    char *arenaBase = nullptr;  // current chunk of the device arena
    size_t arenaSize = 0, arenaUsed = 0;
    sih_evp_config cfg{};
    bool inited = false, sparseGrad = false;
    bool symMetric = false;  // the metric integrals are bitwise symmetric (checked on the device at set-up)
    bool deriveGrad = false; // compact gradients obey the partition of unity to rounding (checked likewise)
    bool deriveRow = false;  // rows of basisIntegralsU / V of marked cells stored as minus the sum of the other rows
    bool counted = false;  // in the per-device count of live contexts
    int persistMode = -1;  // sih_evp_set_persistent: -1 environment, 0 two-kernel form, 1 persistent kernel, 2 with the exchange inside
    bool splitCells = false;  // cell kernel with one thread per (cell, corner) instead of one per cell
    double minDvEdge = 0.0;
    SihDev dev{};
    double *d_rule = nullptr;  // integration rule of the Wachspress integrals
    void *d_stage = nullptr;
    size_t stageBytes = 0;
    // sih_fields_get_start: a second stream with a staging buffer of its own, so that a download can drain while the
    // context's stream computes.  evCopyFrom: what the download has to wait for (recorded on the context's stream);
    // evCopyGot: the layout kernels have read the fields (the context's stream waits for it before it goes on, so
    // nothing overwrites a field that is still being read); copyPending: copies are in flight into host arrays.
    hipStream_t copyStream = nullptr;
    hipEvent_t evCopyFrom = nullptr, evCopyGot = nullptr;
    void *d_stageCopy = nullptr;
    size_t stageCopyBytes = 0;
    bool copyPending = false;
    // host arrays that are not page-locked: the download goes by DMA into this page-locked buffer of the library and
    // sih_fields_get_finish copies it on into the host arrays (copyDst, copyOff, copyLen) with a few threads
    void *h_stage = nullptr;  // the same for the synchronous batched transfers (sih_fields_put / sih_fields_get)
    size_t hStageBytes = 0;
    std::vector<hipEvent_t> stageEvents;
    void *h_stageCopy = nullptr;
    size_t hStageCopyBytes = 0;
    std::vector<void *> copyDst;
    std::vector<size_t> copyOff, copyLen;
    int nSend[2] = {0, 0}, nRecv[2] = {0, 0};
    int *d_send[2] = {nullptr, nullptr}, *d_recv[2] = {nullptr, nullptr};
    // per entry: start offset and length of the neighbour segment it belongs to (send: [0], recv: [1])
    int *d_segStart[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}, *d_segCnt[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    struct HaloTable {
        void **d_planes = nullptr;
        int nPlanes = 0;
        bool isInt = false;
        // combined cell + vertex messages: planes of the cell fields first, then the vertex fields; per list
        // entry the position of its first plane in the message ([0] send side, [1] receive side)
        int nPlanesCell = 0;
        int *d_base[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [location][send / recv]
    };
    std::vector<int> hSendCounts[2], hRecvCounts[2];  // per neighbour, as given to sih_halo_set_lists
    std::vector<int> hSend[2], hRecv[2];              // the lists themselves, internal numbering
    std::vector<int> hVoc, hNe;                       // verticesOnCell (SoA, padC stride) / nEdgesOnCell, internal
    // entity subsets that let the halo exchange overlap with subcycle work (sih_evp_overlap_setup):
    // [0] cells whose strain needs no halo value, [1] the other cells, [2] vertices a neighbour needs
    // (send list), [3] the other vertices the vertex kernel updates
    int *d_part[4] = {nullptr, nullptr, nullptr, nullptr};
    int nPart[4] = {0, 0, 0, 0};
    bool partsReady = false;
    // special boundaries (sih_evp_set_special_boundaries): marked vertices, where their value comes from
    // (internal numbering) with sign +1 / -1 / 0, staging for the gather, replacement masks
    int nSb = 0;
    int *d_sbIdx = nullptr, *d_sbOrigin = nullptr, *d_sbSign = nullptr, *d_sbMaskV = nullptr, *d_sbMaskC = nullptr;
    double *d_sbTmp = nullptr;
    bool sbVelocity = false, sbMasks = false, sbPending = false;
    bool sbDeferred = false;  // velocity boundaries applied by sih_evp_apply_special_boundaries, not inside the iteration
    std::map<std::vector<int>, hipGraphExec_t> partGraphs;
    std::map<std::string, HaloTable> haloTables;  // plane-pointer tables of the field groups exchanged so far
    hipGraphExec_t graphExec = nullptr;
    int graphN = 0;
    int lastForm = 0;   // the form the last loop of iterations took: 0 two kernels, 1 persistent, 2 fused
    int graphForm = 0;  // what the graph holds: 0 the two kernels per iteration, 2 the fused kernel
    int *d_flag = nullptr;
    std::vector<hipEvent_t> events;
};

void sih_set_error(const std::string &s);
#define SIH_MAX_DEVICES 64
int sih_live_contexts(int device);  // contexts of this process on a device (2 for a device number out of range)
void sih_drop_graphs(sih_ctx *c);  // captured graphs no longer match what a launch would queue
SihField *sih_find_field(sih_ctx *c, const char *name);
int sih_add_field(sih_ctx *c, const char *name, SihLoc loc, int lead, bool isInt);
int sih_dev_alloc(sih_ctx *c, void **p, size_t bytes);
void sih_ir_destroy(sih_ctx *c);
// the advected tracers as sih_ir_set_tracers registered them (list order: parents before children), for the column
// package's ridging, which works on the same device fields
struct SihTracerInfo {
    std::string name;
    int parent, L;
    double *val;
};
bool sih_ir_tracer_table(sih_ctx *c, std::vector<SihTracerInfo> &out, int *nCategories);
void sih_ridge_destroy(sih_ctx *c);
double sih_ridge_info(sih_ctx *c);
void sih_weak_destroy(sih_ctx *c);
void sih_upwind_destroy(sih_ctx *c);
void sih_comm_free(sih_ctx *c);
void sih_persist_destroy(sih_ctx *c);
int sih_persist_setup(sih_ctx *c);
bool sih_persist_usable(sih_ctx *c);
int sih_persist_launch(sih_ctx *c, int n);
int sih_persist_status(sih_ctx *c);
double sih_persist_info(sih_ctx *c, int what);
// the same tiles as one launch per iteration, for blocks that cannot stay resident (evp_persist.hip)
#define SIH_FUSED_MAX_CELLS_DEFAULT 70000  // (88 k cells: two kernels 28.1 us an iteration, fused 28.9; 47 k cells: 18.2 / 17.7)
bool sih_fused_usable(sih_ctx *c);
int sih_fused_launch(sih_ctx *c, int m);
int sih_comm_push(sih_ctx *c, int nFields, const char *const *names, hipStream_t stream);
int sih_comm_pull(sih_ctx *c, int nFields, const char *const *names, hipStream_t stream);
int sih_comm_prepare(sih_ctx *c, int nFields, const char *const *names);  // builds the group's tables (not capturable)
bool sih_comm_ready(sih_ctx *c);
// The tagged region of the transport: 8-byte words {half a value | tag << 32} that carry their own arrival signal,
// written and polled from inside the persistent subcycle kernel.  A neighbour's segment holds, for the cells it
// sends, planes of words [stress 0..2][corner][low / high word][place] and, behind them, for its vertices
// [u, v][low / high word][place]; "place" is the entity's rank in the SENDER's own numbering among what it sends
// to that neighbour, so that the lanes of a sending wave, which hold consecutive cells, write consecutive words.
// The words exist SIH_TAGGED_COPIES times, in two banks: a launch of the persistent kernel writes exchange j of its
// own (1-based, at most SIH_TAGGED_COPIES / 2 a launch: longer loops are split) into copy j - 1 of the bank of its
// parity (launches with the exchange inside, counted from the set-up).  A copy is therefore written again two
// launches later at the earliest, and by then every tile of the reading block has read it: the writer, in launch
// L, has either received that block's words of launch L (exchange j >= 2) or finished its own launch L - 1, which
// ended with an exchange in which it received that block's words of launch L - 1 (j = 1); in both cases the
// reading block has started launch L - 1 at least, hence finished launch L - 2 as a whole.  (A first version used
// four copies in turn, arguing from what a block can compute before its neighbour has read; but tiles of one
// block may be several iterations apart, as many as there are tiles between them, and with an exchange after every
// iteration a slow tile found its words overwritten: wrong halo values on one of the test partitions.)
#define SIH_TAGGED_COPIES 32
struct sih_comm_tagged {
    int generation = 0, nRanks = 0;
    char *myBase = nullptr;  // copy 0 of this block's region
    size_t myHalfBytes = 0;  // bytes of one copy
    std::vector<size_t> myWordOff;                   // first word of the segment for messages from rank p
    std::vector<char *> peerBase;                    // rank p's region (half 0) in this address space
    std::vector<size_t> peerHalfBytes, peerWordOff;  // its half size; first word of this rank's segment there
    std::vector<int> sendPos[2], recvPos[2];         // per entry of the send / receive lists (cells, vertices): its place
};
bool sih_comm_tagged_info(sih_ctx *c, sih_comm_tagged *out);
// n iterations in the persistent kernel with the halo update of velocities and stresses inside it, after every
// interval-th iteration and after the last; SIH_ERR_UNSUPPORTED when the block or the transport does not allow it
int sih_persist_launch_exchange(sih_ctx *c, int n, int interval);
bool sih_persist_exchange_usable(sih_ctx *c);
hipStream_t sih_comm_stream(sih_ctx *c);
void sih_comm_events(sih_ctx *c, hipEvent_t *fork, hipEvent_t *join);
int sih_setup_edges(sih_ctx *c, const sih_ir_desc *m, const char *who);
int sih_weak_init(sih_ctx *c);
void sih_weak_bind(sih_ctx *c);
void sih_weak_launch_pre(sih_ctx *c);
void sih_weak_launch_cells(sih_ctx *c);
void sih_weak_launch_hybrid_strain(sih_ctx *c);
void sih_weak_launch_post(sih_ctx *c);
int sih_ir_cfl_warnings(sih_ctx *c);
int sih_ir_recover_warnings(sih_ctx *c);
int sih_ir_flux_walk(sih_ctx *c);
int sih_hip_fail(hipError_t e, const char *what, const char *file, int line);
#define HIPCHK(x)                                                             \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) return sih_hip_fail(e_, #x, __FILE__, __LINE__); \
    } while (0)

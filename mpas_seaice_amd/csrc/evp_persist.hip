// The subcycle loop of a small block as ONE launch: a persistent kernel in which every workgroup keeps a tile of
// 64 cells, and every vertex those cells touch, on the chip for all iterations.
//
// Why: with the two-kernel form (evp_step.hip) a block of a few ten thousand cells -- what one GPU holds when
// QU60 is cut eight ways, or the whole of a coarse mesh -- spends its time on the two kernel boundaries per
// iteration and on re-reading its basis tables, which never change.  Here a thread keeps the basis gradients of
// its cell corner, the corner's stresses and the constants of one vertex in registers and the corner's rows of
// the three integral matrices in LDS.  A tile solves the velocities of ALL vertices of its cells itself (those
// it shares with neighbouring tiles are solved by each of them, from the same operands, to the same bits), so
// the velocities never leave the chip: per iteration only the per-corner partial sums of the stress divergence
// pass through memory, and a vertex waits for exactly the sums it reads, never for the whole grid -- one hand-off
// per iteration, with the arrival of the data as its own signal.
//
// The arithmetic is that of k_sub_cells_split / k_sub_vertices, statement for statement, so the results are
// the same bits.  Iterations that store the end-of-step diagnostics (the last one of a dynamics step) stay with
// the two-kernel form.
//
// Three forms share the tiles: the persistent kernel with tiles of 64 cells (all integral rows in LDS, two workgroups
// a CU: 32 768 cells on 256 CUs), the same kernel with tiles of 42 cells and the rows of the metric integrals read
// from memory in every iteration (four workgroups a CU: 43 008 cells), and, for blocks that cannot stay resident or
// are not alone on their device, k_sub_fused: one launch per iteration, nothing resident, nothing waits (below).
//
// Hand-off rules (per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed):
//   - the sums live in four buffers used in turn (iteration number modulo 4).  In iteration t a corner drains its
//     earlier stores (s_waitcnt vmcnt(0)), stores its sums of iteration t into buffer t and then a marker (a NaN bit
//     pattern that arithmetic never produces) into its slot of buffer t+2; all of these are agent-scope atomic
//     stores (write-through).  The marker of buffer t+1 was stored one iteration earlier, so the drain finds it
//     done and costs nothing;
//   - a vertex reads its three sums of iteration t with agent-scope atomic loads (L1 bypassed) again and again
//     until none of them is the marker.  It cannot meet older data: it read the same three slots of buffer t-1
//     one iteration ago, and their writers had the marker of buffer t acknowledged before they stored those;
//   - a corner may overwrite buffer t+2 (the sums of iteration t-2): a reader of those is a tile that shares the
//     vertex, whose sums of iteration t-1 this tile has read since, and that tile stored them only after all its
//     vertices had finished iteration t-2;
//   - every wait is bounded by the wall clock; a wait that runs out sets a status word, every workgroup leaves
//     the loop at the end of its iteration, and the host reports it.  All workgroups must be resident together:
//     the launch is sized by the occupancy the runtime reports, the kernel begins with a census (every workgroup
//     marks its slots of the first buffer, checks in -- 32 counters in separate cache lines, so that the check-ins
//     do not queue behind one address -- and waits for the others), and a probe launch at set-up settles whether
//     this grid fits before it is ever used.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "sih_internal.h"

#define ECC2 4.0
#define PUNY 1.0e-11
#define RHOW 1026.0
#define DRAGIO 0.00536
#define SIN_TURN 0.0
#define COS_TURN 1.0

// words of the state array; the census counters sit 128 bytes apart behind the first line
enum { PS_EPOCH = 0, PS_LAUNCHES = 1, PS_STATUS = 2, PS_TICKS = 3, PS_XSEQ = 4, PS_XTICKS = 5, PS_XLAUNCHES = 6, PS_CENSUS = 16, PS_CENSUS_N = 32, PS_CENSUS_STRIDE = 16,
       PS_WORDS = PS_CENSUS + PS_CENSUS_N * PS_CENSUS_STRIDE };
#define PS_RING 4  // buffers of the partial sums, used in turn

#define PS_MARKER 0xFFF8C0DEDEADBEEFULL  // "not written yet": a NaN with a payload no operation produces

struct PersistArgs {
    int nTiles;
    const int *tileVertStart, *tileVerts;  // vertices of a tile's cells: tileVerts[tileVertStart[t] .. tileVertStart[t+1])
    const int *tileVertOwned;              // same indexing: 1 where this tile is the one that stores the vertex at the end
    const int *cornerSlot;                 // per (corner, cell) [g * padC + c]: position of its vertex in the tile's list
    unsigned long long *part;              // the partial sums: [buffer 0..PS_RING-1][U, V][corner * padC + cell]
    unsigned long long *state;             // PS_* words
};

// Tables of the in-kernel halo exchange (a copy lives in device memory; the kernel takes a pointer, so that the
// launch without exchange pays nothing for it).  Velocities and stresses travel as in the exchange between launches
// (uVelocity, vVelocity, stress11/22/12 of every list entry), but each value is two 8-byte words
// {low 32 bits | tag << 32}, {high 32 bits | tag << 32} with tag = number of the exchange: a word is there when its
// tag is, no flag and no fence.  The words exist in two banks of copies (sih_internal.h: a launch writes exchange j of
// its own into copy j - 1 of the bank of its parity).  They are laid out in planes ordered by
// the sender's own numbering, so that a sending wave, whose lanes hold consecutive cells, writes consecutive words
// (whole lines on the link, not one 8-byte packet a value).  A thread sends what its own corner
// (or the vertex it stores at the end) holds, straight from its registers into the neighbour's memory, and goes
// on; a thread whose corner or vertex is in a receive list polls its own slots.  Tiles away from the block's edge
// never wait for a neighbour.
struct PersistXchg {
    const int *tileActive;                         // per tile: it sends or receives something
    // what a cell sends: destinations [cellDestStart[c], cellDestStart[c+1]), each the address of the cell's first
    // word (stress11, corner 0, low word) in copy 0 of a neighbour's region, the size of one copy, and the
    // neighbour's count of received cells: the words are planes of that many, [stress][corner][low / high]
    const int *cellDestStart;
    const unsigned long long *cellDestAddr, *cellDestHalf;
    const int *cellDestCnt;
    // what the entries of tileVerts send (only the copy that is stored at the end has destinations): the low word
    // of uVelocity, the size of one copy, the neighbour's count of received vertices (planes [u, v][low / high])
    const int *vertDestStart;
    const unsigned long long *vertDestAddr, *vertDestHalf;
    const int *vertDestCnt;
    const int *recvCBase, *recvCStride;            // per (corner, cell) [g * padC + c]: first word (-1: none), words in a plane
    const int *recvVBase, *recvVStride;            // per entry of tileVerts
    const char *myBase;                            // this block's tagged region, copy 0
    unsigned long long myHalf;
    const double *stress11, *stress22, *stress12;  // for the corners the kernel does not update
};

// what the kernel reads of SihDev, and nothing else: the fewer scalar registers the arguments take, the more
// vector registers are left for the state a thread keeps
struct PersistDev {
    int nCells, nVerticesSolve, padC, padV, relation, useOcean, quadratic, haloVertices, symMetric;
    int deriveRow;  // the arithmetic of k_sub_cells_split with derived integral rows; nEdgesOnCell then carries the marks
    double dtE, dampT, brlx, dtDyn;
    const int *nEdgesOnCell, *voc, *gidx, *vActive, *solveStress, *solveVelocity;
    const double *tanLat, *varDenom, *gradU3, *gradV3, *SU, *SV, *SM, *icePressure;
    const double *totalMassVertex, *massF, *airTiltU, *airTiltV;
    const double *uOcnVert, *vOcnVert, *iceAreaVertex, *uInit, *vInit;
    double *u, *v, *stress11, *stress22, *stress12;
};

static PersistDev makePersistDev(const SihDev &d)
{
    PersistDev q;
    q.nCells = d.nCells; q.nVerticesSolve = d.nVerticesSolve; q.padC = d.padC; q.padV = d.padV;
    q.relation = d.relation; q.useOcean = d.useOcean; q.quadratic = d.quadratic; q.haloVertices = d.haloVertices;
    q.symMetric = d.symMetric;
    q.deriveRow = d.deriveRow;
    q.dtE = d.dtE; q.dampT = d.dampT; q.brlx = d.brlx; q.dtDyn = d.dtDyn;
    q.nEdgesOnCell = d.deriveRow ? d.nEdgesDerive : d.nEdgesOnCell; q.voc = d.voc; q.gidx = d.gidx; q.vActive = d.vActive;
    q.solveStress = d.solveStress; q.solveVelocity = d.solveVelocity;
    q.tanLat = d.tanLat; q.varDenom = d.varDenom; q.gradU3 = d.gradU3; q.gradV3 = d.gradV3;
    q.SU = d.SU; q.SV = d.SV; q.SM = d.SM; q.icePressure = d.icePressure;
    q.totalMassVertex = d.totalMassVertex; q.massF = d.massF; q.airTiltU = d.airTiltU; q.airTiltV = d.airTiltV;
    q.uOcnVert = d.uOcnVert; q.vOcnVert = d.vOcnVert; q.iceAreaVertex = d.iceAreaVertex; q.uInit = d.uInit; q.vInit = d.vInit;
    q.u = d.u; q.v = d.v; q.stress11 = d.stress11; q.stress22 = d.stress22; q.stress12 = d.stress12;
    return q;
}

struct sih_persist_state {
    PersistArgs a{};
    int capacity = 0;  // workgroups the device holds at once
    int arrived = 0;
    int perCU = 0, reason = 0;  // why not usable: 1 option, 2 too many tiles, 3 vertices per tile, 5 probe
    bool usable = false;
    bool fusedReady = false;  // the tile tables exist: the one-launch-per-iteration form (k_sub_fused) can run
    int fusedLaunches = 0;
    double *uAlt = nullptr, *vAlt = nullptr;  // the second copy of the velocities that form alternates with
    const int *tvVert = nullptr, *tvOwned = nullptr;  // the tiles' vertex lists padded to 64 M entries a tile (-1: none)
    int *tvQ = nullptr;                               // [3][tiles * 64 M]: where such a vertex finds its three partial sums
    const int *fusedSlot = nullptr;                   // per (corner, cell): place of its vertex in the list of its 64-cell tile
    int fusedTiles = 0;
    int tileCells = 64;                               // cells of a tile of the persistent kernel
    bool unchecked = false;  // a launch since the status word was last read
    std::vector<void *> allocs;
    // the tiles' vertex lists on the host, for the tables of the in-kernel exchange
    std::vector<int> hVStart, hVerts, hOwned;
    int launches = 0, xLaunches = 0;  // launches so far, and those of them with the exchange inside
    int xGeneration = -1;  // generation of the transport the exchange tables were built for (-1: none)
    bool xUsable = false;
    PersistXchg *d_x = nullptr;
    std::vector<void *> xAllocs;
};

__device__ inline unsigned long long ldAgent(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void stAgent(unsigned long long *p, unsigned long long x)
{
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline unsigned long long ldSystem(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline unsigned int ldSystem32(const unsigned int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline void stSystem(unsigned long long *p, unsigned long long x)
{
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline unsigned long long bitsOf(double x) { return (unsigned long long)__double_as_longlong(x); }
__device__ inline double valueOf(unsigned long long b) { return __longlong_as_double((long long)b); }
__device__ inline double sign1p(double x) { return copysign(1.0, x); }

// At most 128 vector registers a lane (launch bound of four waves a SIMD), so that two workgroups of 6 waves are
// resident on a CU wherever the hardware starts placing their waves; at 3 waves a SIMD the second workgroup was
// not admitted.
// LDS of one workgroup in doubles: staging of corner velocities and stresses (5 M 64), the tile's vertex velocities
// (2 * 64 M), the integral rows (3 M * 64 M).  Requested at launch, not declared statically: with the size in the
// kernel the compiler takes the two workgroups a CU's LDS holds for the occupancy and lets the registers grow past
// 128 a lane, where the hardware no longer admits the second workgroup.
// TC: cells of a tile (64, or 42: four waves a workgroup, four workgroups a CU).  STREAM: the rows of the metric
// integrals are not kept in LDS but read from memory (L2) in every iteration: two thirds of the LDS a cell, so that
// four workgroups of 42 cells fit a CU: 43 008 cells instead of 32 768.
template <int M, int TC, bool STREAM>
constexpr int persistLdsDoubles() { return 5 * M * TC + 2 * TC * M + (STREAM ? 2 : 3) * M * TC * M; }

template <int M, int TC, bool XCHG, bool STREAM>
__global__ void __launch_bounds__(TC * M, 4) k_sub_persistent(PersistDev d, PersistArgs p, int nIter,
                                                             const PersistXchg *__restrict__ xp, int xInterval, int xCells)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double(*su)[TC] = (double(*)[TC])(smem);
    double(*sv)[TC] = (double(*)[TC])(smem + M * TC);
    double(*s11)[TC] = (double(*)[TC])(smem + 2 * M * TC);
    double(*s22)[TC] = (double(*)[TC])(smem + 3 * M * TC);
    double(*s12)[TC] = (double(*)[TC])(smem + 4 * M * TC);
    double *uT = smem + 5 * M * TC, *vT = uT + TC * M;       // velocities of the tile's vertices
    double(*sInt)[TC * M] = (double(*)[TC * M])(vT + TC * M);  // this thread's rows of basisIntegralsU / V (/ Metric)
    __shared__ int okShared, sDead;
    const int lane = threadIdx.x, g = threadIdx.y, tid = g * TC + lane;
    // consecutive tiles on one XCD (workgroups are dealt round-robin over the eight XCDs): neighbouring tiles then
    // mostly hand their sums over through one L2
    const int nT = p.nTiles;
    int tile;
    {
        const int b = blockIdx.x, q = nT / 8, r = nT % 8, x = b % 8;
        tile = x * q + (x < r ? x : r) + b / 8;
    }
    // time limits (read where a wait looks at the clock, every eighth turn: no register is held for them): of the
    // census, and of the waits for sums and halo values.  With the exchange inside, a tile that waits for sums may
    // be waiting, through its neighbours, for another block that entered its step later: the longer limit.
    constexpr int LIMIT_WORD = XCHG ? PS_XTICKS : PS_TICKS;
    const unsigned long long launches = ldAgent(&p.state[PS_LAUNCHES]);
    const unsigned long long epoch = ldAgent(&p.state[PS_EPOCH]);  // iterations done by earlier launches
    const unsigned long long xseq0 = XCHG ? ldAgent(&p.state[PS_XSEQ]) : 0ULL;  // exchanges done by earlier launches
    const unsigned long long xl0 = XCHG ? ldAgent(&p.state[PS_XLAUNCHES]) : 0ULL;  // earlier launches with the exchange inside

    // ---- what this thread keeps for all iterations: its cell corner ...
    const int P = d.padC, PV = d.padV;
    int c = tile * TC + lane;
    const bool inRange = c < d.nCells;
    if (!inRange) c = 0;
    const size_t k = (size_t)g * P + c;
    const bool solve = inRange && d.solveStress[c] == 1;
    const int Ev = inRange ? d.nEdgesOnCell[c] : 0;  // (with the set-up's marks above bit 7 when deriveRow is set)
    const int E = Ev & 255;
    const bool corner = solve && g < E;
    // corner 0 of this cell takes its sums from the other corners' (k_sub_cells_split has the reasons)
    const bool cellDerives = solve && d.deriveRow && (Ev & 512);
    const int slot = corner ? p.cornerSlot[k] : 0;
    double tg = 0.0, P_ice = 0.0, o11 = 0.0, o22 = 0.0, o12 = 0.0;
    double gu3[3] = {0.0, 0.0, 0.0}, gv3[3] = {0.0, 0.0, 0.0};
    int i012 = 0;  // i0 | i1 << 8 | i2 << 16
#pragma unroll
    for (int s = 0; s < M; s++) {
        double x = 0.0, y = 0.0, z = 0.0;
        if (corner && s < E) {
            const size_t kb = (size_t)(g * M + s) * P + c;
            x = d.SU[kb];
            y = d.SV[kb];
            if (!STREAM) z = d.SM[kb];
        }
        sInt[s][tid] = x;
        sInt[M + s][tid] = y;
        if (!STREAM) sInt[2 * M + s][tid] = z;
    }
    if (corner) {
        tg = d.tanLat[d.voc[k]];
        P_ice = d.icePressure[c];
        o11 = d.stress11[k];
        o22 = d.stress22[k];
        o12 = d.stress12[k];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            gu3[q] = d.gradU3[(size_t)(q * M + g) * P + c];
            gv3[q] = d.gradV3[(size_t)(q * M + g) * P + c];
        }
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
        i012 = i0 | (i1 << 8) | (i2 << 16);
    }
    // ... and one vertex of the tile's cells
    const int v0 = p.tileVertStart[tile], nv = p.tileVertStart[tile + 1] - v0;
    const bool hasV = tid < nv;
    const int v = hasV ? p.tileVerts[v0 + tid] : 0;
    const bool owned = hasV && p.tileVertOwned[v0 + tid] != 0;
    // the vertices the vertex kernel updates: k_sub_vertices' conditions
    const bool solV = hasV && (d.haloVertices ? d.vActive[v] != 0 : v < d.nVerticesSolve) && d.solveVelocity[v] == 1;
    int q0 = 0, q1 = 0, q2 = 0;
    double den0 = 1.0, m = 0.0, mf = 0.0, atU = 0.0, atV = 0.0;  // (air stress + tilt force, as k_sub_vertices takes them)
    double uo = 0.0, vo = 0.0, aV = 0.0, uu = 0.0, vv = 0.0;
    if (hasV) {
        uu = d.u[v];
        vv = d.v[v];
    }
    if (solV) {
        q0 = d.gidx[v];
        q1 = d.gidx[PV + v];
        q2 = d.gidx[2 * PV + v];
        den0 = d.varDenom[v];
        m = d.totalMassVertex[v];
        mf = d.massF[v];
        atU = d.airTiltU[v];
        atV = d.airTiltV[v];
        uo = d.uOcnVert[v];
        vo = d.vOcnVert[v];
        aV = d.useOcean ? d.iceAreaVertex[v] : 0.0;
    }
    uT[tid] = uu;
    vT[tid] = vv;
    if (tid == 0) sDead = 0;
    const size_t plane = (size_t)M * P;  // one buffer holds 2 planes of this size: U, then V
    // ---- census: every workgroup of the launch is resident, and has marked its slots of the first buffer, before
    //      anything is computed
    {
        unsigned long long *b1 = p.part + ((epoch + 1ULL) % PS_RING) * 2ULL * plane;
        unsigned long long *b2 = p.part + ((epoch + 2ULL) % PS_RING) * 2ULL * plane;
        if (corner) {
            stAgent(&b1[k], PS_MARKER);
            stAgent(&b1[plane + k], PS_MARKER);
            stAgent(&b2[k], PS_MARKER);
            stAgent(&b2[plane + k], PS_MARKER);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid < 64) {
        bool good = ldAgent(&p.state[PS_STATUS]) == 0ULL;
        if (tid == 0)
            __hip_atomic_fetch_add(&p.state[PS_CENSUS + (blockIdx.x % PS_CENSUS_N) * PS_CENSUS_STRIDE], 1ULL,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // lane j watches counter j: the workgroups with blockIdx % 32 == j check in there
        const int j = tid % PS_CENSUS_N;
        const unsigned long long want =
            (launches + 1ULL) * (unsigned long long)(nT / PS_CENSUS_N + (j < nT % PS_CENSUS_N ? 1 : 0));
        const long long t0 = wall_clock64();
        while (good) {
            const unsigned long long have = ldAgent(&p.state[PS_CENSUS + j * PS_CENSUS_STRIDE]);
            if (__all(have >= want)) break;
            if ((unsigned long long)(wall_clock64() - t0) > ldAgent(&p.state[PS_TICKS]) || ldAgent(&p.state[PS_STATUS]) != 0ULL) {
                if (tid == 0)
                    __hip_atomic_fetch_max(&p.state[PS_STATUS], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = false;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (tid == 0) okShared = good ? 1 : 0;
    }
    __syncthreads();
    const bool alive = okShared != 0;
    // every workgroup read the two words before it checked in, so one of them may move them on for the next launch
    if (alive && tile == 0 && tid == 0) {
        stAgent(&p.state[PS_EPOCH], epoch + (unsigned long long)nIter);
        stAgent(&p.state[PS_LAUNCHES], launches + 1ULL);
        if (XCHG) {
            stAgent(&p.state[PS_XSEQ], xseq0 + (unsigned long long)((nIter + xInterval - 1) / xInterval));
            stAgent(&p.state[PS_XLAUNCHES], xl0 + 1ULL);
        }
    }
    unsigned int xDone = 0;  // exchanges of this launch so far
    const double dtE = d.dtE, T = d.dampT, brlx = d.brlx, dtD = d.dtDyn;
    const int relation = d.relation;
    int sl = (int)(epoch % PS_RING);  // buffer of the iteration before the one at hand
    // The pause between two looks at the partial sums, in units of 512 clocks: the more tiles poll, the more their
    // loads stand in the way of the stores they wait for.  Measured per iteration with pauses of 64 / 512 / 1024 /
    // 2048 / 4096 clocks: 221 tiles 4.70 / 4.67 / 4.75 / 5.05 / 5.77 us, 392 tiles 7.39 / 6.99 / 6.85 / 6.87 / 6.73,
    // 976 small tiles 10.84 / 10.22 / 9.89 / 9.45 / 9.22.
    const int napReps = nT <= 256 ? 1 : nT <= 512 ? 2 : 6;

    for (int it = 1; it <= nIter && alive; it++) {
        sl = (sl + 1) % PS_RING;
        unsigned long long *pU = p.part + (size_t)sl * 2 * plane, *pV = pU + plane;
        // ================= cells: strain, stress, partial sums of the divergence (k_sub_cells_split)
        const double ug = uT[slot], vg = vT[slot];
        su[g][lane] = corner ? ug : 0.0;
        sv[g][lane] = corner ? vg : 0.0;
        // the streamed rows of the metric integrals are requested here and used at the end of the cell phase: they
        // travel (from L2) while strain and stress are computed (requested where they are used: 13.4 instead of 10.7 us
        // an iteration at QU120; requested one iteration ahead, across the wait for the sums: 14.3)
        double bmS[M];
        if (STREAM) {
#pragma unroll
            for (int s = 0; s < M; s++) bmS[s] = corner ? d.SM[(size_t)(g * M + s) * P + c] : 0.0;
        }
        __syncthreads();
        double a11 = 0.0, a22 = 0.0, a12 = 0.0;
        if (corner) {
            double e11 = 0.0, e22 = 0.0, e12 = 0.0;
            const int i0 = i012 & 255, i1 = (i012 >> 8) & 255, i2 = i012 >> 16;
            const double xu[3] = {su[i0][lane], su[i1][lane], su[i2][lane]};
            const double xv[3] = {sv[i0][lane], sv[i1][lane], sv[i2][lane]};
#pragma unroll
            for (int q = 0; q < 3; q++) {
                e11 = e11 + xu[q] * gu3[q];
                e22 = e22 + xv[q] * gv3[q];
                e12 = e12 + 0.5 * (xu[q] * gv3[q] + xv[q] * gu3[q]);
            }
            e11 = e11 - vg * tg;
            e12 = e12 + ug * tg * 0.5;
            if (relation == 3) {
                a11 = 1.0 * e11;
                a22 = 1.0 * e22;
                a12 = 1.0 * e12;
            } else {
                a11 = o11;
                a22 = o22;
                a12 = o12;
                double eD = e11 + e22, eT = e11 - e22, eS = e12 * 2.0;
                double st1 = a11 + a22, st2 = a11 - a22;
                double Delta = sqrt(eD * eD + (eT * eT + eS * eS) / ECC2);
                double pc = P_ice / fmax(Delta, PUNY);
                double den;
                if (relation == 1) {
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
            o11 = a11;
            o22 = a22;
            o12 = a12;
        }
        s11[g][lane] = a11;
        s22[g][lane] = a22;
        s12[g][lane] = a12;
        __syncthreads();
        double cu = 0.0, cv = 0.0;
        if (corner && !d.deriveRow) {
#pragma unroll
            for (int s = 0; s < M; s++) {
                if (s < E) {
                    const double x11 = s11[s][lane], x22 = s22[s][lane], x12 = s12[s][lane];
                    const double bu = sInt[s][tid], bv = sInt[M + s][tid], bm = STREAM ? bmS[s] : sInt[2 * M + s][tid];
                    cu = cu - x11 * bu - x12 * bv - x12 * bm * tg;
                    cv = cv - x22 * bv - x12 * bu + x11 * bm * tg;
                }
            }
        }
        if (d.deriveRow) {  // (the same for every thread of the launch)
            // k_sub_cells_split's arithmetic with derived rows: the terms with basisIntegralsU / V and the metric terms
            // added up separately; corner 0 of a marked cell takes minus the sum of the other corners' former
            double nu = 0.0, nv = 0.0, mu = 0.0, mv = 0.0;
            if (corner) {
#pragma unroll
                for (int s = 0; s < M; s++) {
                    if (s < E) {
                        const double x11 = s11[s][lane], x22 = s22[s][lane], x12 = s12[s][lane];
                        const double bu = sInt[s][tid], bv = sInt[M + s][tid], bm = STREAM ? bmS[s] : sInt[2 * M + s][tid];
                        mu = mu - x12 * bm * tg;
                        mv = mv + x11 * bm * tg;
                        nu = nu - x11 * bu - x12 * bv;
                        nv = nv - x22 * bv - x12 * bu;
                    }
                }
                cu = nu + mu;
                cv = nv + mv;
                if (cellDerives && g > 0) {  // (su / sv: the velocities in them were taken before the barrier above)
                    su[g][lane] = nu;
                    sv[g][lane] = nv;
                }
            }
            __syncthreads();
            if (corner && cellDerives && g == 0) {
                double du = su[1][lane], dv = sv[1][lane];
#pragma unroll
                for (int j = 2; j < M; j++) {
                    if (j < E) {
                        du = du + su[j][lane];
                        dv = dv + sv[j][lane];
                    }
                }
                cu = mu - du;
                cv = mv - dv;
            }
        }
        if (corner) {
            // the markers of the next buffer (stored one iteration ago) are in memory before these sums can be seen
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stAgent(&pU[k], bitsOf(cu));
            stAgent(&pV[k], bitsOf(cv));
            // ... and the buffer after the next takes its markers now: it holds the sums of two iterations ago, which
            // every reader has finished with (this tile has since read what those readers computed from them), and the
            // stores have a whole iteration to land before the wait above comes round again
            unsigned long long *nU = p.part + (size_t)((sl + 2) % PS_RING) * 2 * plane;
            stAgent(&nU[k], PS_MARKER);
            stAgent(&nU[plane + k], PS_MARKER);
        }

        // ================= vertices: divergence, ocean stress coefficient, velocity solve (k_sub_vertices), for
        // every vertex of the tile's cells, as soon as its three pairs of sums have arrived
        unsigned long long bu0 = 0, bu1 = 0, bu2 = 0, bv0 = 0, bv1 = 0, bv2 = 0;
        {
            long long t0 = 0;
            for (int spins = 1;; spins++) {
                bool ready = true;
                if (solV) {
                    bu0 = ldAgent(&pU[q0]);
                    bu1 = ldAgent(&pU[q1]);
                    bu2 = ldAgent(&pU[q2]);
                    bv0 = ldAgent(&pV[q0]);
                    bv1 = ldAgent(&pV[q1]);
                    bv2 = ldAgent(&pV[q2]);
                    ready = bu0 != PS_MARKER && bu1 != PS_MARKER && bu2 != PS_MARKER && bv0 != PS_MARKER &&
                            bv1 != PS_MARKER && bv2 != PS_MARKER;
                }
                if (__all(ready)) break;
                if ((spins & 7) == 0) {
                    const long long now = wall_clock64();
                    if (t0 == 0) {
                        t0 = now;
                    } else if ((unsigned long long)(now - t0) > ldAgent(&p.state[LIMIT_WORD]) || ldAgent(&p.state[PS_STATUS]) != 0ULL) {
                        // give up: this iteration runs to its end on whatever was read, then everyone leaves
                        if (lane == 0) {
                            __hip_atomic_fetch_max(&p.state[PS_STATUS], 2ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            sDead = 1;
                        }
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(8);
                for (int nap = 1; nap < napReps; nap++) __builtin_amdgcn_s_sleep(8);
            }
        }
        if (solV) {
            double dU = 0.0, dV = 0.0;
            dU = dU + valueOf(bu0);
            dV = dV + valueOf(bv0);
            dU = dU + valueOf(bu1);
            dV = dV + valueOf(bv1);
            dU = dU + valueOf(bu2);
            dV = dV + valueOf(bv2);
            dU = dU / den0;
            dV = dV / den0;
            double coeff = 0.0;
            if (d.useOcean) {
                if (d.quadratic) {
                    double du = uo - uu, dv = vo - vv;
                    coeff = DRAGIO * RHOW * aV * sqrt(du * du + dv * dv);
                } else {
                    coeff = DRAGIO * RHOW * aV;
                }
            }
            double l11, l22, r1, r2;
            const double l12 = -mf - coeff * SIN_TURN * sign1p(mf);
            const double l21 = mf + coeff * SIN_TURN * sign1p(mf);
            if (relation == 2) {
                l11 = (brlx + 1.0) * (m / dtD) + coeff * COS_TURN;
                l22 = (brlx + 1.0) * (m / dtD) + coeff * COS_TURN;
                // the velocities at the start of the dynamics step: read where they are needed, the revised
                // relation is the rare one and every register counts here
                r1 = dU + atU + coeff * uo + (m * (brlx * uu + d.uInit[v])) / dtD;
                r2 = dV + atV + coeff * vo + (m * (brlx * vv + d.vInit[v])) / dtD;
            } else {
                l11 = m / dtE + coeff * COS_TURN;
                l22 = m / dtE + coeff * COS_TURN;
                r1 = dU + atU + coeff * uo + (m * uu) / dtE;
                r2 = dV + atV + coeff * vo + (m * vv) / dtE;
            }
            if (relation != 3) {
                const double den = l11 * l22 - l12 * l21;
                uu = (l22 * r1 - l12 * r2) / den;
                vv = (l11 * r2 - l21 * r1) / den;
                uT[tid] = uu;
                vT[tid] = vv;
            }
        }
        __syncthreads();
        if (sDead) break;  // the same answer in every thread: written before the barrier, not again after it
        if (XCHG && (it % xInterval == 0 || it == nIter)) {
            // ================= halo update of velocities and stresses with the neighbouring blocks
            xDone++;
            if (xp->tileActive[tile] == 0) continue;  // nothing of this tile travels (the same answer in every thread)
            const PersistXchg X = *xp;
            const unsigned long long tag = ((xseq0 + xDone) & 0xffffffffULL) << 32;
            // the copy: exchange xDone of this launch, in the bank of this launch's parity (the host splits loops with
            // more exchanges than a bank has copies)
            const unsigned long long par = (xl0 & 1ULL) * (SIH_TAGGED_COPIES / 2) + (unsigned long long)((xDone - 1) % (SIH_TAGGED_COPIES / 2));
            // (xCells == 0: an exchange after every iteration carries the velocities only, as the reference's does;
            // the stresses of the halo cells are then computed here from the received velocities)
            if (inRange && xCells) {
                for (int e = X.cellDestStart[c], eEnd = X.cellDestStart[c + 1]; e < eEnd; e++) {
                    // A corner the kernel does not update (no ice, or past the cell's last edge) sends what memory
                    // holds, as the exchange between launches does.
                    const unsigned long long b11 = bitsOf(corner ? o11 : X.stress11[k]);
                    const unsigned long long b22 = bitsOf(corner ? o22 : X.stress22[k]);
                    const unsigned long long b12 = bitsOf(corner ? o12 : X.stress12[k]);
                    const size_t cnt = (size_t)X.cellDestCnt[e];
                    unsigned long long *dst = (unsigned long long *)(X.cellDestAddr[e] + par * X.cellDestHalf[e]) + 2 * (size_t)g * cnt;
                    const size_t fs = 2 * (size_t)M * cnt;
                    stSystem(dst, (b11 & 0xffffffffULL) | tag);
                    stSystem(dst + cnt, (b11 >> 32) | tag);
                    stSystem(dst + fs, (b22 & 0xffffffffULL) | tag);
                    stSystem(dst + fs + cnt, (b22 >> 32) | tag);
                    stSystem(dst + 2 * fs, (b12 & 0xffffffffULL) | tag);
                    stSystem(dst + 2 * fs + cnt, (b12 >> 32) | tag);
                }
            }
            if (hasV) {
                for (int e = X.vertDestStart[v0 + tid], eEnd = X.vertDestStart[v0 + tid + 1]; e < eEnd; e++) {
                    const unsigned long long bu = bitsOf(uu), bv = bitsOf(vv);
                    unsigned long long *dst = (unsigned long long *)(X.vertDestAddr[e] + par * X.vertDestHalf[e]);
                    const size_t cnt = (size_t)X.vertDestCnt[e];
                    stSystem(dst, (bu & 0xffffffffULL) | tag);
                    stSystem(dst + cnt, (bu >> 32) | tag);
                    stSystem(dst + 2 * cnt, (bv & 0xffffffffULL) | tag);
                    stSystem(dst + 3 * cnt, (bv >> 32) | tag);
                }
            }
            const int rb = inRange && xCells ? X.recvCBase[k] : -1, rs = rb >= 0 ? X.recvCStride[k] : 0;
            const int vb = hasV ? X.recvVBase[v0 + tid] : -1, vstr = vb >= 0 ? X.recvVStride[v0 + tid] : 0;
            const unsigned long long *mine = (const unsigned long long *)(X.myBase + par * X.myHalf);
            // The upper half of every word is the tag: poll those (ten dwords a thread at most), fetch the words
            // once all of them are there.
            const unsigned int tag32 = (unsigned int)(tag >> 32);
            const unsigned int *tc = (const unsigned int *)(mine + (size_t)(rb >= 0 ? rb : 0)) + 1;
            const unsigned int *tv = (const unsigned int *)(mine + (size_t)(vb >= 0 ? vb : 0)) + 1;
            const size_t cw = 2 * (size_t)rs, cf = 4 * (size_t)M * rs, vw = 2 * (size_t)vstr;  // in dwords
            {
                long long t0 = 0;
                for (int spins = 1;; spins++) {
                    bool ready = true;
                    if (rb >= 0) {
                        const unsigned int a0 = ldSystem32(tc), a1 = ldSystem32(tc + cw);
                        const unsigned int a2 = ldSystem32(tc + cf), a3 = ldSystem32(tc + cf + cw);
                        const unsigned int a4 = ldSystem32(tc + 2 * cf), a5 = ldSystem32(tc + 2 * cf + cw);
                        ready = a0 == tag32 && a1 == tag32 && a2 == tag32 && a3 == tag32 && a4 == tag32 && a5 == tag32;
                    }
                    if (vb >= 0) {
                        const unsigned int a0 = ldSystem32(tv), a1 = ldSystem32(tv + vw);
                        const unsigned int a2 = ldSystem32(tv + 2 * vw), a3 = ldSystem32(tv + 3 * vw);
                        ready = ready && a0 == tag32 && a1 == tag32 && a2 == tag32 && a3 == tag32;
                    }
                    if (__all(ready)) break;
                    if ((spins & 7) == 0) {
                        const long long now = wall_clock64();
                        if (t0 == 0) {
                            t0 = now;
                        } else if ((unsigned long long)(now - t0) > ldAgent(&p.state[LIMIT_WORD]) || ldAgent(&p.state[PS_STATUS]) != 0ULL) {
                            if (lane == 0) {
                                __hip_atomic_fetch_max(&p.state[PS_STATUS], 3ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                sDead = 1;
                            }
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            // every word is checked once more as it is taken: one that carries another tag than this exchange's
            // (overwritten by a later exchange before it was read, which the two banks rule out) ends the launch with
            // an error instead of passing on a wrong value
            bool intact = true;
            if (rb >= 0) {
                const unsigned long long *q = mine + (size_t)rb;
                const size_t fs = 2 * (size_t)M * rs;
                const unsigned long long w0 = ldSystem(q), w1 = ldSystem(q + rs), w2 = ldSystem(q + fs), w3 = ldSystem(q + fs + rs);
                const unsigned long long w4 = ldSystem(q + 2 * fs), w5 = ldSystem(q + 2 * fs + rs);
                intact = (unsigned int)(w0 >> 32) == tag32 && (unsigned int)(w1 >> 32) == tag32 && (unsigned int)(w2 >> 32) == tag32 &&
                         (unsigned int)(w3 >> 32) == tag32 && (unsigned int)(w4 >> 32) == tag32 && (unsigned int)(w5 >> 32) == tag32;
                o11 = valueOf((w0 & 0xffffffffULL) | (w1 << 32));
                o22 = valueOf((w2 & 0xffffffffULL) | (w3 << 32));
                o12 = valueOf((w4 & 0xffffffffULL) | (w5 << 32));
            }
            if (vb >= 0) {
                const unsigned long long *q = mine + (size_t)vb;
                const unsigned long long w0 = ldSystem(q), w1 = ldSystem(q + vstr), w2 = ldSystem(q + 2 * (size_t)vstr),
                                         w3 = ldSystem(q + 3 * (size_t)vstr);
                intact = intact && (unsigned int)(w0 >> 32) == tag32 && (unsigned int)(w1 >> 32) == tag32 &&
                         (unsigned int)(w2 >> 32) == tag32 && (unsigned int)(w3 >> 32) == tag32;
                uu = valueOf((w0 & 0xffffffffULL) | (w1 << 32));
                vv = valueOf((w2 & 0xffffffffULL) | (w3 << 32));
                uT[tid] = uu;
                vT[tid] = vv;
            }
            if (!intact && !sDead) {  // (a wait that ran out has said so already)
                __hip_atomic_fetch_max(&p.state[PS_STATUS], 4ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sDead = 1;
            }
            __syncthreads();
            if (sDead) break;
        }
    }

    // ---- stresses and velocities go back to memory for whatever runs next (the last iteration of the step, an
    //      exchange)
    if (alive) {
        // with the exchange inside, what was received goes to memory too, updated by this kernel or not
        const bool gotC = XCHG && xCells && xDone > 0 && inRange && xp->recvCBase[k] >= 0;
        const bool gotV = XCHG && xDone > 0 && hasV && xp->recvVBase[v0 + tid] >= 0;
        if (corner || gotC) {
            d.stress11[k] = o11;
            d.stress22[k] = o22;
            d.stress12[k] = o12;
        }
        if (owned && ((solV && relation != 3) || gotV)) {
            d.u[v] = uu;
            d.v[v] = vv;
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// One launch per iteration, for blocks too large to stay on the chip (or not alone on their device): the same tiles
// and the same statements, but a launch is "vertices of iteration t-1, then cells of iteration t" -- the boundary
// between two launches is the hand-off of the partial sums, so nothing waits and nothing needs to be resident.  A
// loop of m iterations is m + 1 launches (cells only; m - 1 of both; vertices only) instead of 2 m, and the
// dependent chain of a launch is three memory round trips (tables; operands of the vertices; operands of the cells)
// instead of the four or five of the cell kernel followed by the vertex kernel: a tile's vertices, the places of
// their sums and whether it stores them come from tables laid out per tile, and the velocities and tan(lat) reach
// the corners through LDS.
// The velocities alternate between two arrays (a tile reads those of all its vertices while the tile that owns a
// vertex stores the new one), the partial sums between two buffers of the ring.
struct FusedArgs {
    const int *tvVert, *tvOwned, *tvQ, *cornerSlot;
    size_t nEntries;  // tiles * 64 M
    const double *uIn, *vIn;
    double *uOut, *vOut;  // may be null (cells only, nothing to copy)
    const double *sumsIn;
    double *sumsOut;
};

#ifndef SIH_FUSED_MIN_WAVES
#define SIH_FUSED_MIN_WAVES 5
#endif
template <int M, bool VERTS, bool CELLS>
__global__ void __launch_bounds__(64 * M, SIH_FUSED_MIN_WAVES) k_sub_fused(PersistDev d, FusedArgs f, int nT)
{
    __shared__ double su[M][64], sv[M][64], s11[M][64], s22[M][64], s12[M][64];
    __shared__ double uT[64 * M], vT[64 * M], tT[64 * M];
    __shared__ int arrived;  // waves of the tile that have left their part of corner 0's sums (deriveRow)
    const int lane = threadIdx.x, g = threadIdx.y, tid = g * 64 + lane;
    if (CELLS && tid == 0) arrived = 0;  // (two barriers before anyone counts)
    int tile;  // consecutive tiles on one XCD, as in the persistent kernel
    {
        const int b = blockIdx.x, q = nT / 8, r = nT % 8, x = b % 8;
        tile = x * q + (x < r ? x : r) + b / 8;
    }
    const int P = d.padC;
    int c = tile * 64 + lane;
    const bool inRange = c < d.nCells;
    if (!inRange) c = 0;
    const size_t k = (size_t)g * P + c;
    // ---- one vertex of the tile's cells
    const size_t j = (size_t)tile * (64 * M) + tid;
    const int vRaw = f.tvVert[j];
    const bool hasV = vRaw >= 0;
    const int v = hasV ? vRaw : 0;
    const bool owned = hasV && f.tvOwned[j] != 0;
    double uu = f.uIn[v], vv = f.vIn[v];
    const double tl = d.tanLat[v];
    if (VERTS) {
        const int q0 = f.tvQ[j], q1 = f.tvQ[f.nEntries + j], q2 = f.tvQ[2 * f.nEntries + j];
        const size_t plane = (size_t)M * P;
        const bool solV = hasV && (d.haloVertices ? d.vActive[v] != 0 : v < d.nVerticesSolve) && d.solveVelocity[v] == 1;
        // operands of the solve, for every lane (what a lane without a solved vertex reads is valid memory, unused)
        const double pu0 = f.sumsIn[q0], pu1 = f.sumsIn[q1], pu2 = f.sumsIn[q2];
        const double pv0 = f.sumsIn[plane + q0], pv1 = f.sumsIn[plane + q1], pv2 = f.sumsIn[plane + q2];
        const double den0 = d.varDenom[v], m = d.totalMassVertex[v], mf = d.massF[v];
        const double atU = d.airTiltU[v], atV = d.airTiltV[v];  // (air stress + tilt force, as k_sub_vertices takes them)
        const double uo = d.uOcnVert[v], vo = d.vOcnVert[v];
        const double aV = d.useOcean ? d.iceAreaVertex[v] : 0.0;
        const double dtE = d.dtE, brlx = d.brlx, dtD = d.dtDyn;
        const int relation = d.relation;
        if (solV) {
            double dU = 0.0, dV = 0.0;
            dU = dU + pu0;
            dV = dV + pv0;
            dU = dU + pu1;
            dV = dV + pv1;
            dU = dU + pu2;
            dV = dV + pv2;
            dU = dU / den0;
            dV = dV / den0;
            double coeff = 0.0;
            if (d.useOcean) {
                if (d.quadratic) {
                    double du = uo - uu, dv = vo - vv;
                    coeff = DRAGIO * RHOW * aV * sqrt(du * du + dv * dv);
                } else {
                    coeff = DRAGIO * RHOW * aV;
                }
            }
            double l11, l22, r1, r2;
            const double l12 = -mf - coeff * SIN_TURN * sign1p(mf);
            const double l21 = mf + coeff * SIN_TURN * sign1p(mf);
            if (relation == 2) {
                l11 = (brlx + 1.0) * (m / dtD) + coeff * COS_TURN;
                l22 = (brlx + 1.0) * (m / dtD) + coeff * COS_TURN;
                r1 = dU + atU + coeff * uo + (m * (brlx * uu + d.uInit[v])) / dtD;
                r2 = dV + atV + coeff * vo + (m * (brlx * vv + d.vInit[v])) / dtD;
            } else {
                l11 = m / dtE + coeff * COS_TURN;
                l22 = m / dtE + coeff * COS_TURN;
                r1 = dU + atU + coeff * uo + (m * uu) / dtE;
                r2 = dV + atV + coeff * vo + (m * vv) / dtE;
            }
            if (relation != 3) {
                const double den = l11 * l22 - l12 * l21;
                const double un = (l22 * r1 - l12 * r2) / den;
                const double vn = (l11 * r2 - l21 * r1) / den;
                uu = un;
                vv = vn;
            }
        }
    }
    // the tile that owns a vertex stores it, solved or not: both arrays hold every vertex of a tile
    if (owned && f.uOut) {
        f.uOut[v] = uu;
        f.vOut[v] = vv;
    }
    if (!CELLS) return;
    // ---- what the cell phase reads from memory is requested now, when the vertex phase has given its registers
    //      back, whether the corner turns out to be one or not (the planes are padded: every address is valid).
    //      (Requested at the top of the kernel instead, these 27 doubles took it from 90 to 122 registers.)
    int solveFlag = 0, E = 0, slot = 0;
    double P_ice = 0.0, o11 = 0.0, o22 = 0.0, o12 = 0.0;
    double gu3[3] = {0.0, 0.0, 0.0}, gv3[3] = {0.0, 0.0, 0.0};
    double bu[M], bv[M], bm[M];
    {
        solveFlag = d.solveStress[c];
        E = d.nEdgesOnCell[c];  // (with the set-up's marks above bit 7 when deriveRow is set; taken off below)
        slot = f.cornerSlot[k];
        P_ice = d.icePressure[c];
        o11 = d.stress11[k];
        o22 = d.stress22[k];
        o12 = d.stress12[k];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            gu3[q] = d.gradU3[(size_t)(q * M + g) * P + c];
            gv3[q] = d.gradV3[(size_t)(q * M + g) * P + c];
        }
#pragma unroll
        for (int s = 0; s < M; s++) {
            const size_t kb = (size_t)(g * M + s) * P + c;
            // (with deriveRow the corners 0 wait for the cell's mark below before they ask for their rows, if at all)
            bu[s] = 0.0;
            bv[s] = 0.0;
            if (!(d.deriveRow && g == 0)) {
                bu[s] = d.SU[kb];
                bv[s] = d.SV[kb];
            }
            // (a symmetric metric matrix is read from one side of its diagonal only, as in k_sub_cells_split)
            bm[s] = d.SM[d.symMetric ? (size_t)((g > s ? g : s) * M + (g > s ? s : g)) * P + c : kb];
        }
    }
    uT[tid] = uu;
    vT[tid] = vv;
    tT[tid] = tl;
    __syncthreads();
    // ---- cells: strain, stress, partial sums of the divergence (k_sub_cells_split)
    const bool rowCell = d.deriveRow && (E & 512);
    E &= 255;
    const bool corner = inRange && solveFlag == 1 && g < E;
    const bool cellDerives = inRange && solveFlag == 1 && rowCell;
    if (d.deriveRow && g == 0 && corner && !rowCell) {  // a cell without the mark: corner 0 computes from its own rows
#pragma unroll
        for (int s = 0; s < M; s++) {
            if (s < E) {
                bu[s] = d.SU[(size_t)s * P + c];
                bv[s] = d.SV[(size_t)s * P + c];
            }
        }
    }
    const double ug = uT[slot], vg = vT[slot], tg = tT[slot];
    su[g][lane] = corner ? ug : 0.0;
    sv[g][lane] = corner ? vg : 0.0;
    __syncthreads();
    const int relation = d.relation;
    double a11 = 0.0, a22 = 0.0, a12 = 0.0;
    if (corner) {
        const double dtE = d.dtE, T = d.dampT;
        double e11 = 0.0, e22 = 0.0, e12 = 0.0;
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
#pragma unroll
        for (int q = 0; q < 3; q++) {
            e11 = e11 + xu[q] * gu3[q];
            e22 = e22 + xv[q] * gv3[q];
            e12 = e12 + 0.5 * (xu[q] * gv3[q] + xv[q] * gu3[q]);
        }
        e11 = e11 - vg * tg;
        e12 = e12 + ug * tg * 0.5;
        if (relation == 3) {
            a11 = 1.0 * e11;
            a22 = 1.0 * e22;
            a12 = 1.0 * e12;
        } else {
            a11 = o11;
            a22 = o22;
            a12 = o12;
            double eD = e11 + e22, eT = e11 - e22, eS = e12 * 2.0;
            double st1 = a11 + a22, st2 = a11 - a22;
            double Delta = sqrt(eD * eD + (eT * eT + eS * eS) / ECC2);
            double pc = P_ice / fmax(Delta, PUNY);
            double den;
            if (relation == 1) {
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
    }
    s11[g][lane] = a11;
    s22[g][lane] = a22;
    s12[g][lane] = a12;
    __syncthreads();
    if (!d.deriveRow) {  // (the same for every thread of the launch)
        if (corner) {
            double cu = 0.0, cv = 0.0;
#pragma unroll
            for (int s = 0; s < M; s++) {
                if (s < E) {
                    const double x11 = s11[s][lane], x22 = s22[s][lane], x12 = s12[s][lane];
                    cu = cu - x11 * bu[s] - x12 * bv[s] - x12 * bm[s] * tg;
                    cv = cv - x22 * bv[s] - x12 * bu[s] + x11 * bm[s] * tg;
                }
            }
            f.sumsOut[k] = cu;
            f.sumsOut[(size_t)M * P + k] = cv;
        }
        return;
    }
    // k_sub_cells_split's arithmetic with derived rows: the terms with basisIntegralsU / V and the metric terms added
    // up separately; corner 0 of a marked cell takes minus the sum of the other corners' former, which the wave that
    // comes last adds up for the tile's 64 cells
    double nu = 0.0, nv = 0.0, mu = 0.0, mv = 0.0;
    if (corner) {
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
        if (cellDerives) {  // (su / sv: the velocities in them were taken before the barrier above)
            su[g][lane] = g == 0 ? mu : nu;
            sv[g][lane] = g == 0 ? mv : nv;
        }
    }
    int lastWave = 0;
    if (lane == 0)
        lastWave = __hip_atomic_fetch_add(&arrived, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == M - 1 ? 1 : 0;
    lastWave = __builtin_amdgcn_readfirstlane(lastWave);
    if (corner && !(cellDerives && g == 0)) {
        f.sumsOut[k] = nu + mu;
        f.sumsOut[(size_t)M * P + k] = nv + mv;
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
        f.sumsOut[c] = su[0][lane] - du;
        f.sumsOut[(size_t)M * P + c] = sv[0][lane] - dv;
    }
}

// where the vertices of the padded tile lists find their partial sums
__global__ void k_fused_tables(const int *__restrict__ tvVert, const int *__restrict__ gidx, int *__restrict__ tvQ, size_t n,
                               int padV, int noCell)
{
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int v = tvVert[j];
    for (int q = 0; q < 3; q++) tvQ[q * n + j] = v >= 0 ? gidx[(size_t)q * padV + v] : noCell;
}

// ------------------------------------------------------------------------------------------------ host
void sih_persist_destroy(sih_ctx *c)
{
    sih_persist_state *s = c->persist;
    if (!s) return;
    if (s->a.state && hipSetDevice(c->device) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess) {
        unsigned long long w[8];
        if (hipMemcpy(w, s->a.state, sizeof w, hipMemcpyDeviceToHost) == hipSuccess) {
            c->persistCarry[0] = w[PS_EPOCH];
            c->persistCarry[1] = w[PS_LAUNCHES];
            c->persistCarry[2] = w[PS_XSEQ];
            c->persistCarry[3] = w[PS_XLAUNCHES];
        }
    }
    for (void *q : s->allocs) (void)hipFree(q);
    for (void *q : s->xAllocs) (void)hipFree(q);
    delete s;
    c->persist = nullptr;
}

#define PS_SMALL_TILE 42  // cells of the small tile, maxEdges 6: 42 x 6 corners = 252 threads = four waves, one a SIMD
// maxEdges 7: 32 x 7 corners = 224 threads, four waves as well; with the metric rows streamed a tile takes 37.6 KB of
// LDS, four fit a CU: 32 768 cells a device, against 16 384 with tiles of 64 cells (100 KB of LDS, one a CU)
#define PS_SMALL_TILE_7 32

template <int M, int TC, bool STREAM>
static int occupancyFor(int *perCU)
{
    int n = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (k_sub_persistent<M, TC, true, STREAM>), TC * M,
                                                                persistLdsDoubles<M, TC, STREAM>() * sizeof(double));
    if (e != hipSuccess) return sih_hip_fail(e, "hipOccupancyMaxActiveBlocksPerMultiprocessor", __FILE__, __LINE__);
    *perCU = n;
    return SIH_OK;
}

template <typename T>
static int upload(sih_persist_state *s, const T **dst, const std::vector<T> &v)
{
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, sizeof(T) * std::max<size_t>(v.size(), 1)));
    s->allocs.push_back(q);
    if (!v.empty()) HIPCHK(hipMemcpy(q, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    *dst = (const T *)q;
    return SIH_OK;
}

int sih_persist_launch(sih_ctx *c, int n);

// The block cut into tiles of tc consecutive cells.  Every tile lists the vertices of its cells (a vertex shared
// with neighbouring tiles is in each of their lists); the tile that holds the lowest-numbered cell of a vertex is the
// one that stores it at the end.
struct TileTables {
    int tc = 0, nT = 0;
    std::vector<int> vStart, verts, owned, slot;  // slot: per (corner, cell), the place of its vertex in the tile's list
    bool ok = false;                              // false: a tile has more vertices than threads
};

static void buildTiles(const sih_ctx *c, int tc, const std::vector<std::vector<int>> &cellsOfVertex, TileTables &t)
{
    const int M = c->M, nC = c->nCells, nV = c->nVertices;
    t.tc = tc;
    t.nT = (nC + tc - 1) / tc;
    t.vStart.assign(t.nT + 1, 0);
    t.verts.clear();
    t.owned.clear();
    t.slot.assign((size_t)M * c->padC, 0);
    t.ok = false;
    std::vector<int> where(nV, -1);  // position of a vertex in the current tile's list
    for (int tl = 0; tl < t.nT; tl++) {
        const int first = (int)t.verts.size();
        for (int i = tl * tc; i < std::min(nC, (tl + 1) * tc); i++)
            for (int sIdx = 0; sIdx < c->hNe[i]; sIdx++) {
                const int v = c->hVoc[(size_t)sIdx * c->padC + i];
                if (v >= nV) continue;  // "no vertex": the corner reads the zero slot, which position 0 never is
                if (where[v] < first) {
                    where[v] = (int)t.verts.size();
                    t.verts.push_back(v);
                    int low = nC;
                    for (int cc : cellsOfVertex[v]) low = std::min(low, cc);
                    t.owned.push_back(low / tc == tl ? 1 : 0);
                }
                t.slot[(size_t)sIdx * c->padC + i] = where[v] - first;
            }
        t.vStart[tl + 1] = (int)t.verts.size();
        if (t.vStart[tl + 1] - t.vStart[tl] > tc * M) return;  // more vertices than threads: not this block
    }
    if (t.verts.empty()) {
        t.verts.push_back(0);
        t.owned.push_back(0);
    }
    t.ok = true;
}

// Builds the tile tables.  Returns SIH_OK with c->persist->usable = false when the block does not qualify
// (then the fused or the two-kernel form runs); an error only for HIP failures.
int sih_persist_setup(sih_ctx *c)
{
    sih_persist_destroy(c);
    sih_persist_state *s = new sih_persist_state();
    c->persist = s;
    const int M = c->M, nC = c->nCells, nV = c->nVertices;
    if (!c->sparseGrad || !c->splitCells || M > 8 || c->D != 3 || c->cfg.strainScheme != 0 || c->cfg.stressDivergenceScheme != 0 ||
        c->cfg.averageVariationalStrain || c->cfg.useSpecialBoundariesVelocity || c->cfg.useSpecialBoundariesVelocityMasks) {
        s->reason = 1;
        return SIH_OK;
    }
    HIPCHK(hipSetDevice(c->device));
    int rc = SIH_OK;
    if (nC < 1) {
        s->reason = 2;
        return SIH_OK;
    }
    std::vector<std::vector<int>> cellsOfVertex(nV);
    for (int i = 0; i < nC; i++)
        for (int sIdx = 0; sIdx < c->hNe[i]; sIdx++) {
            const int v = c->hVoc[(size_t)sIdx * c->padC + i];
            if (v < nV) cellsOfVertex[v].push_back(i);
        }
    TileTables t64;
    buildTiles(c, 64, cellsOfVertex, t64);
    if (!t64.ok) {
        s->reason = 3;
        return SIH_OK;
    }
    // the buffers of the partial sums, zero where no corner ever writes (the slot that stands for "no cell")
    if ((rc = sih_add_field(c, "_partRing", LOC_CELL, 2 * PS_RING * M, false))) return rc;
    PersistArgs &a = s->a;
    a.part = (unsigned long long *)sih_find_field(c, "_partRing")->d;
    {
        // what k_sub_fused reads: the lists of the 64-cell tiles with 64 M entries a tile (no look-up of where a
        // tile's list begins), the places of the vertices' partial sums beside them, the corners' places in the
        // lists, and the second copy of the velocities
        const int nT = t64.nT;
        const size_t nE = (size_t)nT * 64 * M;
        std::vector<int> pv(nE, -1), po(nE, 0);
        for (int t = 0; t < nT; t++)
            for (int jj = t64.vStart[t]; jj < t64.vStart[t + 1]; jj++) {
                pv[(size_t)t * 64 * M + (jj - t64.vStart[t])] = t64.verts[jj];
                po[(size_t)t * 64 * M + (jj - t64.vStart[t])] = t64.owned[jj];
            }
        if ((rc = upload(s, &s->tvVert, pv))) return rc;
        if ((rc = upload(s, &s->tvOwned, po))) return rc;
        if ((rc = upload(s, &s->fusedSlot, t64.slot))) return rc;
        s->fusedTiles = nT;
        void *tq = nullptr;
        HIPCHK(hipMalloc(&tq, sizeof(int) * 3 * nE));
        s->allocs.push_back(tq);
        s->tvQ = (int *)tq;
        hipLaunchKernelGGL(k_fused_tables, dim3((unsigned)((nE + 255) / 256)), dim3(256), 0, c->stream, s->tvVert, c->dev.gidx,
                           s->tvQ, nE, c->padV, nC);
        HIPCHK(hipGetLastError());
        if ((rc = sih_add_field(c, "_uAlt", LOC_VERTEX, 1, false))) return rc;
        if ((rc = sih_add_field(c, "_vAlt", LOC_VERTEX, 1, false))) return rc;
        s->uAlt = (double *)sih_find_field(c, "_uAlt")->d;
        s->vAlt = (double *)sih_find_field(c, "_vAlt")->d;
        HIPCHK(hipStreamSynchronize(c->stream));
        s->fusedReady = true;
    }

    // ---- the persistent kernel: tiles of 64 cells with all integral rows in LDS if the block fits, else (maxEdges
    //      6) tiles of 42 cells with the metric rows streamed, which hold a third more cells
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, c->device));
    int perCU = 0;
    switch (M) {
        case 3: rc = occupancyFor<3, 64, false>(&perCU); break;
        case 4: rc = occupancyFor<4, 64, false>(&perCU); break;
        case 5: rc = occupancyFor<5, 64, false>(&perCU); break;
        case 6: rc = occupancyFor<6, 64, false>(&perCU); break;
        case 7: rc = occupancyFor<7, 64, false>(&perCU); break;
        default: rc = occupancyFor<8, 64, false>(&perCU); break;
    }
    if (rc) return rc;
    // the runtime's answer can be one too high for kernels that hold many scalar registers; the probe launch below
    // settles whether this grid is resident at once
    s->perCU = perCU;
    s->capacity = perCU * prop.multiProcessorCount;
    if (const char *e = getenv("SIH_PERSIST_CAPACITY")) s->capacity = atoi(e);
    TileTables tSmall;
    const TileTables *use = &t64;
    s->tileCells = 64;
    const char *smallEnv = getenv("SIH_PERSIST_SMALL_TILES");  // 0: tiles of 64 cells only
    const bool smallTiles = !(smallEnv && atoi(smallEnv) == 0);
    if (t64.nT > s->capacity) {
        bool fits = false;
        if ((M == 6 || M == 7) && smallTiles) {
            int perCUs = 0;
            const int tcSmall = M == 6 ? PS_SMALL_TILE : PS_SMALL_TILE_7;
            if ((rc = M == 6 ? occupancyFor<6, PS_SMALL_TILE, true>(&perCUs) : occupancyFor<7, PS_SMALL_TILE_7, true>(&perCUs)))
                return rc;
            int cap = perCUs * prop.multiProcessorCount;
            if (const char *e = getenv("SIH_PERSIST_CAPACITY_SMALL")) cap = atoi(e);
            buildTiles(c, tcSmall, cellsOfVertex, tSmall);
            if (tSmall.ok && tSmall.nT <= cap) {
                fits = true;
                use = &tSmall;
                s->tileCells = tcSmall;
                s->perCU = perCUs;
                s->capacity = cap;
            }
        }
        if (!fits) {  // too many tiles to be resident together: no persistent kernel for this block
            a.nTiles = t64.nT;
            s->reason = 2;
            return SIH_OK;
        }
    }
    s->hVStart = use->vStart;
    s->hVerts = use->verts;
    s->hOwned = use->owned;
    a.nTiles = use->nT;
    if ((rc = upload(s, &a.tileVertStart, use->vStart))) return rc;
    if ((rc = upload(s, &a.tileVerts, use->verts))) return rc;
    if ((rc = upload(s, &a.tileVertOwned, use->owned))) return rc;
    if (use == &t64)
        a.cornerSlot = s->fusedSlot;
    else if ((rc = upload(s, &a.cornerSlot, use->slot)))
        return rc;
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, sizeof(unsigned long long) * PS_WORDS));
    s->allocs.push_back(q);
    HIPCHK(hipMemset(q, 0, sizeof(unsigned long long) * PS_WORDS));
    a.state = (unsigned long long *)q;
    {
        // go on counting where the tables this call replaces had come to (see persistCarry)
        unsigned long long w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        w[PS_EPOCH] = c->persistCarry[0];
        // (not PS_LAUNCHES: the census counters, which start at zero again with these tables, are counted against it)
        w[PS_XSEQ] = c->persistCarry[2];
        w[PS_XLAUNCHES] = c->persistCarry[3];
        HIPCHK(hipMemcpy(a.state, w, sizeof w, hipMemcpyHostToDevice));
    }
    unsigned long long ticks = 2ULL * 100000000ULL;  // 2 s of the 100 MHz wall clock
    if (const char *e = getenv("SIH_PERSIST_TIMEOUT_MS")) ticks = (unsigned long long)(atof(e) * 100000.0);
    HIPCHK(hipMemcpy(a.state + PS_TICKS, &ticks, sizeof ticks, hipMemcpyHostToDevice));
    // waits for another block's halo values: that block may start its step later (10 s, or what was asked for)
    unsigned long long xticks = 10ULL * 100000000ULL;
    if (const char *e = getenv("SIH_HALO_TIMEOUT_MS")) xticks = (unsigned long long)(atof(e) * 100000.0);
    if (getenv("SIH_PERSIST_TIMEOUT_MS") || xticks < ticks) xticks = ticks;
    HIPCHK(hipMemcpy(a.state + PS_XTICKS, &xticks, sizeof xticks, hipMemcpyHostToDevice));
    // probe: a launch of zero iterations runs the census only.  If its workgroups are not all resident together it
    // says so within the time limit and the other forms stay in use.
    s->usable = true;
    HIPCHK(hipStreamSynchronize(c->stream));
    if ((rc = sih_persist_launch(c, 0))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    unsigned long long st = 0;
    HIPCHK(hipMemcpy(&st, a.state + PS_STATUS, sizeof st, hipMemcpyDeviceToHost));
    if (st != 0) {
        s->usable = false;
        s->reason = 5;
        std::vector<unsigned long long> w(PS_WORDS);
        HIPCHK(hipMemcpy(w.data(), a.state, sizeof(unsigned long long) * PS_WORDS, hipMemcpyDeviceToHost));
        for (int j = 0; j < PS_CENSUS_N; j++) s->arrived += (int)w[PS_CENSUS + j * PS_CENSUS_STRIDE];
    }
    return SIH_OK;
}

double sih_persist_info(sih_ctx *c, int what)
{
    if (!c->persist) return 0.0;
    const sih_persist_state *s = c->persist;
    return what == 0 ? s->perCU : what == 1 ? s->reason : what == 2 ? s->a.nTiles : what == 3 ? s->arrived : what == 4 ? s->launches
           : what == 5 ? s->xLaunches : what == 6 ? s->fusedLaunches : s->tileCells;
}

// the kernel reads the compact gradients: a put of a basis array switches it off until sih_evp_static_changed()
bool sih_persist_usable(sih_ctx *c) { return c->persist && c->persist->usable && c->sparseGrad; }

template <int M, int TC, bool STREAM>
static void launchPersistVariant(sih_ctx *c, int n, int interval)
{
    const PersistArgs &a = c->persist->a;
    const size_t lds = persistLdsDoubles<M, TC, STREAM>() * sizeof(double);
    if (interval > 0)
        hipLaunchKernelGGL((k_sub_persistent<M, TC, true, STREAM>), dim3((unsigned)a.nTiles), dim3(TC, M), lds, c->stream,
                           makePersistDev(c->dev), a, n, (const PersistXchg *)c->persist->d_x, interval, interval > 1 ? 1 : 0);
    else
        hipLaunchKernelGGL((k_sub_persistent<M, TC, false, STREAM>), dim3((unsigned)a.nTiles), dim3(TC, M), lds, c->stream,
                           makePersistDev(c->dev), a, n, (const PersistXchg *)nullptr, 1, 0);
}

template <int M>
static void launchPersist(sih_ctx *c, int n, int interval)
{
    if constexpr (M == 6) {
        if (c->persist->tileCells == PS_SMALL_TILE) {
            launchPersistVariant<6, PS_SMALL_TILE, true>(c, n, interval);
            return;
        }
    }
    if constexpr (M == 7) {
        if (c->persist->tileCells == PS_SMALL_TILE_7) {
            launchPersistVariant<7, PS_SMALL_TILE_7, true>(c, n, interval);
            return;
        }
    }
    launchPersistVariant<M, 64, false>(c, n, interval);
}

static int launchAny(sih_ctx *c, int n, int interval)
{
    switch (c->M) {
        case 3: launchPersist<3>(c, n, interval); break;
        case 4: launchPersist<4>(c, n, interval); break;
        case 5: launchPersist<5>(c, n, interval); break;
        case 6: launchPersist<6>(c, n, interval); break;
        case 7: launchPersist<7>(c, n, interval); break;
        default: launchPersist<8>(c, n, interval); break;
    }
    HIPCHK(hipGetLastError());
    c->persist->unchecked = true;
    c->persist->launches++;
    if (interval > 0) c->persist->xLaunches++;
    return SIH_OK;
}

// n iterations (none the last of the dynamics step) as one launch; asynchronous
int sih_persist_launch(sih_ctx *c, int n) { return launchAny(c, n, 0); }

// ---- the one-launch-per-iteration form
bool sih_fused_usable(sih_ctx *c) { return c->persist && c->persist->fusedReady && c->sparseGrad; }

template <int M>
static void launchFused(sih_ctx *c, int m)
{
    sih_persist_state *s = c->persist;
    const int nT = s->fusedTiles;
    const PersistDev pd = makePersistDev(c->dev);
    const size_t ringBuf = (size_t)2 * M * c->padC;  // doubles in one buffer of the ring (U planes, then V planes)
    double *sums[2] = {(double *)s->a.part, (double *)s->a.part + ringBuf};
    double *ub[2] = {c->dev.u, s->uAlt}, *vb[2] = {c->dev.v, s->vAlt};
    FusedArgs f{};
    f.tvVert = s->tvVert;
    f.tvOwned = s->tvOwned;
    f.tvQ = s->tvQ;
    f.cornerSlot = s->fusedSlot;
    f.nEntries = (size_t)nT * 64 * M;
    const dim3 grid((unsigned)nT), block(64, M);
    // Launch 1: cells of iteration 1.  Launch i = 2 .. m: vertices of iteration i-1, cells of iteration i.  Launch
    // m + 1: vertices of iteration m, which must land in the block's own arrays: the launches write the two copies
    // of the velocities in turn, and with an odd m the first launch (which reads the velocities of all its vertices
    // anyway) copies them over, so that the turn comes out right.
    const bool copyFirst = (m % 2) == 1;
    int in = 0;
    f.uIn = ub[0];
    f.vIn = vb[0];
    f.uOut = copyFirst ? ub[1] : nullptr;
    f.vOut = copyFirst ? vb[1] : nullptr;
    f.sumsIn = sums[0];
    f.sumsOut = sums[1];
    hipLaunchKernelGGL((k_sub_fused<M, false, true>), grid, block, 0, c->stream, pd, f, nT);
    if (copyFirst) in = 1;
    for (int i = 2; i <= m + 1; i++) {
        f.uIn = ub[in];
        f.vIn = vb[in];
        f.uOut = ub[in ^ 1];
        f.vOut = vb[in ^ 1];
        f.sumsIn = sums[(i - 1) % 2];
        f.sumsOut = sums[i % 2];
        if (i <= m)
            hipLaunchKernelGGL((k_sub_fused<M, true, true>), grid, block, 0, c->stream, pd, f, nT);
        else
            hipLaunchKernelGGL((k_sub_fused<M, true, false>), grid, block, 0, c->stream, pd, f, nT);
        in ^= 1;
    }
}

// m iterations (none the last of the dynamics step) as m + 1 launches on the context's stream; asynchronous, and
// plain launches: it may be captured in a graph
int sih_fused_launch(sih_ctx *c, int m)
{
    if (!sih_fused_usable(c) || m < 1) {
        sih_set_error("fused subcycle kernel: not available for this block");
        return SIH_ERR_UNSUPPORTED;
    }
    switch (c->M) {
        case 3: launchFused<3>(c, m); break;
        case 4: launchFused<4>(c, m); break;
        case 5: launchFused<5>(c, m); break;
        case 6: launchFused<6>(c, m); break;
        case 7: launchFused<7>(c, m); break;
        default: launchFused<8>(c, m); break;
    }
    HIPCHK(hipGetLastError());
    c->persist->fusedLaunches++;
    return SIH_OK;
}

template <typename T>
static int uploadX(sih_persist_state *s, const T **dst, const std::vector<T> &v)
{
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, sizeof(T) * std::max<size_t>(v.size(), 1)));
    s->xAllocs.push_back(q);
    if (!v.empty()) HIPCHK(hipMemcpy(q, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    *dst = (const T *)q;
    return SIH_OK;
}

// Tables of the in-kernel exchange from the lists given to sih_halo_set_lists and the connected transport.  Built
// on first use and again when the transport has changed.
static int exchangeSetup(sih_ctx *c)
{
    sih_persist_state *s = c->persist;
    sih_comm_tagged t;
    const bool have = sih_comm_tagged_info(c, &t);
    if (s->xGeneration == t.generation && t.generation != 0) return SIH_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (void *q : s->xAllocs) (void)hipFree(q);
    s->xAllocs.clear();
    s->d_x = nullptr;
    s->xUsable = false;
    s->xGeneration = t.generation;
    if (!have || c->dev.relation == 3) return SIH_OK;
    const int M = c->M, nT = s->a.nTiles, nC = c->nCells, nV = c->nVertices, P = c->padC, nR = t.nRanks;
    // where a vertex sits: the tile that stores it at the end, and its place in that tile's list
    std::vector<int> ownerTile(nV, -1), ownerPos(nV, -1);
    for (int tl = 0; tl < nT; tl++)
        for (int j = s->hVStart[tl]; j < s->hVStart[tl + 1]; j++)
            if (s->hOwned[j]) {
                ownerTile[s->hVerts[j]] = tl;
                ownerPos[s->hVerts[j]] = j - s->hVStart[tl];
            }
    // The two banks of the tagged words are safe to reuse because whoever writes to a neighbour also waits for words
    // of that neighbour in every exchange (sih_internal.h): a block that would send to one it receives no vertex
    // from keeps the exchange between launches.
    for (int p = 0; p < nR; p++)
        if (c->hSendCounts[0][p] + c->hSendCounts[1][p] > 0 && c->hRecvCounts[1][p] == 0) return SIH_OK;
    // ---- what every cell and every stored vertex sends
    std::vector<int> tileActive(nT, 0);
    std::vector<std::vector<int>> cellDest(nC), vertDest(s->hVerts.size());  // indices into the flat entry arrays
    std::vector<unsigned long long> cAddr, cHalf, vAddr, vHalf;
    std::vector<int> cCnt, vCnt;
    {
        size_t at = 0;
        for (int p = 0; p < nR; p++) {
            const int cnt = c->hSendCounts[0][p];
            for (int r = 0; r < cnt; r++, at++) {
                const int cell = c->hSend[0][at];
                if (cell < 0 || cell >= nC) return SIH_OK;
                cellDest[cell].push_back((int)cAddr.size());
                cAddr.push_back((unsigned long long)(uintptr_t)t.peerBase[p] + 8ULL * (t.peerWordOff[p] + (size_t)t.sendPos[0][at]));
                cHalf.push_back(t.peerHalfBytes[p]);
                cCnt.push_back(cnt);
                tileActive[cell / s->tileCells] = 1;
            }
        }
        at = 0;
        for (int p = 0; p < nR; p++) {
            const int cnt = c->hSendCounts[0][p], cntV = c->hSendCounts[1][p];
            for (int r = 0; r < cntV; r++, at++) {
                const int v = c->hSend[1][at];
                if (v < 0 || v >= nV || ownerTile[v] < 0) return SIH_OK;  // a vertex no tile holds: no in-kernel exchange
                vertDest[s->hVStart[ownerTile[v]] + ownerPos[v]].push_back((int)vAddr.size());
                vAddr.push_back((unsigned long long)(uintptr_t)t.peerBase[p] +
                                8ULL * (t.peerWordOff[p] + (size_t)6 * M * cnt + (size_t)t.sendPos[1][at]));
                vHalf.push_back(t.peerHalfBytes[p]);
                vCnt.push_back(cntV);
                tileActive[ownerTile[v]] = 1;
            }
        }
    }
    // entries in the order of their senders
    std::vector<int> cStart(nC + 1, 0), vStartD(s->hVerts.size() + 1, 0);
    std::vector<unsigned long long> cAddrS, cHalfS, vAddrS, vHalfS;
    std::vector<int> cCntS, vCntS;
    for (int i = 0; i < nC; i++) {
        for (int e : cellDest[i]) {
            cAddrS.push_back(cAddr[e]);
            cHalfS.push_back(cHalf[e]);
            cCntS.push_back(cCnt[e]);
        }
        cStart[i + 1] = (int)cAddrS.size();
    }
    for (size_t j = 0; j < s->hVerts.size(); j++) {
        for (int e : vertDest[j]) {
            vAddrS.push_back(vAddr[e]);
            vHalfS.push_back(vHalf[e]);
            vCntS.push_back(vCnt[e]);
        }
        vStartD[j + 1] = (int)vAddrS.size();
    }
    // ---- what every corner and every listed vertex receives
    std::vector<int> rcBase((size_t)M * P, -1), rcStride((size_t)M * P, 0);
    std::vector<int> rvBaseOfVertex(nV, -1), rvStrideOfVertex(nV, 0);
    {
        size_t at = 0;
        for (int p = 0; p < nR; p++) {
            const int cnt = c->hRecvCounts[0][p];
            for (int r = 0; r < cnt; r++, at++) {
                const int cell = c->hRecv[0][at];
                if (cell < 0 || cell >= nC) return SIH_OK;
                for (int g = 0; g < M; g++) {
                    rcBase[(size_t)g * P + cell] = (int)(t.myWordOff[p] + (size_t)2 * g * cnt + (size_t)t.recvPos[0][at]);
                    rcStride[(size_t)g * P + cell] = cnt;
                }
                tileActive[cell / s->tileCells] = 1;
            }
        }
        at = 0;
        for (int p = 0; p < nR; p++) {
            const int cnt = c->hRecvCounts[0][p], cntV = c->hRecvCounts[1][p];
            for (int r = 0; r < cntV; r++, at++) {
                const int v = c->hRecv[1][at];
                if (v < 0 || v >= nV || ownerTile[v] < 0) return SIH_OK;
                rvBaseOfVertex[v] = (int)(t.myWordOff[p] + (size_t)6 * M * cnt + (size_t)t.recvPos[1][at]);
                rvStrideOfVertex[v] = cntV;
            }
        }
    }
    std::vector<int> rvBase(s->hVerts.size(), -1), rvStride(s->hVerts.size(), 0);
    for (int tl = 0; tl < nT; tl++)
        for (int j = s->hVStart[tl]; j < s->hVStart[tl + 1]; j++) {
            rvBase[j] = rvBaseOfVertex[s->hVerts[j]];
            rvStride[j] = rvStrideOfVertex[s->hVerts[j]];
            if (rvBase[j] >= 0) tileActive[tl] = 1;
        }
    PersistXchg x{};
    int rc;
    if ((rc = uploadX(s, &x.tileActive, tileActive))) return rc;
    if ((rc = uploadX(s, &x.cellDestStart, cStart))) return rc;
    if ((rc = uploadX(s, &x.cellDestAddr, cAddrS))) return rc;
    if ((rc = uploadX(s, &x.cellDestHalf, cHalfS))) return rc;
    if ((rc = uploadX(s, &x.cellDestCnt, cCntS))) return rc;
    if ((rc = uploadX(s, &x.vertDestStart, vStartD))) return rc;
    if ((rc = uploadX(s, &x.vertDestAddr, vAddrS))) return rc;
    if ((rc = uploadX(s, &x.vertDestHalf, vHalfS))) return rc;
    if ((rc = uploadX(s, &x.vertDestCnt, vCntS))) return rc;
    if ((rc = uploadX(s, &x.recvCBase, rcBase))) return rc;
    if ((rc = uploadX(s, &x.recvCStride, rcStride))) return rc;
    if ((rc = uploadX(s, &x.recvVBase, rvBase))) return rc;
    if ((rc = uploadX(s, &x.recvVStride, rvStride))) return rc;
    x.myBase = t.myBase;
    x.myHalf = t.myHalfBytes;
    x.stress11 = c->dev.stress11;
    x.stress22 = c->dev.stress22;
    x.stress12 = c->dev.stress12;
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, sizeof x));
    s->xAllocs.push_back(q);
    HIPCHK(hipMemcpy(q, &x, sizeof x, hipMemcpyHostToDevice));
    s->d_x = (PersistXchg *)q;
    s->xUsable = true;
    return SIH_OK;
}

bool sih_persist_exchange_usable(sih_ctx *c)
{
    if (!sih_persist_usable(c) || !sih_comm_ready(c)) return false;
    if (exchangeSetup(c) != SIH_OK) return false;
    return c->persist->xUsable;
}

int sih_persist_launch_exchange(sih_ctx *c, int n, int interval)
{
    if (!sih_persist_exchange_usable(c) || n < 1 || interval < 1) {
        sih_set_error("persistent subcycle kernel: the exchange inside the kernel is not available for this block");
        return SIH_ERR_UNSUPPORTED;
    }
    // a launch holds at most as many exchanges as a bank of the tagged words has copies (sih_internal.h): longer loops
    // are split, each part but the last a whole number of intervals, so that the exchanges fall where they would
    const int most = (SIH_TAGGED_COPIES / 2) * interval;
    for (int done = 0; done < n;) {
        const int m = std::min(most, n - done);
        int rc = launchAny(c, m, interval);
        if (rc) return rc;
        done += m;
    }
    return SIH_OK;
}

// waits for the stream; SIH_ERR_STATE if a launch gave up (workgroups not resident together, or a wait ran out)
int sih_persist_status(sih_ctx *c)
{
    if (!sih_persist_usable(c) || !c->persist->unchecked) return SIH_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->persist->unchecked = false;
    unsigned long long st = 0;
    HIPCHK(hipMemcpy(&st, c->persist->a.state + PS_STATUS, sizeof st, hipMemcpyDeviceToHost));
    if (st) {
        sih_set_error(st == 1   ? "persistent subcycle kernel: its workgroups were not resident together (is the GPU shared with other "
                                  "work? SIH_PERSISTENT=0 or sih_evp_set_persistent(ctx, 0) keeps the two-kernel form)"
                      : st == 2 ? "persistent subcycle kernel: a wait for a neighbouring tile ran out of time"
                      : st == 3 ? "persistent subcycle kernel: a neighbouring block's halo values did not arrive in time"
                                : "persistent subcycle kernel: halo values were overwritten before they were read (do all blocks make "
                                  "the same sequence of calls?)");
        c->persist->usable = false;
        return SIH_ERR_STATE;
    }
    return SIH_OK;
}

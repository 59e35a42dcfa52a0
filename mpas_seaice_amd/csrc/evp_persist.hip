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
// pass through memory, and a tile waits for exactly the tiles whose partial sums it reads (per-tile iteration
// counters), never for the whole grid -- one hand-off per iteration.
//
// The arithmetic is that of k_sub_cells_split / k_sub_vertices, statement for statement, so the results are
// the same bits.  Iterations that store the end-of-step diagnostics (the last one of a dynamics step) stay with
// the two-kernel form.
//
// Hand-off rules (per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed):
//   - the partial sums are written with agent-scope atomic stores (write-through), every storing wave drains
//     them (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, then one lane publishes the tile's iteration
//     counter with the same kind of store;
//   - a reader polls the counters of the tiles it depends on with agent-scope atomic loads (one lane per
//     counter), the workgroup meets at a barrier, then reads the sums with agent-scope atomic loads (L1
//     bypassed), optionally behind an agent-scope acquire fence (SIH_PERSIST_ACQUIRE=1);
//   - the sums live in two buffers used in turn: a tile overwrites what it wrote two iterations ago, which every
//     reader has finished with, because the tile has since seen that reader's counter move on;
//   - every wait is bounded by the wall clock; a wait that runs out sets a status word, every workgroup leaves
//     the loop, and the host reports it.  All workgroups must be resident together: the launch is sized by the
//     occupancy the runtime reports, the kernel begins with a census (every workgroup checks in and waits for
//     the others), and a probe launch at set-up settles whether this grid fits before it is ever used.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <set>

#include "sih_internal.h"

#define ECC2 4.0
#define PUNY 1.0e-11
#define RHOW 1026.0
#define DRAGIO 0.00536
#define SIN_TURN 0.0
#define COS_TURN 1.0

enum { PS_EPOCH = 0, PS_LAUNCHES = 1, PS_ARRIVE = 2, PS_FINISH = 3, PS_STATUS = 4, PS_TICKS = 5, PS_WORDS = 16 };

struct PersistArgs {
    int nTiles;
    const int *tileVertStart, *tileVerts;  // vertices of a tile's cells: tileVerts[tileVertStart[t] .. tileVertStart[t+1])
    const int *tileVertOwned;              // same indexing: 1 where this tile is the one that stores the vertex at the end
    const int *cornerSlot;                 // per (corner, cell) [g * padC + c]: position of its vertex in the tile's list
    const int *depStart, *dep;             // tiles whose partial sums a tile's vertices read
    unsigned int *cellDone;                // per tile: last iteration (absolute number) whose sums are published
    double *partU2, *partV2;               // the second buffer of the partial sums (odd iterations)
    unsigned long long *state;             // PS_* words
    int acquire;                           // agent-scope acquire fence after every wait
};

// what the kernel reads of SihDev, and nothing else: the fewer scalar registers the arguments take, the more
// vector registers are left for the state a thread keeps
struct PersistDev {
    int nCells, nVerticesSolve, padC, padV, relation, useOcean, quadratic, haloVertices;
    double dtE, dampT, brlx, dtDyn;
    const int *nEdgesOnCell, *voc, *gidx, *vActive, *solveStress, *solveVelocity;
    const double *tanLat, *varDenom, *gradU3, *gradV3, *SU, *SV, *SM, *icePressure;
    const double *totalMassVertex, *massF, *airVertU, *airVertV, *tiltForceU, *tiltForceV, *ocnStressU, *ocnStressV;
    const double *uOcnVert, *vOcnVert, *iceAreaVertex, *uInit, *vInit;
    double *u, *v, *stress11, *stress22, *stress12, *partU, *partV;
};

static PersistDev makePersistDev(const SihDev &d)
{
    PersistDev q;
    q.nCells = d.nCells; q.nVerticesSolve = d.nVerticesSolve; q.padC = d.padC; q.padV = d.padV;
    q.relation = d.relation; q.useOcean = d.useOcean; q.quadratic = d.quadratic; q.haloVertices = d.haloVertices;
    q.dtE = d.dtE; q.dampT = d.dampT; q.brlx = d.brlx; q.dtDyn = d.dtDyn;
    q.nEdgesOnCell = d.nEdgesOnCell; q.voc = d.voc; q.gidx = d.gidx; q.vActive = d.vActive;
    q.solveStress = d.solveStress; q.solveVelocity = d.solveVelocity;
    q.tanLat = d.tanLat; q.varDenom = d.varDenom; q.gradU3 = d.gradU3; q.gradV3 = d.gradV3;
    q.SU = d.SU; q.SV = d.SV; q.SM = d.SM; q.icePressure = d.icePressure;
    q.totalMassVertex = d.totalMassVertex; q.massF = d.massF; q.airVertU = d.airVertU; q.airVertV = d.airVertV;
    q.tiltForceU = d.tiltForceU; q.tiltForceV = d.tiltForceV; q.ocnStressU = d.ocnStressU; q.ocnStressV = d.ocnStressV;
    q.uOcnVert = d.uOcnVert; q.vOcnVert = d.vOcnVert; q.iceAreaVertex = d.iceAreaVertex; q.uInit = d.uInit; q.vInit = d.vInit;
    q.u = d.u; q.v = d.v; q.stress11 = d.stress11; q.stress22 = d.stress22; q.stress12 = d.stress12;
    q.partU = d.partU; q.partV = d.partV;
    return q;
}

struct sih_persist_state {
    PersistArgs a{};
    int capacity = 0;  // workgroups the device holds at once
    int arrived = 0;
    int perCU = 0, reason = 0;  // why not usable: 1 option, 2 too many tiles, 3 vertices per tile, 4 dependencies, 5 probe
    bool usable = false;
    std::vector<void *> allocs;
};

__device__ inline double ldAgent(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}
__device__ inline void stAgent(double *p, double x)
{
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double sign1p(double x) { return copysign(1.0, x); }

// wave 0 polls cnt[dep[j]] (lane j) until every one has reached `target`; the verdict reaches the whole
// workgroup through LDS.  Returns false when the time limit ran out or another workgroup has given up.
__device__ inline bool waitTiles(const unsigned int *cnt, const int *dep, int nDep, unsigned int target,
                                 unsigned long long *state, unsigned long long limit, int *okShared, int tid,
                                 int acquire)
{
    if (tid < 64) {
        bool good = true;
        const long long t0 = wall_clock64();
        for (int base = 0; base < nDep && good; base += 64) {
            const int j = base + tid;
            const unsigned int *w = cnt + (j < nDep ? dep[j] : dep[base]);
            for (;;) {
                // signed distance: the counters are absolute iteration numbers that may wrap
                const int behind = (int)(target - __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (__all(behind <= 0)) break;
                if ((unsigned long long)(wall_clock64() - t0) > limit ||
                    __hip_atomic_load(&state[PS_STATUS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ULL) {
                    good = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (!good && tid == 0)
            __hip_atomic_fetch_max(&state[PS_STATUS], 2ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) *okShared = good ? 1 : 0;
        if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    const bool ok = *okShared != 0;
    __syncthreads();  // okShared may be rewritten by the next wait
    return ok;
}

// At most 128 vector registers a lane (launch bound of four waves a SIMD), so that two workgroups of 6 waves are
// resident on a CU wherever the hardware starts placing their waves; at 3 waves a SIMD the second workgroup was
// not admitted.
// LDS of one workgroup in doubles: staging of corner velocities and stresses (5 M 64), the tile's vertex velocities
// (2 * 64 M), the integral rows (3 M * 64 M).  Requested at launch, not declared statically: with the size in the
// kernel the compiler takes the two workgroups a CU's LDS holds for the occupancy and lets the registers grow past
// 128 a lane, where the hardware no longer admits the second workgroup.
template <int M>
constexpr int persistLdsDoubles() { return 5 * M * 64 + 2 * 64 * M + 3 * M * 64 * M; }

template <int M>
__global__ void __launch_bounds__(64 * M, 4) k_sub_persistent(PersistDev d, PersistArgs p, int nIter)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double(*su)[64] = (double(*)[64])(smem);
    double(*sv)[64] = (double(*)[64])(smem + M * 64);
    double(*s11)[64] = (double(*)[64])(smem + 2 * M * 64);
    double(*s22)[64] = (double(*)[64])(smem + 3 * M * 64);
    double(*s12)[64] = (double(*)[64])(smem + 4 * M * 64);
    double *uT = smem + 5 * M * 64, *vT = uT + 64 * M;       // velocities of the tile's vertices
    double(*sInt)[64 * M] = (double(*)[64 * M])(vT + 64 * M);  // this thread's rows of basisIntegralsU / V / Metric
    __shared__ int okShared;
    const int lane = threadIdx.x, g = threadIdx.y, tid = g * 64 + lane;
    // consecutive tiles on one XCD (workgroups are dealt round-robin over the eight XCDs): neighbouring tiles then
    // mostly hand their sums over through one L2
    const int nT = p.nTiles;
    int tile;
    {
        const int b = blockIdx.x, q = nT / 8, r = nT % 8, x = b % 8;
        tile = x * q + (x < r ? x : r) + b / 8;
    }
    const unsigned long long limit = __hip_atomic_load(&p.state[PS_TICKS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- census: every workgroup of the launch is resident before anything is computed
    const unsigned long long launches = __hip_atomic_load(&p.state[PS_LAUNCHES], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long epoch = __hip_atomic_load(&p.state[PS_EPOCH], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
        int good = __hip_atomic_load(&p.state[PS_STATUS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ULL;
        __hip_atomic_fetch_add(&p.state[PS_ARRIVE], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long want = (launches + 1ULL) * (unsigned long long)nT;
        const long long t0 = wall_clock64();
        while (good && __hip_atomic_load(&p.state[PS_ARRIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            if ((unsigned long long)(wall_clock64() - t0) > limit ||
                __hip_atomic_load(&p.state[PS_STATUS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ULL) {
                __hip_atomic_fetch_max(&p.state[PS_STATUS], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        okShared = good;
    }
    __syncthreads();
    bool alive = okShared != 0;
    __syncthreads();

    // ---- what this thread keeps for all iterations: its cell corner ...
    const int P = d.padC, PV = d.padV;
    int c = tile * 64 + lane;
    const bool inRange = c < d.nCells;
    if (!inRange) c = 0;
    const size_t k = (size_t)g * P + c;
    const bool solve = inRange && d.solveStress[c] == 1;
    const int E = inRange ? d.nEdgesOnCell[c] : 0;
    const bool corner = solve && g < E;
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
            z = d.SM[kb];
        }
        sInt[s][tid] = x;
        sInt[M + s][tid] = y;
        sInt[2 * M + s][tid] = z;
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
    double den0 = 1.0, m = 0.0, mf = 0.0, airU = 0.0, airV = 0.0, tiltU = 0.0, tiltV = 0.0, osU = 0.0, osV = 0.0;
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
        airU = d.airVertU[v];
        airV = d.airVertV[v];
        tiltU = d.tiltForceU[v];
        tiltV = d.tiltForceV[v];
        osU = d.ocnStressU[v];
        osV = d.ocnStressV[v];
        uo = d.uOcnVert[v];
        vo = d.vOcnVert[v];
        aV = d.useOcean ? d.iceAreaVertex[v] : 0.0;
    }
    uT[tid] = uu;
    vT[tid] = vv;
    __syncthreads();
    const int *dep = p.dep + p.depStart[tile];
    const int nDep = p.depStart[tile + 1] - p.depStart[tile];
    const double dtE = d.dtE, T = d.dampT, brlx = d.brlx, dtD = d.dtDyn;
    const int relation = d.relation;
    const unsigned int e0 = (unsigned int)epoch;

    for (int it = 1; it <= nIter && alive; it++) {
        const unsigned int target = e0 + (unsigned int)it;
        double *pU = (target & 1u) ? p.partU2 : d.partU, *pV = (target & 1u) ? p.partV2 : d.partV;
        // ================= cells: strain, stress, partial sums of the divergence (k_sub_cells_split)
        const double ug = uT[slot], vg = vT[slot];
        su[g][lane] = corner ? ug : 0.0;
        sv[g][lane] = corner ? vg : 0.0;
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
        if (corner) {
            double cu = 0.0, cv = 0.0;
#pragma unroll
            for (int s = 0; s < M; s++) {
                if (s < E) {
                    const double x11 = s11[s][lane], x22 = s22[s][lane], x12 = s12[s][lane];
                    const double bu = sInt[s][tid], bv = sInt[M + s][tid], bm = sInt[2 * M + s][tid];
                    cu = cu - x11 * bu - x12 * bv - x12 * bm * tg;
                    cv = cv - x22 * bv - x12 * bu + x11 * bm * tg;
                }
            }
            stAgent(&pU[k], cu);
            stAgent(&pV[k], cv);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&p.cellDone[tile], target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        // ================= vertices: divergence, ocean stress coefficient, velocity solve (k_sub_vertices), for
        // every vertex of the tile's cells
        alive = waitTiles(p.cellDone, dep, nDep, target, p.state, limit, &okShared, tid, p.acquire);
        if (!alive) break;
        if (solV) {
            double dU = 0.0, dV = 0.0;
            const double pu0 = ldAgent(&pU[q0]), pu1 = ldAgent(&pU[q1]), pu2 = ldAgent(&pU[q2]);
            const double pv0 = ldAgent(&pV[q0]), pv1 = ldAgent(&pV[q1]), pv2 = ldAgent(&pV[q2]);
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
                // the velocities at the start of the dynamics step: read where they are needed, the revised
                // relation is the rare one and every register counts here
                r1 = dU + airU + tiltU + coeff * osU + (m * (brlx * uu + d.uInit[v])) / dtD;
                r2 = dV + airV + tiltV + coeff * osV + (m * (brlx * vv + d.vInit[v])) / dtD;
            } else {
                l11 = m / dtE + coeff * COS_TURN;
                l22 = m / dtE + coeff * COS_TURN;
                r1 = dU + airU + tiltU + coeff * osU + (m * uu) / dtE;
                r2 = dV + airV + tiltV + coeff * osV + (m * vv) / dtE;
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
    }

    // ---- stresses and velocities go back to memory for whatever runs next (the last iteration of the step, an
    //      exchange); the last sums stay in their buffers, nothing reads them before the next cell kernel
    if (alive) {
        if (corner) {
            d.stress11[k] = o11;
            d.stress22[k] = o22;
            d.stress12[k] = o12;
        }
        if (owned && solV && relation != 3) {
            d.u[v] = uu;
            d.v[v] = vv;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned long long old =
            __hip_atomic_fetch_add(&p.state[PS_FINISH], 1ULL, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (launches + 1ULL) * (unsigned long long)nT - 1ULL) {
            // the last workgroup out moves the iteration base and the launch count on for the next launch
            __hip_atomic_store(&p.state[PS_EPOCH], epoch + (unsigned long long)nIter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&p.state[PS_LAUNCHES], launches + 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------ host
void sih_persist_destroy(sih_ctx *c)
{
    sih_persist_state *s = c->persist;
    if (!s) return;
    for (void *q : s->allocs) (void)hipFree(q);
    delete s;
    c->persist = nullptr;
}

template <int M>
static int occupancyFor(int *perCU)
{
    int n = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_sub_persistent<M>, 64 * M, persistLdsDoubles<M>() * sizeof(double));
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

// Builds the tile tables.  Returns SIH_OK with c->persist->usable = false when the block does not qualify
// (then the two-kernel form runs); an error only for HIP failures.
int sih_persist_setup(sih_ctx *c)
{
    sih_persist_destroy(c);
    sih_persist_state *s = new sih_persist_state();
    c->persist = s;
    const int M = c->M, nC = c->nCells, nV = c->nVertices;
    if (!c->sparseGrad || !c->splitCells || M > 8 || c->cfg.strainScheme != 0 || c->cfg.stressDivergenceScheme != 0 ||
        c->cfg.averageVariationalStrain || c->cfg.useSpecialBoundariesVelocity || c->cfg.useSpecialBoundariesVelocityMasks) {
        s->reason = 1;
        return SIH_OK;
    }
    HIPCHK(hipSetDevice(c->device));
    int perCU = 0, rc = SIH_OK;
    switch (M) {
        case 3: rc = occupancyFor<3>(&perCU); break;
        case 4: rc = occupancyFor<4>(&perCU); break;
        case 5: rc = occupancyFor<5>(&perCU); break;
        case 6: rc = occupancyFor<6>(&perCU); break;
        case 7: rc = occupancyFor<7>(&perCU); break;
        default: rc = occupancyFor<8>(&perCU); break;
    }
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, c->device));
    // the runtime's answer can be one too high for kernels that hold many scalar registers; the probe launch below
    // settles whether this grid is resident at once
    s->perCU = perCU;
    s->capacity = perCU * prop.multiProcessorCount;
    if (const char *e = getenv("SIH_PERSIST_CAPACITY")) s->capacity = atoi(e);
    const int nT = (nC + 63) / 64;
    if (nT < 1 || nT > s->capacity) {
        s->reason = 2;
        return SIH_OK;
    }

    // Every tile lists the vertices of its cells (a vertex shared with neighbouring tiles is in each of their
    // lists); the tile that holds the lowest-numbered cell of a vertex is the one that stores it at the end.
    std::vector<std::vector<int>> cellsOfVertex(nV);
    for (int i = 0; i < nC; i++)
        for (int sIdx = 0; sIdx < c->hNe[i]; sIdx++) {
            const int v = c->hVoc[(size_t)sIdx * c->padC + i];
            if (v < nV) cellsOfVertex[v].push_back(i);
        }
    std::vector<int> vStart(nT + 1, 0), verts, ownedFlag, slot((size_t)M * c->padC, 0);
    std::vector<int> dStart(nT + 1, 0), dList;
    {
        std::vector<int> where(nV, -1);  // position of a vertex in the current tile's list
        for (int t = 0; t < nT; t++) {
            const int first = (int)verts.size();
            std::set<int> deps;
            for (int i = t * 64; i < std::min(nC, (t + 1) * 64); i++)
                for (int sIdx = 0; sIdx < c->hNe[i]; sIdx++) {
                    const int v = c->hVoc[(size_t)sIdx * c->padC + i];
                    if (v >= nV) continue;  // "no vertex": the corner reads the zero slot, which position 0 never is
                    if (where[v] < first) {
                        where[v] = (int)verts.size();
                        verts.push_back(v);
                        int low = nC;
                        for (int cc : cellsOfVertex[v]) {
                            low = std::min(low, cc);
                            deps.insert(cc / 64);
                        }
                        ownedFlag.push_back(low / 64 == t ? 1 : 0);
                    }
                    slot[(size_t)sIdx * c->padC + i] = where[v] - first;
                }
            vStart[t + 1] = (int)verts.size();
            if (vStart[t + 1] - vStart[t] > 64 * M) {  // more vertices than threads: not this block
                s->reason = 3;
                return SIH_OK;
            }
            if (deps.size() > 256) {
                s->reason = 4;
                return SIH_OK;
            }
            dList.insert(dList.end(), deps.begin(), deps.end());
            dStart[t + 1] = (int)dList.size();
        }
    }
    if (verts.empty()) {
        verts.push_back(0);
        ownedFlag.push_back(0);
    }
    if (dList.empty()) dList.push_back(0);
    // the second buffer of the partial sums
    if ((rc = sih_add_field(c, "_partU2", LOC_CELL, M, false))) return rc;
    if ((rc = sih_add_field(c, "_partV2", LOC_CELL, M, false))) return rc;
    PersistArgs &a = s->a;
    a.nTiles = nT;
    a.partU2 = (double *)sih_find_field(c, "_partU2")->d;
    a.partV2 = (double *)sih_find_field(c, "_partV2")->d;
    if ((rc = upload(s, &a.tileVertStart, vStart))) return rc;
    if ((rc = upload(s, &a.tileVerts, verts))) return rc;
    if ((rc = upload(s, &a.tileVertOwned, ownedFlag))) return rc;
    if ((rc = upload(s, &a.cornerSlot, slot))) return rc;
    if ((rc = upload(s, &a.depStart, dStart))) return rc;
    if ((rc = upload(s, &a.dep, dList))) return rc;
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, sizeof(unsigned int) * (size_t)nT));
    s->allocs.push_back(q);
    HIPCHK(hipMemset(q, 0, sizeof(unsigned int) * (size_t)nT));
    a.cellDone = (unsigned int *)q;
    HIPCHK(hipMalloc(&q, sizeof(unsigned long long) * PS_WORDS));
    s->allocs.push_back(q);
    HIPCHK(hipMemset(q, 0, sizeof(unsigned long long) * PS_WORDS));
    a.state = (unsigned long long *)q;
    unsigned long long ticks = 2ULL * 100000000ULL;  // 2 s of the 100 MHz wall clock
    if (const char *e = getenv("SIH_PERSIST_TIMEOUT_MS")) ticks = (unsigned long long)(atof(e) * 100000.0);
    HIPCHK(hipMemcpy(a.state + PS_TICKS, &ticks, sizeof ticks, hipMemcpyHostToDevice));
    a.acquire = getenv("SIH_PERSIST_ACQUIRE") ? atoi(getenv("SIH_PERSIST_ACQUIRE")) : 0;
    // probe: a launch of zero iterations runs the census only.  If its workgroups are not all resident together it
    // says so within the time limit and the two-kernel form stays in use.
    s->usable = true;
    HIPCHK(hipStreamSynchronize(c->stream));
    if ((rc = sih_persist_launch(c, 0))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    unsigned long long st = 0;
    HIPCHK(hipMemcpy(&st, a.state + PS_STATUS, sizeof st, hipMemcpyDeviceToHost));
    if (st != 0) {
        s->usable = false;
        s->reason = 5;
        unsigned long long arr = 0;
        HIPCHK(hipMemcpy(&arr, a.state + PS_ARRIVE, sizeof arr, hipMemcpyDeviceToHost));
        s->arrived = (int)arr;
    }
    return SIH_OK;
}

double sih_persist_info(sih_ctx *c, int what)
{
    if (!c->persist) return 0.0;
    return what == 0 ? c->persist->perCU : what == 1 ? c->persist->reason : what == 2 ? c->persist->a.nTiles : c->persist->arrived;
}

bool sih_persist_usable(sih_ctx *c) { return c->persist && c->persist->usable; }

template <int M>
static void launchPersist(sih_ctx *c, int n)
{
    const PersistArgs &a = c->persist->a;
    hipLaunchKernelGGL((k_sub_persistent<M>), dim3((unsigned)a.nTiles), dim3(64, M), persistLdsDoubles<M>() * sizeof(double), c->stream, makePersistDev(c->dev), a, n);
}

// n iterations (none the last of the dynamics step) as one launch; asynchronous
int sih_persist_launch(sih_ctx *c, int n)
{
    switch (c->M) {
        case 3: launchPersist<3>(c, n); break;
        case 4: launchPersist<4>(c, n); break;
        case 5: launchPersist<5>(c, n); break;
        case 6: launchPersist<6>(c, n); break;
        case 7: launchPersist<7>(c, n); break;
        default: launchPersist<8>(c, n); break;
    }
    HIPCHK(hipGetLastError());
    return SIH_OK;
}

// waits for the stream; SIH_ERR_STATE if a launch gave up (workgroups not resident together, or a wait ran out)
int sih_persist_status(sih_ctx *c)
{
    if (!sih_persist_usable(c)) return SIH_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    unsigned long long st = 0;
    HIPCHK(hipMemcpy(&st, c->persist->a.state + PS_STATUS, sizeof st, hipMemcpyDeviceToHost));
    if (st) {
        sih_set_error(st == 1 ? "persistent subcycle kernel: its workgroups were not resident together"
                              : "persistent subcycle kernel: a wait for a neighbouring tile ran out of time");
        c->persist->usable = false;
        return SIH_ERR_STATE;
    }
    return SIH_OK;
}

// Halo exchange between blocks that live in different processes (one block per GPU) or on different
// devices of one process, done entirely inside the library: what mpas_dmpar_exch_group_full_halo_exch does for
// the reference (the per-subcycle update of uVelocity / vVelocity, velocity_solver.F:2543-2584, the
// once-per-step updates of the pre- and post-cycle, the tracer halo of the advection).
//
// Transport: every block owns one receive allocation that its neighbours map into their address space
// (hipIpcOpenMemHandle across processes, peer access inside one).  A neighbour's "push" kernel gathers the
// listed entries of the fields and stores them straight into its slot of that allocation (over xGMI when the
// two blocks sit on different GPUs), then raises a sequence number in the receiver's flag word; the
// receiver's "pull" kernel waits for the sequence numbers of all its neighbours and scatters the slots into
// its halo entries.  No host synchronisation, no staging copy, no collective: two launches per exchange, both
// capturable in a HIP graph (sequence numbers and buffer parity live in device memory).
//
// Ordering rules the protocol relies on
//   - payload and flags are written and read with system-scope atomic accesses (write-through / cache-bypassing),
//     the allocation is uncached device memory where the runtime offers it;
//   - a pushing workgroup drains its stores (s_waitcnt vmcnt(0)), meets at a barrier, one lane issues a
//     system-scope release fence and counts the workgroup in; the last one raises the neighbours' flags;
//   - a pulling workgroup's lane 0 polls the flags (bounded by a wall-clock limit), issues a system-scope
//     acquire fence, the workgroup meets at a barrier, then loads;
//   - every slot exists twice (parity of the sequence number): a block can be at most one exchange ahead of a
//     neighbour, because its next push is stream-ordered behind its own pull, which needs the neighbour's push.
#include <unistd.h>

#include <algorithm>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "sih_internal.h"

#define SIH_COMM_MAGIC 0x5349484c494e4b31ULL  // "SIHLINK1"
#define SIH_COMM_MAXR 64
#define SIH_FLAG_STRIDE 128  // bytes between two flag words

enum { ST_SEND_SEQ = 0, ST_RECV_SEQ = 1, ST_PUSH_DONE = 2, ST_PULL_DONE = 3, ST_TIMEOUT = 4, ST_SPIN_TICKS = 5, ST_WORDS = 16 };

struct sih_comm_blob {
    unsigned long long magic;
    int rank, nRanks, device, pad0;
    long long pid;
    unsigned long long rawPtr;      // device address in the owning process
    unsigned long long allocBytes, dataOff, halfBytes;
    hipIpcMemHandle_t mem;
    unsigned long long slotOff[SIH_COMM_MAXR];  // byte offset inside one half of the slot for messages from rank p
    unsigned long long slotCap[SIH_COMM_MAXR];  // its capacity in bytes
    // the tagged region (velocities and stresses exchanged from inside the persistent subcycle kernel):
    // SIH_TAGGED_COPIES copies (two banks, see sih_internal.h) of 8-byte words behind the plain halves, then the set-up words (one per received cell and vertex, written
    // once by the sender: its own index of that entity, from which both sides derive the order of the words).
    // Per rank p: cells and vertices received from it, first word of its segment in a half, first set-up word.
    unsigned long long taggedOff, taggedHalfBytes, setupOff;
    unsigned int maxEdges, pad1;
    unsigned int rCntC[SIH_COMM_MAXR], rCntV[SIH_COMM_MAXR];
    unsigned long long tWordOff[SIH_COMM_MAXR], sWordOff[SIH_COMM_MAXR];
};

struct CommTable {  // one field group
    void **d_planes = nullptr;
    int nPlanesC = 0, nPlanesV = 0;
    bool isInt = false;
};

// Receive buffers are uncached (or fine-grained) device memory, and such memory is never given back to the runtime
// while the process lives: ordinary allocations that came to lie on a freed uncached buffer were seen to hand later
// kernels the words of earlier exchanges (fields of contexts created afterwards held halo payloads; with ordinary
// memory for the buffer, or with the buffer kept, never).  A freed context's buffer waits here for the next
// sih_comm_create on the same device; the buffers are a few megabytes.
struct CommBufferSlot {
    char *p;
    size_t bytes;
    int device;
    bool inUse;
};
static std::vector<CommBufferSlot> g_commBuffers;
static std::mutex g_commBuffersMutex;

static char *takeCommBuffer(int device, size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_commBuffersMutex);
    for (CommBufferSlot &b : g_commBuffers)
        if (!b.inUse && b.device == device && b.bytes >= bytes) {
            b.inUse = true;
            return b.p;
        }
    return nullptr;
}
static void keepCommBuffer(char *p, size_t bytes, int device)
{
    std::lock_guard<std::mutex> lock(g_commBuffersMutex);
    g_commBuffers.push_back({p, bytes, device, true});
}
// true if the buffer is one of the kept ones (then it is marked free for reuse and must not be hipFree'd)
static bool releaseCommBuffer(char *p)
{
    std::lock_guard<std::mutex> lock(g_commBuffersMutex);
    for (CommBufferSlot &b : g_commBuffers)
        if (b.p == p) {
            b.inUse = false;
            return true;
        }
    return false;
}

struct sih_comm_state {
    int rank = 0, nRanks = 0;
    int maxPlanesC = 0, maxPlanesV = 0;
    char *base = nullptr;  // receive allocation: [flags][half 0][half 1]
    size_t allocBytes = 0, dataOff = 0, halfBytes = 0;
    std::vector<size_t> slotOff, slotCap;      // receive side, per peer
    std::vector<size_t> peerSlotCap;           // capacity of my slot at peer p
    std::vector<char *> peerBase;              // peer p's allocation in this address space
    std::vector<char> peerIpc;                 // opened with hipIpcOpenMemHandle
    std::vector<int> sendTot, recvTot;         // list entries (cells + vertices) exchanged with peer p
    // device tables
    int *d_sPeer[2] = {nullptr, nullptr}, *d_sRel[2] = {nullptr, nullptr};
    int *d_rPeer[2] = {nullptr, nullptr}, *d_rRel[2] = {nullptr, nullptr};
    int *d_sCnt[2] = {nullptr, nullptr}, *d_rCnt[2] = {nullptr, nullptr};
    char **d_dstBase = nullptr;
    unsigned long long *d_dstHalf = nullptr;
    unsigned long long **d_dstFlag = nullptr;
    unsigned long long *d_srcOff = nullptr;
    int *d_sendPeers = nullptr, *d_recvPeers = nullptr;
    int nSendPeers = 0, nRecvPeers = 0;
    unsigned long long *d_state = nullptr;
    bool connected = false;
    int generation = 0;  // changes whenever the lists or the peers' addresses may have changed
    size_t taggedOff = 0, taggedHalfBytes = 0, setupOff = 0, setupWords = 0;
    std::vector<size_t> tWordOff, sWordOff;              // receive side, per peer: first word of its segment / set-up words
    std::vector<char *> peerTagged;                      // peer p's tagged region (half 0) in this address space
    std::vector<size_t> peerTHalf, peerTWordOff;         // its half size; first word of my segment there
    std::vector<int> sendPos[2], recvPos[2];             // per list entry: its place in the sender's own order (recv: once known)
    bool recvPosKnown = false;
    std::map<std::string, CommTable> tables;
    hipStream_t commStream = nullptr;
    hipEvent_t evFork = nullptr, evJoin = nullptr;
};

struct PushArgs {
    void *const *planes;
    int nfC, nfV;
    const int *idxC, *peerC, *relC;
    const int *idxV, *peerV, *relV;
    int nC, nV;
    const int *cntC, *cntV;  // per peer
    char *const *dstBase;
    const unsigned long long *dstHalf;
    unsigned long long *const *dstFlag;
    const int *sendPeers;
    int nSendPeers;
    unsigned long long *state;
};

struct PullArgs {
    void *const *planes;
    int nfC, nfV;
    const int *idxC, *peerC, *relC;
    const int *idxV, *peerV, *relV;
    int nC, nV;
    const int *cntC, *cntV;
    const char *base;  // my data region (half 0)
    unsigned long long halfBytes;
    const unsigned long long *srcOff;
    const unsigned long long *flags;  // my flag words, SIH_FLAG_STRIDE apart
    const int *recvPeers;
    int nRecvPeers;
    unsigned long long *state;
};

// T: unsigned int for int32 fields, unsigned long long for float64 fields (the values travel as raw bits)
template <typename T>
__global__ void __launch_bounds__(256) k_halo_push(PushArgs a)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int fi = blockIdx.y;
    const unsigned long long seq = __hip_atomic_load(&a.state[ST_SEND_SEQ], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ULL;
    const unsigned long long par = seq & 1ULL;
    if (fi < a.nfC) {
        if (t < a.nC) {
            const int p = a.peerC[t];
            const size_t off = (size_t)fi * a.cntC[p] + a.relC[t];
            T *dst = (T *)(a.dstBase[p] + par * a.dstHalf[p]) + off;
            const T val = ((const T *)a.planes[fi])[a.idxC[t]];
            __hip_atomic_store(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    } else if (t < a.nV) {
        const int p = a.peerV[t];
        const size_t off = (size_t)a.nfC * a.cntC[p] + (size_t)(fi - a.nfC) * a.cntV[p] + a.relV[t];
        T *dst = (T *)(a.dstBase[p] + par * a.dstHalf[p]) + off;
        const T val = ((const T *)a.planes[fi])[a.idxV[t]];
        __hip_atomic_store(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long total = (unsigned long long)gridDim.x * gridDim.y;
        const unsigned long long old =
            __hip_atomic_fetch_add(&a.state[ST_PUSH_DONE], 1ULL, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == total - 1ULL) {
            // every workgroup's stores are out: publish
            __hip_atomic_store(&a.state[ST_PUSH_DONE], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.state[ST_SEND_SEQ], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            for (int k = 0; k < a.nSendPeers; k++)
                __hip_atomic_store(a.dstFlag[a.sendPeers[k]], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_halo_pull(PullArgs a)
{
    __shared__ int ok;
    const int fi = blockIdx.y;
    const unsigned long long seq = __hip_atomic_load(&a.state[ST_RECV_SEQ], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ULL;
    const unsigned long long par = seq & 1ULL;
    if (threadIdx.x == 0) {
        // bounded wait: every neighbour's flag must have reached this exchange's sequence number
        const unsigned long long limit = __hip_atomic_load(&a.state[ST_SPIN_TICKS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        int good = 1;
        for (int k = 0; k < a.nRecvPeers && good; k++) {
            const unsigned long long *f =
                (const unsigned long long *)((const char *)a.flags + (size_t)a.recvPeers[k] * SIH_FLAG_STRIDE);
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
                if (wall_clock64() - t0 > limit) {
                    good = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!good) __hip_atomic_store(&a.state[ST_TIMEOUT], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = good;
    }
    __syncthreads();
    if (ok) {
        // a workgroup takes every gridDim.x-th run of 256 entries: the launch is a few hundred workgroups whatever
        // the size of the halo (all of them wait above)
        const char *half = a.base + par * a.halfBytes;
        const int stride = gridDim.x * blockDim.x;
        if (fi < a.nfC) {
            for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < a.nC; t += stride) {
                const int p = a.peerC[t];
                const size_t off = (size_t)fi * a.cntC[p] + a.relC[t];
                const T *src = (const T *)(half + a.srcOff[p]) + off;
                ((T *)a.planes[fi])[a.idxC[t]] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        } else {
            for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < a.nV; t += stride) {
                const int p = a.peerV[t];
                const size_t off = (size_t)a.nfC * a.cntC[p] + (size_t)(fi - a.nfC) * a.cntV[p] + a.relV[t];
                const T *src = (const T *)(half + a.srcOff[p]) + off;
                ((T *)a.planes[fi])[a.idxV[t]] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long total = (unsigned long long)gridDim.x * gridDim.y;
        const unsigned long long old =
            __hip_atomic_fetch_add(&a.state[ST_PULL_DONE], 1ULL, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == total - 1ULL) {
            __hip_atomic_store(&a.state[ST_PULL_DONE], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.state[ST_RECV_SEQ], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------
static int upInts(int **dst, const std::vector<int> &v)
{
    if (*dst) {
        HIPCHK(hipFree(*dst));
        *dst = nullptr;
    }
    HIPCHK(hipMalloc((void **)dst, sizeof(int) * std::max<size_t>(v.size(), 1)));
    if (!v.empty()) HIPCHK(hipMemcpy(*dst, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice));
    return SIH_OK;
}

void sih_comm_free(sih_ctx *c)
{
    sih_comm_state *s = c->comm;
    if (!s) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (s->commStream) (void)hipStreamSynchronize(s->commStream);
    for (int p = 0; p < (int)s->peerBase.size(); p++)
        if (s->peerBase[p] && s->peerIpc[p]) (void)hipIpcCloseMemHandle(s->peerBase[p]);
    for (int k = 0; k < 2; k++)
        for (int *q : {s->d_sPeer[k], s->d_sRel[k], s->d_rPeer[k], s->d_rRel[k], s->d_sCnt[k], s->d_rCnt[k]})
            if (q) (void)hipFree(q);
    for (void *q : {(void *)s->d_dstBase, (void *)s->d_dstHalf, (void *)s->d_dstFlag, (void *)s->d_srcOff,
                    (void *)s->d_sendPeers, (void *)s->d_recvPeers, (void *)s->d_state})
        if (q) (void)hipFree(q);
    for (auto &kv : s->tables)
        if (kv.second.d_planes) (void)hipFree(kv.second.d_planes);
    if (s->base && !releaseCommBuffer(s->base)) (void)hipFree(s->base);
    if (s->evFork) (void)hipEventDestroy(s->evFork);
    if (s->evJoin) (void)hipEventDestroy(s->evJoin);
    if (s->commStream) (void)hipStreamDestroy(s->commStream);
    delete s;
    c->comm = nullptr;
}

extern "C" size_t sih_comm_handle_size(void) { return sizeof(sih_comm_blob); }

extern "C" int sih_comm_create(sih_ctx *c, int rank, int nRanks, int maxCellPlanes, int maxVertexPlanes)
{
    if (!c || nRanks < 1 || nRanks > SIH_COMM_MAXR || rank < 0 || rank >= nRanks || maxCellPlanes < 0 ||
        maxVertexPlanes < 0) {
        sih_set_error("sih_comm_create: bad argument (at most 64 ranks)");
        return SIH_ERR_ARG;
    }
    for (int loc = 0; loc < 2; loc++)
        if ((int)c->hSendCounts[loc].size() != nRanks || (int)c->hRecvCounts[loc].size() != nRanks) {
            sih_set_error("sih_comm_create: call sih_halo_set_lists for cells and vertices first, with one count per rank");
            return SIH_ERR_STATE;
        }
    HIPCHK(hipSetDevice(c->device));
    sih_comm_free(c);
    sih_drop_graphs(c);
    sih_comm_state *s = new sih_comm_state();
    c->comm = s;
    s->rank = rank;
    s->nRanks = nRanks;
    s->maxPlanesC = maxCellPlanes;
    s->maxPlanesV = maxVertexPlanes;
    s->slotOff.assign(nRanks, 0);
    s->slotCap.assign(nRanks, 0);
    s->sendTot.assign(nRanks, 0);
    s->recvTot.assign(nRanks, 0);
    size_t off = 0;
    for (int p = 0; p < nRanks; p++) {
        s->sendTot[p] = c->hSendCounts[0][p] + c->hSendCounts[1][p];
        s->recvTot[p] = c->hRecvCounts[0][p] + c->hRecvCounts[1][p];
        size_t cap = ((size_t)maxCellPlanes * c->hRecvCounts[0][p] + (size_t)maxVertexPlanes * c->hRecvCounts[1][p]) * 8;
        cap = (cap + 255) & ~(size_t)255;
        s->slotOff[p] = off;
        s->slotCap[p] = cap;
        off += cap;
    }
    if (s->sendTot[rank] || s->recvTot[rank]) {
        sih_set_error("sih_comm_create: a block does not exchange with itself");
        return SIH_ERR_ARG;
    }
    // the two halves of a slot are enough only while a sender cannot run more than one exchange ahead of its
    // receiver, and what holds it back is its own pull waiting for that receiver's push
    for (int p = 0; p < nRanks; p++)
        if ((s->sendTot[p] > 0) != (s->recvTot[p] > 0)) {
            char msg[200];
            snprintf(msg, sizeof msg, "sih_comm_create: block %d %s block %d without %s it (halos of MPAS blocks are mutual; "
                     "the exchange's two buffer halves rely on it)", rank, s->sendTot[p] > 0 ? "sends to" : "receives from", p,
                     s->sendTot[p] > 0 ? "receiving from" : "sending to");
            sih_set_error(msg);
            delete s;
            c->comm = nullptr;
            return SIH_ERR_ARG;
        }
    s->halfBytes = std::max<size_t>(off, 256);
    s->dataOff = ((size_t)nRanks * SIH_FLAG_STRIDE + 4095) & ~(size_t)4095;
    // tagged words: per received cell the three stresses at every corner, per received vertex the two velocities,
    // two words a value
    s->tWordOff.assign(nRanks, 0);
    s->sWordOff.assign(nRanks, 0);
    size_t tWords = 0, sWords = 0;
    for (int p = 0; p < nRanks; p++) {
        s->tWordOff[p] = tWords;
        s->sWordOff[p] = sWords;
        tWords += (size_t)6 * c->M * c->hRecvCounts[0][p] + (size_t)4 * c->hRecvCounts[1][p];
        sWords += (size_t)c->hRecvCounts[0][p] + (size_t)c->hRecvCounts[1][p];
    }
    s->taggedHalfBytes = (std::max<size_t>(tWords * 8, 256) + 255) & ~(size_t)255;
    s->taggedOff = s->dataOff + 2 * s->halfBytes;
    s->setupOff = s->taggedOff + SIH_TAGGED_COPIES * s->taggedHalfBytes;
    s->setupWords = sWords;
    s->allocBytes = s->setupOff + ((std::max<size_t>(sWords * 8, 256) + 255) & ~(size_t)255);
    static int generations = 0;
    s->generation = ++generations;
    // uncached device memory where the runtime has it (remote stores and local polls then never meet a stale
    // cache line); fine-grained or ordinary memory otherwise -- every access is a system-scope atomic anyway
    void *p = takeCommBuffer(c->device, s->allocBytes);
    if (!p) {
        hipError_t e = hipExtMallocWithFlags(&p, s->allocBytes, hipDeviceMallocUncached);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipExtMallocWithFlags(&p, s->allocBytes, hipDeviceMallocFinegrained);
        }
        if (e == hipSuccess) {
            keepCommBuffer((char *)p, s->allocBytes, c->device);
        } else {
            (void)hipGetLastError();
            HIPCHK(hipMalloc(&p, s->allocBytes));
        }
    }
    s->base = (char *)p;
    HIPCHK(hipMemset(s->base, 0, s->allocBytes));
    HIPCHK(hipMalloc((void **)&s->d_state, sizeof(unsigned long long) * ST_WORDS));
    HIPCHK(hipMemset(s->d_state, 0, sizeof(unsigned long long) * ST_WORDS));
    unsigned long long ticks = 10ULL * 100000000ULL;  // 10 s of the 100 MHz wall clock
    if (const char *ev = getenv("SIH_HALO_TIMEOUT_MS")) ticks = (unsigned long long)atof(ev) * 100000ULL;
    HIPCHK(hipMemcpy(s->d_state + ST_SPIN_TICKS, &ticks, sizeof ticks, hipMemcpyHostToDevice));

    // per-entry tables: which peer an entry of a list belongs to and its position inside that peer's segment
    for (int loc = 0; loc < 2; loc++) {
        for (int dir = 0; dir < 2; dir++) {
            const std::vector<int> &cnt = dir == 0 ? c->hSendCounts[loc] : c->hRecvCounts[loc];
            std::vector<int> peer, rel;
            for (int q = 0; q < nRanks; q++)
                for (int i = 0; i < cnt[q]; i++) {
                    peer.push_back(q);
                    rel.push_back(i);
                }
            int rc;
            if ((rc = upInts(dir == 0 ? &s->d_sPeer[loc] : &s->d_rPeer[loc], peer))) return rc;
            if ((rc = upInts(dir == 0 ? &s->d_sRel[loc] : &s->d_rRel[loc], rel))) return rc;
            if ((rc = upInts(dir == 0 ? &s->d_sCnt[loc] : &s->d_rCnt[loc], cnt))) return rc;
        }
    }
    std::vector<int> sp, rp;
    for (int q = 0; q < nRanks; q++) {
        if (s->sendTot[q] > 0) sp.push_back(q);
        if (s->recvTot[q] > 0) rp.push_back(q);
    }
    s->nSendPeers = (int)sp.size();
    s->nRecvPeers = (int)rp.size();
    int rc;
    if ((rc = upInts(&s->d_sendPeers, sp))) return rc;
    if ((rc = upInts(&s->d_recvPeers, rp))) return rc;
    std::vector<unsigned long long> so(nRanks);
    for (int q = 0; q < nRanks; q++) so[q] = s->slotOff[q];
    HIPCHK(hipMalloc((void **)&s->d_srcOff, sizeof(unsigned long long) * nRanks));
    HIPCHK(hipMemcpy(s->d_srcOff, so.data(), sizeof(unsigned long long) * nRanks, hipMemcpyHostToDevice));
    HIPCHK(hipStreamCreate(&s->commStream));
    HIPCHK(hipEventCreateWithFlags(&s->evFork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&s->evJoin, hipEventDisableTiming));
    return SIH_OK;
}

extern "C" int sih_comm_get_handle(sih_ctx *c, void *blob)
{
    if (!c || !c->comm || !blob) {
        sih_set_error("sih_comm_get_handle: no communicator");
        return SIH_ERR_STATE;
    }
    sih_comm_state *s = c->comm;
    HIPCHK(hipSetDevice(c->device));
    sih_comm_blob b;
    memset(&b, 0, sizeof b);
    b.magic = SIH_COMM_MAGIC;
    b.rank = s->rank;
    b.nRanks = s->nRanks;
    b.device = c->device;
    b.pid = (long long)getpid();
    b.rawPtr = (unsigned long long)(uintptr_t)s->base;
    b.allocBytes = s->allocBytes;
    b.dataOff = s->dataOff;
    b.halfBytes = s->halfBytes;
    HIPCHK(hipIpcGetMemHandle(&b.mem, s->base));
    b.taggedOff = s->taggedOff;
    b.taggedHalfBytes = s->taggedHalfBytes;
    b.setupOff = s->setupOff;
    b.maxEdges = (unsigned int)c->M;
    for (int p = 0; p < s->nRanks; p++) {
        b.slotOff[p] = s->slotOff[p];
        b.slotCap[p] = s->slotCap[p];
        b.rCntC[p] = (unsigned int)c->hRecvCounts[0][p];
        b.rCntV[p] = (unsigned int)c->hRecvCounts[1][p];
        b.tWordOff[p] = s->tWordOff[p];
        b.sWordOff[p] = s->sWordOff[p];
    }
    memcpy(blob, &b, sizeof b);
    return SIH_OK;
}

// one-time stores of the set-up words into the neighbours' memory
__global__ void k_setup_store(const unsigned long long *dst, const unsigned long long *val, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) __hip_atomic_store((unsigned long long *)(uintptr_t)dst[i], val[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int sih_comm_connect(sih_ctx *c, const void *allHandles)
{
    if (!c || !c->comm || !allHandles) {
        sih_set_error("sih_comm_connect: no communicator");
        return SIH_ERR_STATE;
    }
    sih_comm_state *s = c->comm;
    HIPCHK(hipSetDevice(c->device));
    const int n = s->nRanks;
    const sih_comm_blob *all = (const sih_comm_blob *)allHandles;
    s->peerBase.assign(n, nullptr);
    s->peerIpc.assign(n, 0);
    s->peerSlotCap.assign(n, 0);
    s->peerTagged.assign(n, nullptr);
    s->peerTHalf.assign(n, 0);
    s->peerTWordOff.assign(n, 0);
    s->recvPosKnown = false;
    // the order in which this block's kernel meets the entities it sends: by its own (internal) index
    std::vector<unsigned long long> setupDst;
    std::vector<unsigned long long> setupVal;
    for (int loc = 0; loc < 2; loc++) {
        s->sendPos[loc].assign(c->hSend[loc].size(), 0);
        size_t at = 0;
        for (int p = 0; p < n; p++) {
            const int cnt = c->hSendCounts[loc][p];
            std::vector<std::pair<int, int>> byIndex(cnt);
            for (int r = 0; r < cnt; r++) byIndex[r] = {c->hSend[loc][at + r], r};
            std::sort(byIndex.begin(), byIndex.end());
            for (int k = 0; k < cnt; k++) s->sendPos[loc][at + byIndex[k].second] = k;
            at += cnt;
        }
    }
    std::vector<char *> dstBase(n, nullptr);
    std::vector<unsigned long long> dstHalf(n, 0);
    std::vector<unsigned long long *> dstFlag(n, nullptr);
    for (int p = 0; p < n; p++) {
        sih_comm_blob b;
        memcpy(&b, all + p, sizeof b);
        if (b.magic != SIH_COMM_MAGIC || b.rank != p || b.nRanks != n) {
            sih_set_error("sih_comm_connect: handle table is not in rank order or comes from another library version");
            return SIH_ERR_ARG;
        }
        if (p == s->rank || s->sendTot[p] == 0) continue;  // nothing goes to this rank
        char *base = nullptr;
        if (b.pid == (long long)getpid()) {
            base = (char *)(uintptr_t)b.rawPtr;
            if (b.device != c->device) {
                hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return sih_hip_fail(e, "hipDeviceEnablePeerAccess", __FILE__, __LINE__);
                (void)hipGetLastError();
            }
        } else {
            void *q = nullptr;
            HIPCHK(hipIpcOpenMemHandle(&q, b.mem, hipIpcMemLazyEnablePeerAccess));
            base = (char *)q;
            s->peerIpc[p] = 1;
        }
        s->peerBase[p] = base;
        s->peerSlotCap[p] = b.slotCap[s->rank];
        // the tagged segment the peer keeps for this rank must be what this rank will send
        if (b.maxEdges == (unsigned int)c->M && b.rCntC[s->rank] == (unsigned int)c->hSendCounts[0][p] &&
            b.rCntV[s->rank] == (unsigned int)c->hSendCounts[1][p]) {
            s->peerTagged[p] = base + b.taggedOff;
            s->peerTHalf[p] = b.taggedHalfBytes;
            s->peerTWordOff[p] = b.tWordOff[s->rank];
            // set-up words for this peer: the internal index of every entity this rank sends to it, in list order
            size_t atC = 0, atV = 0;
            for (int q = 0; q < p; q++) {
                atC += c->hSendCounts[0][q];
                atV += c->hSendCounts[1][q];
            }
            unsigned long long *w = (unsigned long long *)(base + b.setupOff) + b.sWordOff[s->rank];
            for (int r = 0; r < c->hSendCounts[0][p]; r++) {
                setupDst.push_back((unsigned long long)(uintptr_t)(w + r));
                setupVal.push_back((unsigned long long)(unsigned int)c->hSend[0][atC + r] | (1ULL << 32));
            }
            for (int r = 0; r < c->hSendCounts[1][p]; r++) {
                setupDst.push_back((unsigned long long)(uintptr_t)(w + c->hSendCounts[0][p] + r));
                setupVal.push_back((unsigned long long)(unsigned int)c->hSend[1][atV + r] | (1ULL << 32));
            }
        }
        dstBase[p] = base + b.dataOff + b.slotOff[s->rank];
        dstHalf[p] = b.halfBytes;
        dstFlag[p] = (unsigned long long *)(base + (size_t)s->rank * SIH_FLAG_STRIDE);
    }
    for (void *q : {(void *)s->d_dstBase, (void *)s->d_dstHalf, (void *)s->d_dstFlag})
        if (q) (void)hipFree(q);
    HIPCHK(hipMalloc((void **)&s->d_dstBase, sizeof(char *) * n));
    HIPCHK(hipMalloc((void **)&s->d_dstHalf, sizeof(unsigned long long) * n));
    HIPCHK(hipMalloc((void **)&s->d_dstFlag, sizeof(unsigned long long *) * n));
    HIPCHK(hipMemcpy(s->d_dstBase, dstBase.data(), sizeof(char *) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_dstHalf, dstHalf.data(), sizeof(unsigned long long) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_dstFlag, dstFlag.data(), sizeof(unsigned long long *) * n, hipMemcpyHostToDevice));
    if (!setupDst.empty()) {
        unsigned long long *dDst = nullptr, *dVal = nullptr;
        const size_t nb = sizeof(unsigned long long) * setupDst.size();
        HIPCHK(hipMalloc((void **)&dDst, nb));
        HIPCHK(hipMalloc((void **)&dVal, nb));
        HIPCHK(hipMemcpy(dDst, setupDst.data(), nb, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dVal, setupVal.data(), nb, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_setup_store, dim3((unsigned)((setupDst.size() + 255) / 256)), dim3(256), 0, c->stream, dDst, dVal,
                           (int)setupDst.size());
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));  // the words are in the neighbours' memory when this call returns
        HIPCHK(hipFree(dDst));
        HIPCHK(hipFree(dVal));
    }
    s->connected = true;
    static int connects = 1000000;
    s->generation = ++connects;
    return SIH_OK;
}

// what the persistent subcycle kernel needs of the transport for its in-kernel exchange; false when a peer's
// segment does not match this rank's send lists (then the launches exchange between the chunks as before)
bool sih_comm_tagged_info(sih_ctx *c, sih_comm_tagged *out)
{
    sih_comm_state *s = c->comm;
    if (!s || !s->connected) return false;
    out->generation = s->generation;
    out->nRanks = s->nRanks;
    for (int p = 0; p < s->nRanks; p++)
        if (s->sendTot[p] > 0 && !s->peerTagged[p]) return false;
    if (!s->recvPosKnown) {
        // the neighbours' set-up words: all there once every neighbour has connected
        std::vector<unsigned long long> w(std::max<size_t>(s->setupWords, 1));
        if (hipStreamSynchronize(c->stream) != hipSuccess) return false;
        if (s->setupWords &&
            hipMemcpy(w.data(), s->base + s->setupOff, sizeof(unsigned long long) * s->setupWords, hipMemcpyDeviceToHost) != hipSuccess)
            return false;
        for (size_t i = 0; i < s->setupWords; i++)
            if ((w[i] >> 32) != 1ULL) {
                out->generation = 0;  // not yet: ask again later
                return false;
            }
        for (int loc = 0; loc < 2; loc++) {
            s->recvPos[loc].assign(c->hRecv[loc].size(), 0);
            size_t at = 0;
            for (int p = 0; p < s->nRanks; p++) {
                const int cnt = c->hRecvCounts[loc][p];
                const size_t first = s->sWordOff[p] + (loc == 1 ? (size_t)c->hRecvCounts[0][p] : 0);
                std::vector<std::pair<unsigned int, int>> byIndex(cnt);
                for (int r = 0; r < cnt; r++) byIndex[r] = {(unsigned int)(w[first + r] & 0xffffffffULL), r};
                std::sort(byIndex.begin(), byIndex.end());
                for (int k = 0; k < cnt; k++) s->recvPos[loc][at + byIndex[k].second] = k;
                at += cnt;
            }
        }
        s->recvPosKnown = true;
    }
    out->myBase = s->base + s->taggedOff;
    out->myHalfBytes = s->taggedHalfBytes;
    out->myWordOff = s->tWordOff;
    out->peerBase = s->peerTagged;
    out->peerHalfBytes = s->peerTHalf;
    out->peerWordOff = s->peerTWordOff;
    for (int loc = 0; loc < 2; loc++) {
        out->sendPos[loc] = s->sendPos[loc];
        out->recvPos[loc] = s->recvPos[loc];
    }
    return true;
}

extern "C" int sih_comm_set_timeout(sih_ctx *c, double milliseconds)
{
    if (!c || !c->comm || !(milliseconds > 0.0)) {
        sih_set_error("sih_comm_set_timeout: no communicator or bad value");
        return SIH_ERR_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    unsigned long long ticks = (unsigned long long)(milliseconds * 100000.0);
    HIPCHK(hipMemcpy(c->comm->d_state + ST_SPIN_TICKS, &ticks, sizeof ticks, hipMemcpyHostToDevice));
    return SIH_OK;
}

extern "C" int sih_comm_status(sih_ctx *c)
{
    if (!c || !c->comm) return SIH_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    unsigned long long st[ST_WORDS];
    HIPCHK(hipMemcpy(st, c->comm->d_state, sizeof st, hipMemcpyDeviceToHost));
    if (st[ST_TIMEOUT]) {
        char buf[160];
        snprintf(buf, sizeof buf, "halo exchange %llu: a neighbour's message did not arrive within the time limit",
                 st[ST_TIMEOUT]);
        sih_set_error(buf);
        return SIH_ERR_COMM;
    }
    // the exchange inside the persistent subcycle kernel keeps its own status word: a launch that gave up (a
    // neighbour's words late or overwritten) is an error of the transport as well
    return sih_persist_status(c);
}

extern "C" int sih_comm_destroy(sih_ctx *c)
{
    if (!c) return SIH_ERR_ARG;
    sih_comm_free(c);
    sih_drop_graphs(c);
    return SIH_OK;
}

// plane pointers of a field group: cell fields first, then vertex fields, in the order given
static int commTable(sih_ctx *c, int nFields, const char *const *names, CommTable **out)
{
    sih_comm_state *s = c->comm;
    std::string key;
    for (int f = 0; f < nFields; f++) {
        key += '|';
        key += names[f] ? names[f] : "";
    }
    auto it = s->tables.find(key);
    if (it == s->tables.end()) {
        std::vector<void *> planes[2];
        bool isInt = false;
        for (int f = 0; f < nFields; f++) {
            SihField *fl = names[f] ? sih_find_field(c, names[f]) : nullptr;
            if (!fl || (fl->loc != LOC_CELL && fl->loc != LOC_VERTEX)) {
                sih_set_error(std::string("sih_halo_exchange: not a cell or vertex field: ") + (names[f] ? names[f] : "(null)"));
                return SIH_ERR_ARG;
            }
            if (f == 0) isInt = fl->isInt;
            if (fl->isInt != isInt) {
                sih_set_error("sih_halo_exchange: fields of one call must share a type");
                return SIH_ERR_ARG;
            }
            const int pad = fl->loc == LOC_CELL ? c->padC : c->padV;
            for (int l = 0; l < fl->lead; l++)
                planes[fl->loc == LOC_CELL ? 0 : 1].push_back((char *)fl->d + (size_t)l * pad * (isInt ? 4 : 8));
        }
        CommTable tb;
        tb.isInt = isInt;
        tb.nPlanesC = (int)planes[0].size();
        tb.nPlanesV = (int)planes[1].size();
        const size_t es = isInt ? 4 : 8;
        for (int p = 0; p < s->nRanks; p++) {
            const size_t need = ((size_t)tb.nPlanesC * c->hSendCounts[0][p] + (size_t)tb.nPlanesV * c->hSendCounts[1][p]) * es;
            if (s->sendTot[p] > 0 && need > s->peerSlotCap[p]) {
                sih_set_error("sih_halo_exchange: the field group is larger than the plane counts given to sih_comm_create");
                return SIH_ERR_ARG;
            }
        }
        std::vector<void *> all(planes[0]);
        all.insert(all.end(), planes[1].begin(), planes[1].end());
        HIPCHK(hipMalloc((void **)&tb.d_planes, sizeof(void *) * std::max<size_t>(all.size(), 1)));
        if (!all.empty()) HIPCHK(hipMemcpy(tb.d_planes, all.data(), sizeof(void *) * all.size(), hipMemcpyHostToDevice));
        it = s->tables.emplace(key, tb).first;
    }
    *out = &it->second;
    return SIH_OK;
}

static int checkComm(sih_ctx *c, const char *who, int nFields, const char *const *names)
{
    if (!c || !c->comm || !c->comm->connected) {
        sih_set_error(std::string(who) + ": call sih_comm_create and sih_comm_connect first");
        return SIH_ERR_STATE;
    }
    if (nFields < 1 || !names) {
        sih_set_error(std::string(who) + ": bad argument");
        return SIH_ERR_ARG;
    }
    return SIH_OK;
}

bool sih_comm_ready(sih_ctx *c) { return c && c->comm && c->comm->connected; }
hipStream_t sih_comm_stream(sih_ctx *c) { return c->comm->commStream; }
void sih_comm_events(sih_ctx *c, hipEvent_t *fork, hipEvent_t *join)
{
    *fork = c->comm->evFork;
    *join = c->comm->evJoin;
}
int sih_comm_prepare(sih_ctx *c, int nFields, const char *const *names)
{
    CommTable *tb;
    return commTable(c, nFields, names, &tb);
}

int sih_comm_push(sih_ctx *c, int nFields, const char *const *names, hipStream_t stream)
{
    sih_comm_state *s = c->comm;
    CommTable *tb;
    int rc = commTable(c, nFields, names, &tb);
    if (rc) return rc;
    const int nC = tb->nPlanesC ? c->nSend[0] : 0, nV = tb->nPlanesV ? c->nSend[1] : 0;
    PushArgs a;
    a.planes = tb->d_planes;
    a.nfC = tb->nPlanesC;
    a.nfV = tb->nPlanesV;
    a.idxC = c->d_send[0];
    a.peerC = s->d_sPeer[0];
    a.relC = s->d_sRel[0];
    a.idxV = c->d_send[1];
    a.peerV = s->d_sPeer[1];
    a.relV = s->d_sRel[1];
    a.nC = nC;
    a.nV = nV;
    a.cntC = s->d_sCnt[0];
    a.cntV = s->d_sCnt[1];
    a.dstBase = s->d_dstBase;
    a.dstHalf = s->d_dstHalf;
    a.dstFlag = s->d_dstFlag;
    a.sendPeers = s->d_sendPeers;
    a.nSendPeers = s->nSendPeers;
    a.state = s->d_state;
    // at least one workgroup runs even when there is nothing to send: it keeps the sequence numbers moving
    dim3 grid((unsigned)std::max(1, (std::max(nC, nV) + 255) / 256), (unsigned)std::max(1, tb->nPlanesC + tb->nPlanesV));
    if (tb->isInt)
        hipLaunchKernelGGL(k_halo_push<unsigned int>, grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(k_halo_push<unsigned long long>, grid, dim3(256), 0, stream, a);
    HIPCHK(hipGetLastError());
    return SIH_OK;
}

int sih_comm_pull(sih_ctx *c, int nFields, const char *const *names, hipStream_t stream)
{
    sih_comm_state *s = c->comm;
    CommTable *tb;
    int rc = commTable(c, nFields, names, &tb);
    if (rc) return rc;
    const int nC = tb->nPlanesC ? c->nRecv[0] : 0, nV = tb->nPlanesV ? c->nRecv[1] : 0;
    PullArgs a;
    a.planes = tb->d_planes;
    a.nfC = tb->nPlanesC;
    a.nfV = tb->nPlanesV;
    a.idxC = c->d_recv[0];
    a.peerC = s->d_rPeer[0];
    a.relC = s->d_rRel[0];
    a.idxV = c->d_recv[1];
    a.peerV = s->d_rPeer[1];
    a.relV = s->d_rRel[1];
    a.nC = nC;
    a.nV = nV;
    a.cntC = s->d_rCnt[0];
    a.cntV = s->d_rCnt[1];
    a.base = s->base + s->dataOff;
    a.halfBytes = s->halfBytes;
    a.srcOff = s->d_srcOff;
    a.flags = (const unsigned long long *)s->base;
    a.recvPeers = s->d_recvPeers;
    a.nRecvPeers = s->nRecvPeers;
    a.state = s->d_state;
    // Every workgroup of this kernel waits for the neighbours' flags before it copies, so its size is bounded (about
    // one workgroup a CU in all) and the workgroups stride over the entries: sized by the halo, a deep halo of a large
    // block made thousands of waiting waves, and several blocks sharing one device (ranks rehearsed on one GPU)
    // could fill it with them and keep out the very kernels they were waiting for.
    const int nPlanes = std::max(1, tb->nPlanesC + tb->nPlanesV);
    const int needX = std::max(1, (std::max(nC, nV) + 255) / 256);
    dim3 grid((unsigned)std::min(needX, std::max(1, 256 / nPlanes)), (unsigned)nPlanes);
    if (tb->isInt)
        hipLaunchKernelGGL(k_halo_pull<unsigned int>, grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(k_halo_pull<unsigned long long>, grid, dim3(256), 0, stream, a);
    HIPCHK(hipGetLastError());
    return SIH_OK;
}

extern "C" int sih_halo_exchange_start(sih_ctx *c, int nFields, const char *const *names)
{
    int rc = checkComm(c, "sih_halo_exchange_start", nFields, names);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    return sih_comm_push(c, nFields, names, c->stream);
}

extern "C" int sih_halo_exchange_finish(sih_ctx *c, int nFields, const char *const *names)
{
    int rc = checkComm(c, "sih_halo_exchange_finish", nFields, names);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    return sih_comm_pull(c, nFields, names, c->stream);
}

extern "C" int sih_halo_exchange(sih_ctx *c, int nFields, const char *const *names)
{
    int rc = checkComm(c, "sih_halo_exchange", nFields, names);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    if ((rc = sih_comm_push(c, nFields, names, c->stream))) return rc;
    return sih_comm_pull(c, nFields, names, c->stream);
}

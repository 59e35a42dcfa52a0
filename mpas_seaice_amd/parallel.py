"""Halo exchange between MPAS blocks: the transport the reference gets from ``mpas_dmpar_*halo_exch``
(/root/reference/src/shared/mpas_seaice_velocity_solver.F:2543-2584 and the once-per-step exchanges).

Two transports share the same exchange lists (partition.BlockHalo):
  * LocalComm  - all blocks live in this process (several blocks on one GPU, or host arrays);
  * DistComm   - one block per process, ``torch.distributed`` all_to_all_single: RCCL on GPUs
                 (backend "nccl"), gloo on host tensors.
Device blocks pack / unpack through the C ABI (sih_halo_pack / sih_halo_unpack) into torch device
buffers; host blocks (``HostFields``) index numpy pool arrays, which is what the CPU-side tests use.
"""
from __future__ import annotations

import os

import numpy as np

from . import cases
from .partition import partition_mesh

LOC_ID = {"cell": 0, "vertex": 1}

# location and integer-ness of the fields the velocity solver exchanges
FIELD_LOC = {
    "iceAreaCellInitial": "cell", "totalMassCell": "cell", "icePressure": "cell", "airStressCellU": "cell",
    "airStressCellV": "cell", "seaSurfaceTiltU": "cell", "seaSurfaceTiltV": "cell",
    "solveVelocity": "vertex", "interiorVertex": "vertex", "uVelocity": "vertex", "vVelocity": "vertex",
    "oceanStressU": "vertex", "oceanStressV": "vertex",
    "stress11": "cell", "stress22": "cell", "stress12": "cell", "solveStress": "cell",
}
INT_FIELDS = {"solveVelocity", "interiorVertex", "solveStress"}
FIELD_POOL = {
    "iceAreaCellInitial": "icestate", "totalMassCell": "icestate", "icePressure": "velocity_solver",
    "airStressCellU": "velocity_solver", "airStressCellV": "velocity_solver", "seaSurfaceTiltU": "ocean_coupling",
    "seaSurfaceTiltV": "ocean_coupling", "solveVelocity": "velocity_solver", "interiorVertex": "boundary",
    "uVelocity": "velocity_solver", "vVelocity": "velocity_solver", "oceanStressU": "velocity_solver",
    "oceanStressV": "velocity_solver", "stress11": "velocity_variational", "stress22": "velocity_variational",
    "stress12": "velocity_variational", "solveStress": "velocity_solver",
}


def field_loc(name):
    """Fields not listed above are advected tracers: cell fields of the `tracers` pool."""
    return FIELD_LOC.get(name, "cell")


def field_pool(name):
    if name.endswith("LocalMin") or name.endswith("LocalMax"):
        return "tracer_monotonicity"  # bounds of the advection's monotonicity check, one pair per tracer
    return FIELD_POOL.get(name, "tracers")


def _planes_of(a):
    """(lead..., n) Fortran-ordered pool array -> 2-D view with one row per leading index combination,
    first index fastest (the plane order of the device library)."""
    if a.ndim == 1:
        return a.reshape((1, -1))
    return a.reshape((-1, a.shape[-1]), order="F")


def _group(names):
    groups = {}
    for n in names:
        groups.setdefault((field_loc(n), n in INT_FIELDS), []).append(n)
    return groups


class _Lists:
    """Concatenated send / recv lists of one block for one location, neighbours in ascending order."""

    def __init__(self, halo, loc, nblocks):
        self.peers = list(range(nblocks))
        self.send_counts = [len(halo.send[loc].get(p, ())) for p in self.peers]
        self.recv_counts = [len(halo.recv[loc].get(p, ())) for p in self.peers]
        cat = lambda d: (np.concatenate([d[p] for p in self.peers if p in d]).astype(np.int32)  # noqa: E731
                         if any(p in d for p in self.peers) else np.zeros(0, dtype=np.int32))
        self.send_idx = cat(halo.send[loc])
        self.recv_idx = cat(halo.recv[loc])
        self.send_off = np.concatenate([[0], np.cumsum(self.send_counts)]).astype(np.int64)
        self.recv_off = np.concatenate([[0], np.cumsum(self.recv_counts)]).astype(np.int64)


class HostFields:
    """pack / unpack on numpy pool arrays (used with the CPU oracle flow in tests); same neighbour-major
    message layout as sih_halo_pack: [peer][plane][entry], a field with several entries per entity
    (stress11: maxEdges per cell) contributing one plane per entry."""

    def __init__(self, block):
        self.block = block

    def planes(self, names):
        out = []
        for n in names:
            a2 = _planes_of(self.block.structs[field_pool(n)][n])
            out.extend(a2[l] for l in range(a2.shape[0]))
        return out

    def pack(self, lists, names, is_int):
        pl = self.planes(names)
        nf = len(pl)
        out = np.empty(nf * lists.send_idx.size, dtype=np.int32 if is_int else np.float64)
        for p in lists.peers:
            o, c = int(lists.send_off[p]), lists.send_counts[p]
            for f, a in enumerate(pl):
                out[o * nf + f * c:o * nf + (f + 1) * c] = a[lists.send_idx[o:o + c] - 1]
        return out

    def unpack(self, lists, names, buf):
        pl = self.planes(names)
        nf = len(pl)
        for p in lists.peers:
            o, c = int(lists.recv_off[p]), lists.recv_counts[p]
            for f, a in enumerate(pl):
                a[lists.recv_idx[o:o + c] - 1] = buf[o * nf + f * c:o * nf + (f + 1) * c]


class CommBase:
    def __init__(self, domain, halos, nblocks):
        self.nblocks = nblocks
        self.lists = []
        for b, h in zip(domain.blocklist, halos):
            self.lists.append({loc: _Lists(h, loc, nblocks) for loc in ("cell", "vertex")})
            b.halo = h
        self.device = all(getattr(b, "ctx", None) is not None for b in domain.blocklist)
        if self.device:
            for b, L in zip(domain.blocklist, self.lists):
                for loc in ("cell", "vertex"):
                    b.ctx.halo_set_lists(LOC_ID[loc], L[loc].send_idx, L[loc].recv_idx, L[loc].send_counts,
                                         L[loc].recv_counts)
        self._bufs = {}

    def n_planes(self, domain, names):
        tot = 0
        for n in names:
            a = domain.blocklist[0].structs[field_pool(n)][n]
            tot += 1 if a.ndim == 1 else int(np.prod(a.shape[:-1]))
        return tot

    def halo_exchange(self, domain, names):
        for _ in self._steps(domain, names):
            pass

    def start_exchange(self, domain, names):
        """Pack and start the transfer; work queued on the blocks' streams before ``finish()`` of the
        returned object runs while the messages travel and must not touch the receive lists' entities."""
        gen = self._steps(domain, names)
        next(gen, None)
        return _Pending(gen)

    def _torch_buf(self, key, n, is_int, device):
        import torch
        k = (key, is_int)
        t = self._bufs.get(k)
        if t is None or t.numel() < max(n, 1):
            if t is not None:
                # pack / unpack kernels of the library's streams may still read or write the buffer that is about to
                # be released: torch's allocator knows nothing of those streams
                torch.cuda.synchronize()
            t = torch.empty(max(n, 1), dtype=torch.int32 if is_int else torch.float64, device=device)
            self._bufs[k] = t
        return t


class _Pending:
    """An exchange whose messages are packed and on their way; ``finish()`` waits for them and unpacks."""

    def __init__(self, gen):
        self._gen = gen

    def finish(self):
        if self._gen is not None:
            for _ in self._gen:
                pass
            self._gen = None


def _combined(names):
    """Cell and vertex float fields of one exchange travel in one message (device path)."""
    g = _group(names)
    return len(g) == 2 and ("cell", False) in g and ("vertex", False) in g


class LocalComm(CommBase):
    """All blocks in this process; block i of the partition is domain.blocklist[i]."""

    def allreduce_min(self, x):
        return x

    def allreduce_sum(self, a):
        return a

    def _steps_combined(self, domain, names):
        import torch
        g = _group(names)
        nsC, nsV = g[("cell", False)], g[("vertex", False)]
        nfC, nfV = self.n_planes(domain, nsC), self.n_planes(domain, nsV)
        ordered = nsC + nsV
        dev = torch.device("cuda", torch.cuda.current_device())
        sends, recvs, soff, roff = [], [], [], []
        for i, b in enumerate(domain.blocklist):
            LC, LV = self.lists[i]["cell"], self.lists[i]["vertex"]
            sc = np.array(LC.send_counts) * nfC + np.array(LV.send_counts) * nfV
            rc = np.array(LC.recv_counts) * nfC + np.array(LV.recv_counts) * nfV
            soff.append(np.concatenate([[0], np.cumsum(sc)]))
            roff.append(np.concatenate([[0], np.cumsum(rc)]))
            s = self._torch_buf(("s2", i), int(sc.sum()), False, dev)
            r = self._torch_buf(("r2", i), int(rc.sum()), False, dev)
            b.ctx.halo_pack(2, ordered, s.data_ptr())
            sends.append(s)
            recvs.append(r)
        yield
        for b in domain.blocklist:
            b.ctx.sync()
        for i in range(self.nblocks):
            for p in range(self.nblocks):
                cnt = int(roff[i][p + 1] - roff[i][p])
                if cnt:
                    ro, so = int(roff[i][p]), int(soff[p][i])
                    recvs[i][ro:ro + cnt].copy_(sends[p][so:so + cnt])
        torch.cuda.synchronize()
        for i, b in enumerate(domain.blocklist):
            b.ctx.halo_unpack(2, ordered, recvs[i].data_ptr())

    def _steps(self, domain, names):
        """Generator: everything up to the first ``yield`` packs, the rest transfers and unpacks."""
        if self.device and _combined(names):
            yield from self._steps_combined(domain, names)
            return
        groups = list(_group(names).items())
        for gi, ((loc, is_int), ns) in enumerate(groups):
            nf = self.n_planes(domain, ns)
            if self.device:
                import torch
                dev = torch.device("cuda", torch.cuda.current_device())
                sends, recvs = [], []
                for i, b in enumerate(domain.blocklist):
                    L = self.lists[i][loc]
                    s = self._torch_buf(("s", i, loc), nf * L.send_idx.size, is_int, dev)
                    r = self._torch_buf(("r", i, loc), nf * L.recv_idx.size, is_int, dev)
                    b.ctx.halo_pack(LOC_ID[loc], ns, s.data_ptr())
                    sends.append(s)
                    recvs.append(r)
                if gi == len(groups) - 1:
                    yield
                for b in domain.blocklist:
                    b.ctx.sync()
                for i in range(self.nblocks):
                    Li = self.lists[i][loc]
                    for p in range(self.nblocks):
                        cnt = Li.recv_counts[p] * nf
                        if cnt:
                            Lp = self.lists[p][loc]
                            ro, so = int(Li.recv_off[p]) * nf, int(Lp.send_off[i]) * nf
                            recvs[i][ro:ro + cnt].copy_(sends[p][so:so + cnt])
                torch.cuda.synchronize()
                for i, b in enumerate(domain.blocklist):
                    b.ctx.halo_unpack(LOC_ID[loc], ns, recvs[i].data_ptr())
            else:
                hf = [HostFields(b) for b in domain.blocklist]
                sends = [hf[i].pack(self.lists[i][loc], ns, is_int) for i in range(self.nblocks)]
                for i in range(self.nblocks):
                    Li = self.lists[i][loc]
                    r = np.empty(nf * Li.recv_idx.size, dtype=sends[i].dtype)
                    for p in range(self.nblocks):
                        cnt = Li.recv_counts[p] * nf
                        if cnt:
                            Lp = self.lists[p][loc]
                            ro, so = int(Li.recv_off[p]) * nf, int(Lp.send_off[i]) * nf
                            r[ro:ro + cnt] = sends[p][so:so + cnt]
                    hf[i].unpack(Li, ns, r)


class DistComm(CommBase):
    """One block per process; neighbour exchange as a single all_to_all_single per field group."""

    def __init__(self, domain, halo, nblocks, rank):
        import torch.distributed as dist
        assert len(domain.blocklist) == 1
        self.rank = rank
        self.dist = dist
        super().__init__(domain, [halo], nblocks)
        if self.device:
            # torch orders its transfers (RCCL's own stream, or the copies that stage gloo messages) against
            # its current stream, so the block works on that stream too.  The legacy default stream has no
            # handle to pass and synchronises with every other stream: use an explicit one instead.
            import torch
            if torch.cuda.current_stream().cuda_stream == 0:
                self._stream = torch.cuda.Stream()
                torch.cuda.set_stream(self._stream)
            domain.blocklist[0].ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def allreduce_min(self, x):
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return float(t.item())

    def allreduce_sum(self, a):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def _a2a(self, r, s, nr_tot, ns_tot, recv_split, send_split, b):
        """Start the transfer; returns a callable that completes it (device buffers)."""
        import torch
        if self.dist.get_backend() == "gloo":
            # gloo moves host memory only: stage through the host with synchronous copies
            b.ctx.sync()
            sh = s[:ns_tot].cpu()
            rh = torch.empty(nr_tot, dtype=sh.dtype)
            self.dist.all_to_all_single(rh, sh, recv_split, send_split)

            def done():
                r[:nr_tot].copy_(rh)
                torch.cuda.synchronize()
            return done
        # RCCL runs the transfer on its own stream, ordered after what is queued on the current stream
        # so far; wait() makes the current stream wait for it without blocking the host
        work = self.dist.all_to_all_single(r[:nr_tot], s[:ns_tot], recv_split, send_split, async_op=True)
        return work.wait

    def _steps(self, domain, names):
        """Generator: everything up to the first ``yield`` packs and starts the transfer, the rest
        waits for it and unpacks."""
        import torch
        b = domain.blocklist[0]
        if self.device and _combined(names):
            g = _group(names)
            nsC, nsV = g[("cell", False)], g[("vertex", False)]
            key = tuple(names)
            plan = self._plans.get(key) if hasattr(self, "_plans") else None
            if plan is None:
                if not hasattr(self, "_plans"):
                    self._plans = {}
                nfC, nfV = self.n_planes(domain, nsC), self.n_planes(domain, nsV)
                LC, LV = self.lists[0]["cell"], self.lists[0]["vertex"]
                send_split = [c * nfC + v * nfV for c, v in zip(LC.send_counts, LV.send_counts)]
                recv_split = [c * nfC + v * nfV for c, v in zip(LC.recv_counts, LV.recv_counts)]
                plan = self._plans[key] = (nsC + nsV, send_split, recv_split, sum(send_split), sum(recv_split))
            ordered, send_split, recv_split, ns_tot, nr_tot = plan
            dev = torch.device("cuda", torch.cuda.current_device())
            s = self._torch_buf(("s2",), ns_tot, False, dev)
            r = self._torch_buf(("r2",), nr_tot, False, dev)
            b.ctx.halo_pack(2, ordered, s.data_ptr())
            done = self._a2a(r, s, nr_tot, ns_tot, recv_split, send_split, b)
            yield
            done()
            b.ctx.halo_unpack(2, ordered, r.data_ptr())
            return
        groups = list(_group(names).items())
        for gi, ((loc, is_int), ns) in enumerate(groups):
            L = self.lists[0][loc]
            nf = self.n_planes(domain, ns)
            send_split = [c * nf for c in L.send_counts]
            recv_split = [c * nf for c in L.recv_counts]
            ns_tot, nr_tot = nf * L.send_idx.size, nf * L.recv_idx.size
            if self.device:
                dev = torch.device("cuda", torch.cuda.current_device())
                s = self._torch_buf(("s", loc), ns_tot, is_int, dev)
                r = self._torch_buf(("r", loc), nr_tot, is_int, dev)
                b.ctx.halo_pack(LOC_ID[loc], ns, s.data_ptr())
                done = self._a2a(r, s, nr_tot, ns_tot, recv_split, send_split, b)
                if gi == len(groups) - 1:
                    yield
                done()
                b.ctx.halo_unpack(LOC_ID[loc], ns, r.data_ptr())
            else:
                hf = HostFields(b)
                st = torch.from_numpy(hf.pack(L, ns, is_int))
                rt = torch.empty(nr_tot, dtype=st.dtype)
                self.dist.all_to_all_single(rt, st, recv_split, send_split)
                hf.unpack(L, ns, rt.numpy())


class PeerComm(DistComm):
    """One block per process; the halo messages travel inside the library (sih_halo_exchange): every neighbour
    stores its message straight into this block's receive buffer, mapped through HIP IPC (over xGMI between
    GPUs), and the receiver waits for it on the device.  ``torch.distributed`` only carries the 2.7 KB handles
    at set-up and the scalar reductions; no collective and no host synchronisation is on the data path."""
    in_library = True  # the subcycle loop can run with its exchanges inside the library

    def __init__(self, domain, halo, nblocks, rank, max_cell_planes=None, max_vertex_planes=4):
        super().__init__(domain, halo, nblocks, rank)
        if not self.device:
            raise RuntimeError("PeerComm needs device blocks")
        b = domain.blocklist[0]
        if max_cell_planes is None:
            tracer_planes = sum(int(np.prod(a.shape[:-1])) for a in b.structs["tracers"].values()
                                if isinstance(a, np.ndarray) and a.ndim >= 2)
            max_cell_planes = max(3 * int(b.dimensions["maxEdges"]) + 4, tracer_planes)
        b.ctx.comm_create(rank, nblocks, max_cell_planes, max_vertex_planes)
        mine = b.ctx.comm_handle()
        if nblocks > 1:
            handles = [None] * nblocks
            self.dist.all_gather_object(handles, mine)
        else:
            handles = [mine]
        b.ctx.comm_connect(handles)
        if nblocks > 1:
            self.dist.barrier()  # every rank has mapped its neighbours before the first message

    def _steps(self, domain, names):
        b = domain.blocklist[0]
        groups = {}
        for n in names:
            groups.setdefault(n in INT_FIELDS, []).append(n)
        groups = list(groups.values())
        for ns in groups[:-1]:
            b.ctx.halo_exchange(ns)
        b.ctx.halo_exchange_start(groups[-1])
        yield
        b.ctx.halo_exchange_finish(groups[-1])

    def check(self, domain):
        """raises if a neighbour's message ever failed to arrive (waits for the block's stream)"""
        domain.blocklist[0].ctx.comm_status()

    def agree_on_exchange_inside(self, domain, allow=True):
        """The persistent subcycle kernel can carry the halo exchange itself, but only if every block of the run
        does so: the ranks vote once (after sih_evp_init, a minimum over all of them) and set the mode on all
        blocks or on none.  Returns what was agreed."""
        ctx = domain.blocklist[0].ctx
        ready = 1 if (allow and os.environ.get("SIH_PERSIST_EXCHANGE", "1") != "0"
                      and os.environ.get("SIH_PERSISTENT", "1") != "0"
                      and ctx.info("persistentExchangeReady") == 1.0) else 0
        agreed = (self.allreduce_min(float(ready)) if self.nblocks > 1 else float(ready)) >= 1.0
        ctx.set_persistent(2 if agreed else -1)
        self.exchange_inside = agreed
        return agreed


# ----------------------------------------------------------------------------------------------
def make_partitioned_domain(mesh, name, nblocks, configs=None, only_block=None, cell_owner=None, **case_kw):
    """Domain holding all blocks of the partition (only_block=None) or just one of them.  ``cell_owner``:
    block of every cell, e.g. from a METIS ``graph.info.part.N`` file (mesh_io.read_graph_partition);
    default is a cut along a space-filling curve."""
    num_halos = int((configs or {}).get("config_num_halos", 2))
    meshes, halos = partition_mesh(mesh, nblocks, num_halos=num_halos, cell_owner=cell_owner)
    if only_block is not None:
        dom = cases.make_domain([meshes[only_block]], name, configs=configs, **case_kw)
        return dom, halos[only_block], halos
    dom = cases.make_domain(meshes, name, configs=configs, **case_kw)
    return dom, halos, halos


def gather_global(domains_blocks, halos, pool, name, n_global, loc):
    """Assemble a global field from the owned parts of block fields (for comparisons)."""
    first = domains_blocks[0].structs[pool][name]
    lead = first.shape[:-1]
    out = np.zeros(lead + (n_global,), dtype=first.dtype)
    for b, h in zip(domains_blocks, halos):
        nown = h.nOwned[loc]
        out[..., h.globalID[loc][:nown]] = b.structs[pool][name][..., :nown]
    return out


def oracle_halo_callback(comm):
    """Adapter: the oracle flow calls halo(domain, pool, names)."""
    def halo(domain, pool, names):
        comm.halo_exchange(domain, names)
    return halo


def link_self_test(dom, halo, comm, rounds=3):
    """Push known patterns through the transport and compare every halo entry with what its owner holds.
    Returns the number of wrong entries on this rank (0 on a working link)."""
    b = dom.blocklist[0]
    nV, nC, M = b.dimensions["nVertices"], b.dimensions["nCells"], b.dimensions["maxEdges"]
    gv, gc = halo.globalID["vertex"], halo.globalID["cell"]
    bad = 0
    for r in range(rounds):
        u = np.zeros(nV + 1)
        v = np.zeros(nV + 1)
        s11 = np.zeros((M, nC + 1), order="F")
        own_v, own_c = halo.nOwned["vertex"], halo.nOwned["cell"]
        u[:own_v] = np.sin(0.001 * gv[:own_v] + r)
        v[:own_v] = gv[:own_v] * (r + 1.0)
        s11[:, :own_c] = (np.arange(M)[:, None] + 1.0) * (gc[None, :own_c] + 0.25 * r)
        u[own_v:nV] = v[own_v:nV] = -1.0e30
        s11[:, own_c:nC] = -1.0e30
        b.ctx.put_many([("uVelocity", u), ("vVelocity", v), ("stress11", s11)])
        comm.halo_exchange(dom, ["uVelocity", "vVelocity", "stress11"])
        b.ctx.get_many([("uVelocity", u), ("vVelocity", v), ("stress11", s11)])
        # entries in the receive lists must now hold their owners' values; entries no list names keep the marker
        rv = comm.lists[0]["vertex"].recv_idx - 1
        rc = comm.lists[0]["cell"].recv_idx - 1
        bad += int((u[rv] != np.sin(0.001 * gv[rv] + r)).sum() + (v[rv] != gv[rv] * (r + 1.0)).sum())
        bad += int((s11[:, rc] != (np.arange(M)[:, None] + 1.0) * (gc[None, rc] + 0.25 * r)).sum())
    return bad


def thinner_halo_that_fits(mesh, nblocks, interval, room_cells, cell_owner=None, candidates=(4, 2)):
    """The halo depth to try first: `interval` if the largest block of the partition fits `room_cells` with it (or
    with no candidate either), else the deepest candidate below it with which every block fits.  Every rank computes
    the same answer from the same partition."""
    def largest_block(k):
        _, halos = partition_mesh(mesh, nblocks, num_halos=max(2, k), cell_owner=cell_owner)
        return max(len(h.globalID["cell"]) for h in halos)
    if largest_block(interval) <= room_cells:
        return interval
    for k in candidates:
        if k < interval and largest_block(k) <= room_cells:
            return k
    return interval


def bench_multi_gpu(mesh, name, nsub, steps, warmup, local_rank, renumber=True, interval=None, cell_owner=None):
    """One block per rank: K steps of nsub subcycles; halos are `interval` layers deep and are exchanged every
    `interval` subcycles (uVelocity, vVelocity, stress11/22/12).  Transport (SIH_TRANSPORT): "p2p", the library's
    own (messages stored into the neighbours' IPC-mapped receive buffers, the whole loop with its exchanges
    queued by sih_evp_subcycle_exchange), checked with known patterns before it is used; or "torch",
    all_to_all_single of torch.distributed (RCCL), which is also the fall-back when the check fails."""
    import os
    import time

    import torch
    import torch.distributed as dist

    from . import velocity_solver as vs
    interval_from_env = os.environ.get("SIH_HALO_INTERVAL") if interval is None else str(interval)
    if interval is None:
        # Measured on one GPU (scripts/bench_block.py, QU60, block of an 8-way partition): 14.3 / 14.6 / 14.8 /
        # 16.2 us per subcycle with 2 / 4 / 8 / 16 halo layers; an exchange costs the same two launches whatever
        # the depth, so 8 layers win for any exchange cost between 5 and 20 us.
        interval = int(os.environ.get("SIH_HALO_INTERVAL", "8"))
    use_graph = os.environ.get("SIH_MULTI_GRAPH", "1") != "0"
    backend = os.environ.get("SIH_DIST_BACKEND", "nccl")  # "gloo" only for rehearsals on a box with one GPU
    transport = os.environ.get("SIH_TRANSPORT", "p2p")
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    def reduce_scalar(x, op):
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=op)
        return float(t.item())
    rank, world = dist.get_rank(), dist.get_world_size()
    overlap = os.environ.get("SIH_OVERLAP", "0") != "0"
    state = {}
    # which rank sits on which device, as the ranks themselves see it
    props = torch.cuda.get_device_properties(local_rank)
    mine = {"rank": rank, "device": int(local_rank), "name": props.name, "uuid": str(getattr(props, "uuid", "")),
            "pid": os.getpid()}
    rank_table = [None] * world
    dist.all_gather_object(rank_table, mine)

    def sync():
        torch.cuda.synchronize()
        state["ctx"].sync()  # the block's own stream (the p2p transport does not run on torch's)

    def step():
        ctx, dom, k = state["ctx"], state["dom"], state["interval"]
        if state["transport"] == "p2p":
            ctx.subcycle_exchange(nsub, last=False, interval=k, overlap=overlap, use_graph=use_graph)
            return
        if overlap:
            vs.subcycle_overlapped(dom, dom.comm, nsub, k, state["names"], use_graph=use_graph)
            return
        done = 0
        while done < nsub:
            m = min(k, nsub - done)
            if use_graph and m == k:
                ctx.subcycle_graph(m)
            else:
                ctx.subcycle(m, last=False)
            done += m
            dom.comm.halo_exchange(dom, state["names"])

    def setup(k, transport):
        """partition with k halo layers, block on the device, transport, pre-cycle; then the vote on the exchange
        inside the kernel and its check on this machine's links"""
        cfg = {"config_elastic_subcycle_number": nsub, "config_num_halos": max(2, k),
               "config_halo_exchange_interval": k, "config_overlap_halo_exchange": overlap,
               "config_subcycle_graph": use_graph}
        dom, halo, _ = make_partitioned_domain(mesh, name, world, configs=cfg, only_block=rank, cell_owner=cell_owner)
        vs.seaice_mesh_pool_create(dom, device=local_rank, renumber=renumber)
        ctx = dom.blocklist[0].ctx
        note = ""
        if transport == "p2p":
            ok = 1.0
            try:
                if os.environ.get("SIH_TEST_REFUSE_P2P"):  # tests of the fall-back below
                    raise RuntimeError("refused by SIH_TEST_REFUSE_P2P")
                dom.comm = PeerComm(dom, halo, world, rank)
            except Exception as e:  # noqa: BLE001 -- whatever fails, every rank must take the same branch
                ok, note = 0.0, f"p2p set-up failed on rank {rank}: {e}"
            if reduce_scalar(ok, dist.ReduceOp.MIN) < 1.0:
                transport = "torch"
            else:
                bad = 0
                try:  # a message that never arrives ends a bounded wait with an error: that, too, is a failed check
                    bad = link_self_test(dom, halo, dom.comm)
                    dom.comm.check(dom)
                except Exception as e:  # noqa: BLE001
                    bad, note = bad + 1, f"rank {rank}: {e}"
                if reduce_scalar(float(bad), dist.ReduceOp.MAX) > 0.0:
                    transport, note = "torch", note or "p2p link check found wrong halo values"
                    dist.barrier()
                    ctx.comm_destroy()
        if transport != "p2p":
            dom.comm = DistComm(dom, halo, world, rank)  # also moves the block onto torch's (explicit) current stream
        vs.seaice_init_velocity_solver(dom)
        for b in dom.blocklist:
            vs._upload(dom, b)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        vs.velocity_solver_pre_subcycle(dom)
        torch.cuda.synchronize()
        ctx.sync()
        pre_ms = (time.perf_counter() - t0) * 1e3
        names = ["uVelocity", "vVelocity"] if k == 1 else ["uVelocity", "vVelocity", "stress11", "stress22", "stress12"]
        state.update(dom=dom, halo=halo, ctx=ctx, interval=k, transport=transport, note=note, pre_ms=pre_ms, names=names)
        # Small blocks can run all iterations of a step as one launch with the exchanges inside it.  All ranks or
        # none: they vote, and then check it on this machine's links -- one step each way from the same state must
        # give the same bits on every rank -- before the timed loop relies on it.
        inside, inside_note = False, ""
        if transport == "p2p":
            inside = dom.comm.agree_on_exchange_inside(dom)
            if inside:
                b0 = dom.blocklist[0]
                nV1, nC1, M = b0.dimensions["nVertices"] + 1, b0.dimensions["nCells"] + 1, b0.dimensions["maxEdges"]
                shapes = {n: ((nV1,) if n.endswith("Velocity") else (M, nC1)) for n in names}

                def grab():
                    out = {}
                    for n, sh in shapes.items():
                        out[n] = np.zeros(sh, order="F")
                        ctx.get(n, out[n])
                    return out

                def put_back(st):
                    for n, a in st.items():
                        ctx.put(n, a)
                start = grab()
                res, bad = [], 0.0
                for mode in (1, 2):
                    put_back(start)
                    ctx.set_persistent(mode)
                    sync()
                    dist.barrier()
                    try:
                        step()
                        sync()
                        dom.comm.check(dom)
                        res.append(grab())
                    except Exception as e:  # noqa: BLE001 -- every rank must reach the vote below
                        bad, inside_note = 1.0, f"rank {rank}, mode {mode}: {e}"  # (no way out of the loop: its barriers)
                if not bad and any(not np.array_equal(res[0][n], res[1][n]) for n in names):
                    bad, inside_note = 1.0, f"rank {rank}: the exchange inside the kernel changed the results"
                inside = reduce_scalar(bad, dist.ReduceOp.MAX) == 0.0
                ctx.set_persistent(2 if inside else 1)
                put_back(start)
                sync()
                dist.barrier()
        state.update(inside=inside, inside_note=inside_note)

    def teardown():
        dom = state["dom"]
        # what is torn down may be a trial that failed (a launch that gave up, a message that never came): its errors
        # have been counted in the vote, and every rank must reach the barriers below
        for wait in (sync, (lambda: dom.comm.check(dom)) if state["transport"] == "p2p" else sync):
            try:
                wait()
            except Exception:  # noqa: BLE001
                pass
        dist.barrier()
        vs.seaice_mesh_pool_destroy(dom)
        dist.barrier()

    # The halo depth: 8 layers, one exchange per 8 iterations, unless a thinner halo lets every block stay resident in
    # the persistent kernel with the exchanges inside it (QU60 on 4 GPUs: 46.8 k cells with 8 layers, 42.4 k with 2,
    # against room for 43 008) -- tried first where the sizes say so, kept only if every rank confirms it.
    first = interval
    if interval_from_env is None and transport == "p2p" and int(mesh.maxEdges) == 6 \
            and os.environ.get("SIH_PERSISTENT", "1") != "0" and os.environ.get("SIH_PERSIST_EXCHANGE", "1") != "0":
        room = 4 * 42 * torch.cuda.get_device_properties(local_rank).multi_processor_count
        room = int(os.environ.get("SIH_BENCH_ROOM_CELLS", room))  # (rehearsals of this choice on small meshes)
        first = thinner_halo_that_fits(mesh, world, interval, room, cell_owner=cell_owner)
    setup(first, transport)
    if first != interval and not state["inside"]:
        thin_note = (f"{first} halo layers tried for the persistent kernel: " +
                     (state["inside_note"] or "not every block could stay resident"))
        teardown()
        setup(interval, transport)
        state["note"] = (state["note"] + "; " if state["note"] else "") + thin_note
    dom, halo, ctx = state["dom"], state["halo"], state["ctx"]
    transport, note, pre_ms, names = state["transport"], state["note"], state["pre_ms"], state["names"]
    interval, inside, inside_note = state["interval"], state["inside"], state["inside_note"]
    for _ in range(warmup):
        step()
    sync()
    dist.barrier()
    sync()
    launches0 = ctx.info("persistentLaunches")
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dist.barrier()
    sync()
    wall = time.perf_counter() - t0
    launches_timed = ctx.info("persistentLaunches") - launches0
    form_timed = ctx.info("iterationForm")  # of the last loop of the timed steps: 0 two kernels, 1 persistent, 2 fused
    wall = reduce_scalar(wall, dist.ReduceOp.MAX)
    if transport == "p2p":
        dom.comm.check(dom)  # raises if any message of the timed loop failed to arrive
    t0 = time.perf_counter()
    ctx.subcycle(1, last=True)
    dom.comm.halo_exchange(dom, names)
    vs.velocity_solver_post_subcycle(dom)
    sync()
    post_ms = (time.perf_counter() - t0) * 1e3
    # the cost of one exchange: 50 of them queued back to back (each needs the neighbours' previous one, so the
    # chain runs at the pace of launch + transfer + flag latency), host-timed over the whole chain
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(50):
        dom.comm.halo_exchange(dom, names)
    sync()
    ex_us = reduce_scalar((time.perf_counter() - t0) / 50 * 1e6, dist.ReduceOp.MAX)
    _, cell_ms, vert_ms = ctx.subcycle_timed(8, use_graph=False, sample_every=1)
    u = np.zeros(dom.blocklist[0].dimensions["nVertices"] + 1)
    ctx.get("uVelocity", u)
    if reduce_scalar(1.0 if np.isfinite(u).all() else 0.0, dist.ReduceOp.MIN) < 1.0:
        raise SystemExit("non-finite velocities after the benchmark loop")
    n_block_cells = dom.blocklist[0].dimensions["nCells"]
    persistent_launches = reduce_scalar(launches_timed, dist.ReduceOp.MIN)  # of the timed loop, on every rank
    form_all = reduce_scalar(form_timed, dist.ReduceOp.MIN)
    form_same = reduce_scalar(form_timed, dist.ReduceOp.MAX) == form_all
    if transport == "p2p":
        dom.comm.check(dom)
    dist.barrier()
    vs.seaice_mesh_pool_destroy(dom)
    dist.barrier()
    real_backend = dist.get_backend()
    dist.destroy_process_group()
    if transport == "p2p" and inside:
        how = ("every step one launch of the persistent subcycle kernel; its edge tiles store velocities and stresses "
               "into the neighbours' IPC-mapped receive buffers (peer stores over xGMI) as tagged words and poll for "
               "theirs, the other tiles never wait for a neighbour")
    elif transport == "p2p":
        how = ("messages stored by the sender into the neighbours' IPC-mapped receive buffers (peer stores over xGMI), "
               "sequence flags awaited on the device, iterations and exchanges replayed as HIP graphs")
    else:
        how = f"torch.distributed all_to_all_single over {real_backend}" + (" (RCCL)" if real_backend == "nccl" else "")
    return {"wall": wall, "pre_ms": pre_ms, "post_ms": post_ms, "cell_ms": cell_ms, "vert_ms": vert_ms,
            "n_cells_total": int(mesh.nCells), "n_block_cells": int(n_block_cells), "interval": interval,
            "overlap": overlap, "graph": bool(use_graph), "transport": transport, "transport_text": how,
            "transport_note": note, "exchange_us": ex_us, "rendezvous_backend": real_backend,
            "exchange_inside_kernel": bool(inside), "exchange_inside_note": inside_note,
            "world": int(world), "ranks": rank_table,
            "persistent_kernel": bool(persistent_launches > 0 or (form_same and form_all == 1.0)),
            "fused_kernel": bool(form_same and form_all == 2.0)}

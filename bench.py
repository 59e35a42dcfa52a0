#!/usr/bin/env python3
"""EVP subcycle benchmark: cell-subcycle-updates/s at QU60 on N MI355X GPUs.

One "step" = one dynamics step's subcycle loop = 120 EVP subcycles (strain, stress, stress
divergence, ocean stress coefficient, velocity solve [+ halo exchange when N > 1]) over the QU60
mesh (163 842 cells) with the synthetic idealised forcing of mpas_seaice_amd.cases.  Pre- and
post-subcycle stages are outside the timed region and reported separately (pre_ms / post_ms).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak; ~6300 GB/s is what a streaming copy reaches


def b_alg_per_cell(M):
    """Algorithmic (compulsory) bytes per cell per subcycle, dense basis storage, two fused kernels:
    cell part 16 + 4M + 16M^2 + 56M + 24*(nV/nC), vertex part (nV/nC)*(148+16) + 4 + 24M + 24M^2 with
    nV/nC = 2 (M = 6: 1000 + 1340 = 2340 B)."""
    cell = 16 + 4 * M + 16 * M * M + 56 * M + 24 * 2
    vert = 2 * (148 + 16) + 4 + 24 * M + 24 * M * M
    return cell, vert


def b_alg_cell_kernel(M):
    """Share of those bytes moved by this library's fused cell kernel (strain + stress + the inner
    sums of the stress divergence): both gradient matrices, the three integral matrices, stress read
    and write, replacement pressure, connectivity, ice strength and the vertex velocities it gathers."""
    return 16 * M * M + 24 * M * M + 48 * M + 8 * M + (4 * M + 8) + 8 + 24 * 2


def b_layout_cell_kernel(M, symmetric_metric=True, derived_gradient=True, derived_row=True):
    """Bytes per cell this library's layout makes the cell kernel move: like b_alg_cell_kernel, but the basis
    gradients are stored compactly (three non-zeros per column: 48 M instead of 16 M^2 bytes), the per-corner
    partial sums of the stress divergence are written (16 M), and of the metric integrals, mirrored at set-up to be
    symmetric bit for bit, one side of the diagonal is fetched (4 M (M + 1) instead of 8 M^2 bytes); of the three
    gradients of a corner two are fetched; of the M rows of basisIntegralsU / V all but the first corner's."""
    metric = 4 * M * (M + 1) if symmetric_metric else 8 * M * M
    gradients = 32 * M if derived_gradient else 48 * M   # the corner's own gradient is minus the sum of the other two
    integrals = 16 * M * (M - 1) if derived_row else 16 * M * M   # corner 0's sums are minus the other corners'
    return gradients + integrals + metric + 48 * M + 16 * M + (4 * M + 8) + 8 + 24 * 2


def kernel_source_hash(files=("evp_step.hip", "sih_internal.h")):
    """Of the sources of the subcycle kernels; ir_source_hash() is the same for the incremental remap's.  The counter
    figures kept under profiles/ carry both, and a figure taken with other sources is not reported."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "mpas_seaice_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def ir_source_hash():
    return kernel_source_hash(("ir_advect.hip", "sih_internal.h"))


def committed_profile():
    """profiles/traffic_latest.json: HBM bytes per launch from the PMC counters and rocprofv3's average kernel
    durations, both from EARLIER profile runs of bench.py at QU60 (scripts/profile_round.sh; counters cannot be
    collected inside a timed run), or {} when the file is missing or unreadable."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as fh:
            return json.load(fh)
    except Exception:  # noqa: BLE001
        return {}


def hbm_only_pass(local_rank, nsub=24, steps=5):
    """The same subcycle on QU30 (655 362 cells, 1.4 GB touched per subcycle: nothing survives in the 256 MB
    Infinity Cache from one subcycle to the next), so that the line carries what the access pattern gets from
    HBM alone next to the QU60 figure."""
    from mpas_seaice_amd import cases, mesh as meshmod, velocity_solver as vs
    mesh = meshmod.make_named_mesh("QU30")
    dom = cases.make_domain(mesh, "QU30", configs={"config_elastic_subcycle_number": nsub})
    vs.seaice_mesh_pool_create(dom, device=local_rank)
    try:
        vs.seaice_init_velocity_solver(dom)
        ctx = dom.blocklist[0].ctx
        vs._upload(dom, dom.blocklist[0])
        vs.velocity_solver_pre_subcycle(dom)
        ctx.subcycle_timed(nsub, use_graph=False, sample_every=0)
        tot = cell = vert = 0.0
        for _ in range(steps):
            t, tc, tv = ctx.subcycle_timed(nsub, use_graph=False, sample_every=4)
            tot, cell, vert = tot + t, cell + tc, vert + tv
        M = int(mesh.maxEdges)
        cell /= steps
        b = b_alg_cell_kernel(M) * mesh.nCells
        return {"workload": f"QU30 ({mesh.nCells} cells), {nsub} subcycles x {steps}", "kernel_ms": cell,
                "vertex_kernel_ms": vert / steps, "achieved": b / (cell * 1e-3) / 1e9,
                "frac": b / (cell * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "updates_per_s": mesh.nCells * nsub * steps / (tot * 1e-3)}
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def cpu_baseline(mesh, name, max_seconds=25.0):
    """The CPU oracle (a restatement of the reference loops, OpenMP where the reference has it) timed
    on this host: whole-mesh subcycles until ~max_seconds of work."""
    from mpas_seaice_amd import cases
    from oracle import evp_flow as ora
    dom = cases.make_domain(mesh, name)
    ora.seaice_init_velocity_solver(dom)
    ora.velocity_solver_pre_subcycle(dom)
    nC = mesh.nCells
    t0 = time.perf_counter()
    ora.subcycle_velocity_solver(dom, 4)
    t1 = time.perf_counter() - t0
    n = int(max(4, min(4800, (max_seconds / 2.0) / max(t1 / 4.0, 1e-6))))  # about 12 s of work
    t0 = time.perf_counter()
    ora.subcycle_velocity_solver(dom, n)
    dt = time.perf_counter() - t0
    return {"value": nC * n / dt, "unit": "cell-subcycle-updates/s", "cores": ora.num_threads(), "kind": "port",
            "sample": f"{n} subcycles of the full {name} mesh ({nC} cells), OpenMP threads = cores"}


def advection_timing(dom, mesh, name, reps=7, with_cpu=True):
    """Secondary measurement: one incremental-remap advection step (7 tracers, 19 layer fields x 5
    categories = 95 scalars per cell) on the same mesh, tracers resident on the device, timed with HIP
    events inside the library; next to it the CPU oracle on the same mesh (it materialises the reference's
    per-quadrature-point arrays: 15 GB of host memory at QU60; meshes beyond 200 000 cells fall back to QU240)."""
    from mpas_seaice_amd import advection as adv
    adv.seaice_init_advection(dom)
    b = dom.blocklist[0]
    for k in ("uVelocity", "vVelocity"):  # the velocities the subcycle loop left on the device
        b.ctx.get(k, b.structs["velocity_solver"][k])
    adv.seaice_run_advection(dom, download=False)
    dt = float(b.structs["velocity_solver"]["dynamicsTimeStep"][0])
    tot, ph = [], []
    for _ in range(reps):
        t, p = b.ctx.ir_run_timed(dt)
        tot.append(t)
        ph.append(p)
    nCat = b.dimensions["nCategories"]
    ns = sum(t.nLayers for t in dom.solver_state["tracersHead"]) * nCat
    ms = float(np.median(tot))
    # Two accountings of the compulsory traffic, both against the same time.  (1) The nominal figure of the hot-path
    # scope: per advected scalar and cell 20 B (value read, mask, value written) + 40 B of reconstruction scratch, and
    # 300 B of geometry per cell.  (2) This layout's own: value read + written, centre / gradient / product written
    # and read back, three edge fluxes per cell written once and read twice = 152 B per scalar and cell.  (3) What the
    # counters saw, when the committed profile was taken with these kernel sources.
    nominal_bytes = mesh.nCells * (ns * (20 + 40) + 300)
    layout_bytes = mesh.nCells * ns * (8 + 8 + 4 * 8 * 2 + 3 * 8 * 3)
    prof = committed_profile()
    counted = None
    if name == "QU60" and prof.get("ir_source_hash") == ir_source_hash():
        counted = float(sum(v for k, v in prof.get("advection_hbm_bytes_per_step", {}).items() if "_init_" not in k))

    def rate(b):
        return {"bytes_per_step": b, "GBps": b / (ms * 1e-3) / 1e9, "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    out = {"workload": f"incremental remap, {ns} scalars per cell on {name}", "ms_per_step": ms,
           "tracer_cell_updates_per_s": mesh.nCells * ns / (ms * 1e-3),
           "phase_ms": dict(zip(["convert+masks", "reconstruct", "departure_triangles", "fluxes", "update+zap"],
                                [round(float(x), 4) for x in np.median(np.array(ph), axis=0)])),
           "traffic": {"nominal_60B_per_scalar_plus_300B_per_cell": rate(nominal_bytes),
                       "layout_152B_per_scalar": rate(layout_bytes),
                       "counted_in_an_earlier_profile": rate(counted) if counted else None},
           "nominal_GBps": layout_bytes / (ms * 1e-3) / 1e9}
    a = np.zeros_like(b.structs["tracers"]["iceAreaCategory"])
    b.ctx.get("iceAreaCategory", a)
    if not np.isfinite(a).all():
        raise SystemExit("non-finite ice area after the advection loop")
    if with_cpu:
        from mpas_seaice_amd import cases, mesh as meshmod
        from oracle import evp_flow as ora, ir_flow as ir
        if mesh.nCells <= 200000:
            small, m2 = name, mesh
        else:
            small = "QU240"
            m2 = meshmod.make_named_mesh(small)
        d2 = cases.make_domain(m2, small, configs={"config_elastic_subcycle_number": 4})
        ora.seaice_init_velocity_solver(d2)
        ir.seaice_init_advection(d2)
        ora.seaice_run_velocity_solver(d2)
        t0 = time.perf_counter()
        nrep = 5 if m2.nCells < 50000 else 3
        for _ in range(nrep):
            ir.seaice_run_advection(d2)
        dtc = (time.perf_counter() - t0) / nrep
        out["cpu_baseline"] = {"value": m2.nCells * ns / dtc, "unit": "tracer-cell-updates/s", "cores": ora.num_threads(),
                               "kind": "port", "sample": f"{nrep} advection steps on {small} ({m2.nCells} cells)"}
    return out


def _free_port():
    import socket
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, extra_env=None, limit_s=None):
    """`python bench.py --gpus N` without a launcher: starts N fresh rank processes of this file (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run sets them) and waits for them.  The parent never
    imports torch and never touches a GPU.  Rank 0's standard output (the one JSON line) is passed through as it is;
    the other ranks' standard output goes to standard error.  Returns (exit code, lines rank 0 printed, codes of all
    ranks).  A rank that ends with an error ends the others: they are given 20 s to leave by themselves (every wait of
    the library is bounded and every collective has a time limit), then terminated by PID."""
    import subprocess
    import threading
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(_free_port()), "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                "SIH_BENCH_SELF_LAUNCHED": "1"})
    env.update(extra_env or {})
    if limit_s is None:
        limit_s = float(os.environ.get("SIH_BENCH_RANK_LIMIT_S", "1100"))
    procs, lines0 = [], []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1))

    def relay(p, r):
        for line in p.stdout:
            if r == 0:
                lines0.append(line)
                sys.stdout.write(line)
                sys.stdout.flush()
            else:
                sys.stderr.write(f"[rank {r}] {line}")
    threads = [threading.Thread(target=relay, args=(p, r), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    t0, first_bad = time.time(), None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        codes = [p.poll() for p in procs]
        if first_bad is None and any(c not in (None, 0) for c in codes):
            first_bad = time.time()
        over = time.time() - t0 > limit_s
        if over or (first_bad is not None and time.time() - first_bad > 20.0):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            time.sleep(5.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            if over:
                sys.stderr.write(f"bench.py: ranks still running after {limit_s:.0f} s were ended\n")
    for t in threads:
        t.join(timeout=5.0)
    codes = [p.wait() for p in procs]
    rc = 0
    for c in codes:
        if c != 0:
            rc = c if c > 0 else 128 - c
            break
    return rc, lines0, codes


def self_launch(args, argv):
    """The parent of a plain `python bench.py --gpus N`.  One retry, and only after a clean failure: if a rank raised a
    Python error (exit code 1: a bounded wait of the library's own transport ran out, a check of its results failed)
    and no rank was ended by a signal or an abort, the run is started once more with the conservative settings --
    halo messages as torch.distributed all_to_all_single over RCCL, no persistent kernel -- and the line says so."""
    rc, lines, codes = launch_ranks(args.gpus, argv)
    if rc == 0 or os.environ.get("SIH_BENCH_NO_RETRY", "0") != "0":
        return rc
    clean = all(c in (0, 1) or c == -15 for c in codes) and any(c == 1 for c in codes)
    if not clean or os.environ.get("SIH_TRANSPORT", "p2p") == "torch" or any(l.lstrip().startswith("{") for l in lines):
        return rc
    sys.stderr.write(f"bench.py: the ranks ended with codes {codes}; once more with SIH_TRANSPORT=torch SIH_PERSISTENT=0\n")
    rc2, _, _ = launch_ranks(args.gpus, argv, extra_env={
        "SIH_TRANSPORT": "torch", "SIH_PERSISTENT": "0",
        "SIH_BENCH_RETRY_NOTE": f"second launch: the first one (the library's own transport) ended with exit codes {codes}"})
    return rc2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mesh", default="QU60", help="QU480/QU240/QU120/QU60/QU30/QU15, var<n> (variable resolution, "
                    "maxEdges 7, 10 n^2 + 2 cells), or an MPAS grid file (*.nc)")
    ap.add_argument("--partition-file", default=None, help="METIS graph.info.part.N for --gpus N (default: "
                    "cut along a space-filling curve)")
    ap.add_argument("--subcycles", type=int, default=120)
    ap.add_argument("--graph", type=int, default=int(os.environ.get("SIH_BENCH_GRAPH", "2")),
                    help="how one GPU queues the subcycles of a step: 2 (default) the first iteration as two launches "
                         "between HIP events (the roofline's kernel times, inside the timed region), the others replayed "
                         "from a HIP graph as the library's own step (sih_evp_run) does; 1 all of them from the graph "
                         "(kernel times sampled after the timed region); 0 all as separate launches, every "
                         "--sample-every-th between events")
    ap.add_argument("--sample-every", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-renumber", action="store_true")
    ap.add_argument("--no-advection", action="store_true", help="skip the incremental-remap timing")
    ap.add_argument("--no-hbm-only", action="store_true", help="skip the QU30 pass (HBM-only rate of the cell kernel)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N`: this process only starts the ranks and passes on rank 0's line
        sys.exit(self_launch(args, sys.argv[1:]))

    import torch
    from mpas_seaice_amd import cases, mesh as meshmod, velocity_solver as vs

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the solver has no CPU path")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    n_dev = max(torch.cuda.device_count(), 1)
    if world > n_dev:
        # more ranks than GPUs (a rehearsal on a box with one GPU): the ranks share devices, which RCCL refuses, so
        # the rendezvous and the scalar reductions go over gloo; the halo messages still travel inside the library
        os.environ.setdefault("SIH_DIST_BACKEND", "gloo")
    cell_owner = None
    mesh_label = os.path.basename(args.mesh)
    if args.mesh.endswith(".nc"):
        from mpas_seaice_amd import mesh_io
        mesh = mesh_io.read_mpas_mesh(args.mesh, earth_radius=6371229.0)
        args.mesh = None  # idealised state from latitude / longitude only
        if args.partition_file:
            cell_owner = mesh_io.read_graph_partition(args.partition_file, mesh.nCells)
    else:
        mesh = meshmod.make_named_mesh(args.mesh)
    nsub = args.subcycles
    M = int(mesh.maxEdges)
    advection = None

    sym_metric = os.environ.get("SIH_METRIC_INTEGRALS") != "as_computed"
    derived_grad = os.environ.get("SIH_BASIS_GRADIENTS") != "as_computed"
    derived_row = os.environ.get("SIH_BASIS_INTEGRALS") != "as_computed"
    if world == 1:
        dom = cases.make_domain(mesh, args.mesh, configs={"config_elastic_subcycle_number": nsub})
        vs.seaice_mesh_pool_create(dom, device=local_rank, renumber=not args.no_renumber)
        vs.seaice_init_velocity_solver(dom)
        ctx = dom.blocklist[0].ctx
        for b in dom.blocklist:
            vs._upload(dom, b)
        vs.velocity_solver_pre_subcycle(dom)  # first call: code objects are loaded on first launch
        for b in dom.blocklist:
            vs._upload(dom, b)  # same inputs again, so that the timed call starts from the same state
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        vs.velocity_solver_pre_subcycle(dom)
        ctx.sync()
        pre_ms = (time.perf_counter() - t0) * 1e3
        # What the library's own step (sih_evp_run) runs the iterations as: two kernels an iteration on a mesh of this
        # benchmark's size; on meshes small enough, the persistent or the fused kernel (then that is what is timed)
        ctx.subcycle_graph(nsub)
        ctx.sync()
        single_form = int(ctx.info("iterationForm"))
        sym_metric = bool(ctx.info("symmetricMetric"))
        derived_grad = bool(ctx.info("derivedGradient"))
        derived_row = bool(ctx.info("derivedIntegralRow"))
        cell_ms, vert_ms, nsamp = 0.0, 0.0, 0
        if single_form != 0:
            for _ in range(args.warmup):
                ctx.subcycle_graph(nsub)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ctx.subcycle_graph(nsub)
            ctx.sync()
            wall = time.perf_counter() - t0
        else:
            def one_step(sample_every):
                if args.graph == 2:
                    # the first iteration as two launches between HIP events, the other nsub - 1 replayed from a graph
                    _, tc, tv = ctx.subcycle_timed(1, use_graph=False, sample_every=1 if sample_every else 0)
                    ctx.subcycle_graph(nsub - 1)
                    return tc, tv
                _, tc, tv = ctx.subcycle_timed(nsub, use_graph=args.graph, sample_every=sample_every)
                return tc, tv
            for _ in range(args.warmup):
                one_step(0)
            torch.cuda.synchronize()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                tc, tv = one_step(args.sample_every)
                if tc > 0:
                    cell_ms += tc
                    vert_ms += tv
                    nsamp += 1
            torch.cuda.synchronize()
            ctx.sync()
            wall = time.perf_counter() - t0
        t0 = time.perf_counter()
        ctx.subcycle(1, last=True)
        vs.velocity_solver_post_subcycle(dom)
        ctx.sync()
        post_ms = (time.perf_counter() - t0) * 1e3
        if nsamp == 0:  # graph replay has no per-kernel events: sample the kernels once afterwards
            _, cell_ms, vert_ms = ctx.subcycle_timed(nsub, use_graph=False, sample_every=4)
            nsamp = 1
        cell_ms /= nsamp
        vert_ms /= nsamp
        u = np.zeros(mesh.nVertices + 1)
        ctx.get("uVelocity", u)
        if not np.isfinite(u).all():
            raise SystemExit("non-finite velocities after the benchmark loop")
        n_cells_total = mesh.nCells
        if not args.no_advection:
            advection = advection_timing(dom, mesh, args.mesh, with_cpu=not args.no_cpu_baseline)
        vs.seaice_mesh_pool_destroy(dom)
        parallelism = "1 block"
    else:
        single_form = 0
        from mpas_seaice_amd import parallel
        res = parallel.bench_multi_gpu(mesh, args.mesh, nsub, args.steps, args.warmup, local_rank,
                                       renumber=not args.no_renumber, cell_owner=cell_owner)
        wall, pre_ms, post_ms, cell_ms, vert_ms, n_cells_total = (res[k] for k in (
            "wall", "pre_ms", "post_ms", "cell_ms", "vert_ms", "n_cells_total"))
        parallelism = (f"{world} MPAS blocks, one per GPU, {res['interval']} halo layers, uVelocity/vVelocity/stress "
                       f"exchanged every {res['interval']} subcycles: {res['transport_text']}"
                       + (", the push on a second stream beside interior work" if res.get("overlap") else "")
                       + (f" [{res['transport_note']}]" if res.get("transport_note") else ""))
        block_cells = res["n_block_cells"]
        args.graph = int(res.get("graph", False))  # the chunks between two exchanges are replayed from a graph
        if rank != 0:
            return

    ms_per_step = wall / args.steps * 1e3
    value = n_cells_total * nsub * args.steps / wall
    b_cell, b_vert = b_alg_per_cell(M)
    cells_per_launch = n_cells_total if world == 1 else block_cells
    roof_bytes = b_alg_cell_kernel(M) * cells_per_launch
    achieved = roof_bytes / (cell_ms * 1e-3) / 1e9 if cell_ms > 0 else 0.0
    # HBM bytes per launch from the PMC counters: they are collected in passes of their own (rocprofv3 --pmc,
    # scripts/refresh_traffic.py) and kept under profiles/ with a hash of the kernel source they were taken with;
    # a figure taken with other kernels is not reported
    traffic, traffic_note, rocprof_ms = None, None, None
    tj = committed_profile()
    if world == 1 and mesh_label == "QU60" and tj:  # the counters were collected on this workload
        if tj.get("kernel_source_hash") == kernel_source_hash():
            traffic = tj.get("cell_kernel_hbm_bytes_per_launch")
            traffic_note = ("NOT measured in this run: profiles/traffic_latest.json, an earlier profile of the same "
                            "kernel sources; " + str(tj.get("source")))
            if tj.get("cell_kernel_avg_ns_rocprof"):
                rocprof_ms = {"cell_kernel": tj["cell_kernel_avg_ns_rocprof"] * 1e-6,
                              "vertex_kernel": (tj.get("vertex_kernel_avg_ns_rocprof") or 0.0) * 1e-6,
                              "from": tj.get("kernel_stats_file")}
        else:
            traffic_note = "profiles/traffic_latest.json was taken with other kernel sources: not reported"
    layout_bytes = b_layout_cell_kernel(M, sym_metric, derived_grad, derived_row) * cells_per_launch
    # a mesh with cells of several degrees: the same accounting with every cell's own number of corners in place of
    # maxEdges (the planes are maxEdges deep, but a corner a cell does not have is never read)
    degree_note = None
    if world == 1:
        ne = np.asarray(mesh.nEdgesOnCell[:mesh.nCells], dtype=np.int64)
        if ne.min() != ne.max():
            by_degree = float(sum(b_alg_cell_kernel(int(e)) * int(c) for e, c in zip(*np.unique(ne, return_counts=True))))
            degree_note = {"cells_by_degree": {int(e): int(c) for e, c in zip(*np.unique(ne, return_counts=True))},
                           "algorithmic_bytes_per_launch_by_degree": by_degree,
                           "frac_by_degree": (by_degree / (cell_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if cell_ms > 0 else 0.0}
    out = {
        "metric": "EVP cell-subcycle-updates/sec at " + mesh_label, "value": value, "unit": "cell-subcycle-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{mesh_label} full-sphere mesh ({n_cells_total} cells, maxEdges {M}), "
                               f"{nsub} EVP subcycles per step, variational Wachspress operators, idealised forcing",
                   "parallelism": parallelism, "hip_graph": bool(args.graph),
                   "launches": ({2: "per step: one iteration as two launches between HIP events, the others replayed from a HIP graph",
                                 1: "per step: all iterations replayed from a HIP graph",
                                 0: "two launches per iteration"}.get(int(args.graph)) if world == 1 else None)},
        "roofline": {"bound": "hbm", "kernel": "k_sub_cells_split (fused strain + EVP stress + stress-divergence partial sums, one thread per cell corner)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "algorithmic_bytes_per_launch": roof_bytes,
                     "kernel_ms": cell_ms, "vertex_kernel_ms": vert_ms,
                     # rocprofv3's average durations of the same kernels in the committed kernel trace (an earlier
                     # run): frac can be recomputed from algorithmic_bytes_per_launch and either duration
                     "kernel_ms_rocprof": rocprof_ms,
                     # the same kernel time against the bytes this layout really has to move (compact gradients)
                     # and, when the counters are current, against the bytes they counted
                     "layout_bytes_per_launch": layout_bytes,
                     # what the layout leaves out of the algorithmic bytes: the compact gradients always; one side of the
                     # mirrored metric integrals, the corner's own gradient and the first corner's rows of
                     # basisIntegralsU / V unless switched off in the environment.  The algorithmic count is that of
                     # the dense formulation: with these reductions the kernel moves fewer bytes than it names, so
                     # `frac` can pass 1 while frac_layout and frac_counter cannot
                     "layout": {"compact_gradients": True, "symmetric_metric_integrals": bool(sym_metric),
                                "derived_corner_gradient": bool(derived_grad),
                                "derived_first_corner_sums": bool(derived_row)},
                     "frac_note": "achieved / frac price the kernel's time with the algorithmic bytes of the dense "
                                  "formulation; this layout moves fewer (layout_bytes_per_launch), so frac can pass 1: "
                                  "frac_layout and frac_counter are the fractions of peak by bytes really moved",
                     "frac_layout": (layout_bytes / (cell_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if cell_ms > 0 else 0.0,
                     "frac_counter": (traffic / (cell_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and cell_ms > 0 else None,
                     "traffic_source": traffic_note},
        "subcycle_hbm": {"algorithmic_bytes_per_cell_subcycle": b_cell + b_vert,
                         "achieved_GBps": (b_cell + b_vert) * value / 1e9,
                         "frac_of_peak": (b_cell + b_vert) * value / 1e9 / HBM_PEAK_GBS},
        "pre_ms": pre_ms, "post_ms": post_ms,
    }
    if degree_note:
        out["roofline"]["degrees"] = degree_note
    if world == 1 and single_form != 0:
        it_ms = ms_per_step / nsub
        eq = (b_cell + b_vert) * n_cells_total / (it_ms * 1e-3) / 1e9
        out["roofline"].update({
            "kernel": ("k_sub_persistent (all iterations of a step in one launch: tiles resident on the chip, per-corner "
                       "sums handed from tile to tile)" if single_form == 1 else
                       "k_sub_fused (one launch per iteration: velocity solve of a tile's vertices, then strain, EVP stress "
                       "and stress-divergence partial sums of its 64 cells)"),
            "bound": "hbm", "achieved": eq, "frac": eq / HBM_PEAK_GBS, "kernel_ms": it_ms,
            "two_kernel_form_cell_kernel_ms": cell_ms, "two_kernel_form_vertex_kernel_ms": vert_ms, "vertex_kernel_ms": None,
            "algorithmic_bytes_per_launch": (b_cell + b_vert) * n_cells_total, "layout_bytes_per_launch": None,
            "frac_layout": None, "traffic": None, "frac_counter": None,
            "traffic_source": "equivalent rate: the bytes of one iteration of the two-kernel form (cells and vertices) over "
                              "the time an iteration took" + (" (its tables never leave the chip)" if single_form == 1 else "")})
    if world > 1:
        out["world"] = {"ranks_at_rendezvous": res["world"], "backend": res["rendezvous_backend"],
                        "devices_visible_per_rank": n_dev, "ranks": res["ranks"],
                        "launcher": "bench.py itself" if os.environ.get("SIH_BENCH_SELF_LAUNCHED") else "external",
                        "note": os.environ.get("SIH_BENCH_RETRY_NOTE") or None}
        out["exchange"] = {"transport": res["transport"], "rendezvous_backend": res["rendezvous_backend"],
                           "us_per_exchange_in_a_chain_of_50": res["exchange_us"], "block_cells_with_halo": block_cells,
                           "inside_the_subcycle_kernel": res["exchange_inside_kernel"],
                           "inside_note": res["exchange_inside_note"] or None}
        if res["persistent_kernel"]:
            # the block is small enough to stay on the chip: one launch runs all iterations of a step, the two-kernel
            # timings above are not what ran.  Its tables never leave registers and LDS, so the figure below (the
            # bytes the two-kernel form would have to move per iteration over the time an iteration took) is an
            # equivalent rate, not HBM traffic.
            it_ms = ms_per_step / nsub
            eq = (b_cell + b_vert) * block_cells / (it_ms * 1e-3) / 1e9
            out["roofline"].update({
                "kernel": "k_sub_persistent (all iterations of a step in one launch: tiles of 64 cells resident on the "
                          "chip, per-corner sums handed from tile to tile, halo values"
                          + (" exchanged inside the launch)" if res["exchange_inside_kernel"] else " exchanged between launches)"),
                "bound": "hbm", "achieved": eq, "frac": eq / HBM_PEAK_GBS, "kernel_ms": it_ms, "vertex_kernel_ms": None,
                "algorithmic_bytes_per_launch": (b_cell + b_vert) * block_cells, "layout_bytes_per_launch": None,
                "frac_layout": None, "traffic": None, "frac_counter": None,
                "traffic_source": "equivalent rate of a chip-resident kernel: bytes of the two-kernel form per iteration "
                                  "over the time an iteration took (step time / iterations, exchanges included)"})
        elif res.get("fused_kernel"):
            # a block of this size runs one launch per iteration over tiles of 64 cells (the vertices of one iteration,
            # then the cells of the next): the two-kernel timings above are not what ran
            it_ms = ms_per_step / nsub
            eq = (b_cell + b_vert) * block_cells / (it_ms * 1e-3) / 1e9
            out["roofline"].update({
                "kernel": "k_sub_fused (one launch per iteration: velocity solve of the tile's vertices, then strain, EVP "
                          "stress and stress-divergence partial sums of its 64 cells; halo values exchanged between launches)",
                "bound": "hbm", "achieved": eq, "frac": eq / HBM_PEAK_GBS, "kernel_ms": it_ms, "vertex_kernel_ms": None,
                "algorithmic_bytes_per_launch": (b_cell + b_vert) * block_cells, "layout_bytes_per_launch": None,
                "frac_layout": None, "traffic": None, "frac_counter": None,
                "traffic_source": "bytes of one iteration (cells and vertices) over the time an iteration took (step "
                                  "time / iterations, exchanges included)"})
    if world == 1 and not args.no_hbm_only and mesh_label == "QU60":
        out["roofline"]["hbm_only"] = hbm_only_pass(local_rank)
    if world == 1 and advection is not None:
        out["advection"] = advection
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(mesh, args.mesh)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

"""Several MPAS blocks driven through the HIP library: halo pack/unpack and the staged pre/post-cycle."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from mpas_seaice_amd import cases, parallel, velocity_solver as vs
from util import get_mesh

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-10


@pytest.mark.parametrize("nblocks,interval", [(2, 1), (4, 1), (3, 3), (4, 4)])
def test_blocks_on_one_gpu_match_single_block_oracle(oracle, nblocks, interval):
    """interval k > 1: k halo layers, redundant halo updates, one exchange per k subcycles."""
    mesh = get_mesh("QU480")
    cfg = {"config_elastic_subcycle_number": 40}
    ref = cases.make_domain(mesh, "QU480", configs=cfg, open_water_band=(-0.2, 0.3))
    oracle.seaice_init_velocity_solver(ref)
    oracle.seaice_run_velocity_solver(ref)
    pcfg = dict(cfg, config_num_halos=max(2, interval), config_halo_exchange_interval=interval)
    dom, halos, _ = parallel.make_partitioned_domain(mesh, "QU480", nblocks, configs=pcfg,
                                                     open_water_band=(-0.2, 0.3))
    vs.seaice_mesh_pool_create(dom, device=0)
    try:
        dom.comm = parallel.LocalComm(dom, halos, nblocks)
        vs.seaice_init_velocity_solver(dom)
        vs.seaice_run_velocity_solver(dom)
    finally:
        vs.seaice_mesh_pool_destroy(dom)
    for pool, name, loc, n in (("velocity_solver", "uVelocity", "vertex", mesh.nVertices),
                               ("velocity_solver", "vVelocity", "vertex", mesh.nVertices),
                               ("velocity_variational", "stress11", "cell", mesh.nCells),
                               ("velocity_variational", "stress12", "cell", mesh.nCells),
                               ("velocity_solver", "divergence", "cell", mesh.nCells),
                               ("velocity_solver", "oceanStressCellU", "cell", mesh.nCells),
                               ("velocity_solver", "solveVelocity", "vertex", mesh.nVertices)):
        g = parallel.gather_global(dom.blocklist, halos, pool, name, n, loc).astype(np.float64)
        r = ref.blocklist[0].structs[pool][name][..., :n].astype(np.float64)
        err = np.abs(g - r).max() / max(np.abs(r).max(), 1e-300)
        assert err <= TOL, (name, err)


@pytest.mark.parametrize("extra", [[], ["overlap"], ["upwind"], ["p2p"], ["p2p", "overlap"], ["p2p", "upwind"],
                                   ["p2p", "graph"], ["p2p", "graph", "overlap"], ["p2p", "graph", "deep"],
                                   ["p2p", "graph", "deep", "overlap"]])
def test_two_ranks_sharing_the_gpu_over_gloo(extra):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), "gpu", "geo8"] + extra
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count(": ok") == 2, out.stdout
    if "deep" in extra and "p2p" in extra:
        # small blocks, each alone in its process: the subcycle loop is one launch with the exchanges inside it
        assert out.stdout.count("launches 2 with exchange 1") == 2, out.stdout


@pytest.mark.parametrize("mesh,ranks,extra", [("QU240", 2, ["p2p", "deep"]), ("QU480", 4, ["p2p", "deep"]),
                                              ("var16", 3, ["p2p", "graph", "deep"])])
def test_exchange_inside_the_persistent_kernel(mesh, ranks, extra):
    """More tiles a block and more ranks than the two-rank cases above: every rank's block runs its subcycle loop
    as one launch that sends and receives the halo values itself (three halo layers, an exchange every three
    iterations and after the eleventh), checked on every rank against the single-block oracle."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), "gpu", mesh] + extra
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count(": ok") == ranks, out.stdout
    assert out.stdout.count("launches 2 with exchange 1") == ranks, out.stdout


def _run_ranks(mesh, ranks, extra, env_extra, timeout=400):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), "gpu", mesh] + extra
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.parametrize("extra", [["p2p", "deep"], ["p2p"], ["p2p", "graph", "deep"], ["p2p", "reinit"], ["p2p", "deep", "reinit"]])
def test_exchange_inside_the_kernel_changes_no_bit(tmp_path, extra):
    """the same three-rank run with the exchanges between the launches and inside the one launch (velocities and
    stresses every third iteration with three halo layers; velocities after every iteration, the reference's
    order; the chunks between two exchanges replayed from HIP graphs that hold the persistent kernel): every array
    of every block, halo entries included, must be the same bits"""
    res = []
    for inside in ("0", "1"):
        d = tmp_path / inside
        d.mkdir()
        out = _run_ranks("QU240", 3, extra, {"SIH_PERSIST_EXCHANGE": inside, "SIH_TEST_DUMP": str(d)})
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
        assert out.stdout.count(f"with exchange {inside}") == 3, out.stdout
        res.append([np.load(d / f"rank{r}.npz") for r in range(3)])
    for a, b in zip(*res):
        assert np.abs(a["u"]).max() > 0
        for k in a.files:
            assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("mesh,ranks,extra", [("QU240", 2, ["p2p", "deep"]), ("QU240", 3, ["p2p"]), ("QU480", 4, ["p2p", "deep"]),
                                              ("QU240", 3, ["p2p", "long"])])
def test_exchange_inside_the_persistent_kernel_with_small_tiles(tmp_path, mesh, ranks, extra):
    """The variant of the persistent kernel for blocks that tiles of 64 cells cannot hold (tiles of 42 cells, the
    rows of the metric integrals streamed), forced here on small blocks by declaring room for a single tile of 64:
    the exchange inside it against the exchange between the launches of the 64-cell form, every array of every
    block, halo entries included, the same bits; and against the single-block oracle on every rank."""
    res = []
    for small in (False, True):
        d = tmp_path / str(int(small))
        d.mkdir()
        env = {"SIH_TEST_DUMP": str(d), "SIH_PERSIST_EXCHANGE": "1" if small else "0"}
        if small:
            env.update(SIH_PERSIST_CAPACITY="1", SIH_PERSIST_CAPACITY_SMALL="1024")
        out = _run_ranks(mesh, ranks, extra, env)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
        assert out.stdout.count(": ok") == ranks, out.stdout
        assert out.stdout.count(f"tile cells {42 if small else 64}") == ranks, out.stdout
        # 39 iterations with an exchange after each are three launches: a launch holds at most 16 exchanges
        assert out.stdout.count(f"with exchange {int(small) * (3 if 'long' in extra else 1)} ") == ranks, out.stdout
        res.append([np.load(d / f"rank{r}.npz") for r in range(ranks)])
    for a, b in zip(*res):
        assert np.abs(a["u"]).max() > 0
        for k in a.files:
            assert np.array_equal(a[k], b[k]), k


def test_a_silent_neighbour_ends_the_kernel_with_the_exchange_inside():
    """Rank 1 agrees to exchange inside the kernel and then never launches: rank 0's launch must not hang.  Its
    waits are bounded (here 300 ms, and as much for the exchanges between launches that precede it in the step), every
    workgroup leaves, and the host gets an error that says why."""
    out = _run_ranks("geo8", 2, ["p2p", "deep", "silent"], {"SIH_PERSIST_TIMEOUT_MS": "300", "SIH_HALO_TIMEOUT_MS": "300"},
                     timeout=120)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "rank 0 reported:" in out.stdout and "did not arrive in time" in out.stdout, out.stdout


def test_a_launch_that_gave_up_is_reported_by_the_transport_status_alone():
    """The multi-rank shim asks only the transport's status after the subcycle loop (sih_comm_status): a persistent
    launch whose neighbour never sent its halo values must surface there, without any other call that reads the
    kernel's status word."""
    out = _run_ranks("geo8", 2, ["p2p", "deep", "statusonly"], {"SIH_PERSIST_TIMEOUT_MS": "300", "SIH_HALO_TIMEOUT_MS": "300"},
                     timeout=120)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "rank 0 status call reported:" in out.stdout and "persistent subcycle kernel" in out.stdout, out.stdout


@pytest.mark.parametrize("nblocks,interval,graph", [(2, 1, False), (4, 1, True), (3, 3, False), (4, 4, True)])
def test_overlapped_exchange_is_bit_identical(nblocks, interval, graph):
    """The exchange hidden behind interior work (send vertices first, interior cells of the next iteration
    while the messages travel, rim cells after the unpack) must not change a bit."""
    mesh = get_mesh("QU480")
    res = []
    for overlap in (False, True):
        pcfg = {"config_elastic_subcycle_number": 25, "config_num_halos": max(2, interval),
                "config_halo_exchange_interval": interval, "config_overlap_halo_exchange": overlap,
                "config_subcycle_graph": graph}
        dom, halos, _ = parallel.make_partitioned_domain(mesh, "QU480", nblocks, configs=pcfg,
                                                         open_water_band=(-0.2, 0.3))
        vs.seaice_mesh_pool_create(dom, device=0)
        try:
            dom.comm = parallel.LocalComm(dom, halos, nblocks)
            vs.seaice_init_velocity_solver(dom)
            vs.seaice_run_velocity_solver(dom)
        finally:
            vs.seaice_mesh_pool_destroy(dom)
        res.append({(pool, name): parallel.gather_global(dom.blocklist, halos, pool, name, n, loc)
                    for pool, name, loc, n in (("velocity_solver", "uVelocity", "vertex", mesh.nVertices),
                                               ("velocity_solver", "vVelocity", "vertex", mesh.nVertices),
                                               ("velocity_variational", "stress11", "cell", mesh.nCells),
                                               ("velocity_variational", "stress22", "cell", mesh.nCells),
                                               ("velocity_variational", "stress12", "cell", mesh.nCells),
                                               ("velocity_solver", "shear", "cell", mesh.nCells))})
    assert np.abs(res[0][("velocity_solver", "uVelocity")]).max() > 0
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k


def test_subcycle_parts_error_paths():
    from mpas_seaice_amd import lib as sihlib
    dom = cases.make_domain(get_mesh("QU480"), "QU480")
    vs.seaice_mesh_pool_create(dom)
    try:
        vs.seaice_init_velocity_solver(dom)
        ctx = dom.blocklist[0].ctx
        ctx.subcycle_parts([0, 3])  # whole-block pieces need no subsets
        with pytest.raises(sihlib.SeaiceHipError):
            ctx.subcycle_parts([1])  # subsets not set up
        with pytest.raises(sihlib.SeaiceHipError):
            ctx.subcycle_parts([9])
        ctx.overlap_setup()  # no halo lists: everything is interior / rest
        ctx.subcycle_parts([1, 2, 4, 5], True)
        ctx.sync()
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_weak_scheme_on_three_blocks(oracle):
    mesh = get_mesh("QU480")
    cfg = {"config_elastic_subcycle_number": 30, "config_strain_scheme": "weak",
           "config_stress_divergence_scheme": "weak"}
    ref = cases.make_domain(mesh, "QU480", configs=cfg, open_water_band=(-0.2, 0.3))
    oracle.seaice_init_velocity_solver(ref)
    oracle.seaice_run_velocity_solver(ref)
    dom, halos, _ = parallel.make_partitioned_domain(mesh, "QU480", 3, configs=cfg, open_water_band=(-0.2, 0.3))
    vs.seaice_mesh_pool_create(dom, device=0)
    try:
        dom.comm = parallel.LocalComm(dom, halos, 3)
        vs.seaice_init_velocity_solver(dom)
        vs.seaice_run_velocity_solver(dom)
    finally:
        vs.seaice_mesh_pool_destroy(dom)
    for pool, name, loc, n in (("velocity_solver", "uVelocity", "vertex", mesh.nVertices),
                               ("velocity_solver", "vVelocity", "vertex", mesh.nVertices),
                               ("velocity_weak", "stress11", "cell", mesh.nCells),
                               ("velocity_weak", "stress12", "cell", mesh.nCells),
                               ("velocity_solver", "shear", "cell", mesh.nCells)):
        g = parallel.gather_global(dom.blocklist, halos, pool, name, n, loc).astype(np.float64)
        r = ref.blocklist[0].structs[pool][name][..., :n].astype(np.float64)
        err = np.abs(g - r).max() / max(np.abs(r).max(), 1e-300)
        assert err <= TOL, (name, err)


def test_strain_averaging_on_three_blocks_follows_the_reference_per_block_loop(oracle):
    """seaice_average_strains_on_vertex visits a block's owned vertices only (variational.F:715), so corners at
    halo vertices keep their own strain and a partitioned run differs slightly from the one-block run: the
    device has to agree with the oracle run on the same partition, block by block."""
    mesh = get_mesh("QU480")
    cfg = {"config_elastic_subcycle_number": 30, "config_average_variational_strain": True}
    ref, rhalos, _ = parallel.make_partitioned_domain(mesh, "QU480", 3, configs=cfg, open_water_band=(-0.2, 0.3))
    rcomm = parallel.LocalComm(ref, rhalos, 3)
    ref.halo = parallel.oracle_halo_callback(rcomm)
    ref.allreduce_min = rcomm.allreduce_min
    oracle.seaice_init_velocity_solver(ref)
    oracle.seaice_run_velocity_solver(ref)
    dom, halos, _ = parallel.make_partitioned_domain(mesh, "QU480", 3, configs=cfg, open_water_band=(-0.2, 0.3))
    vs.seaice_mesh_pool_create(dom, device=0)
    try:
        dom.comm = parallel.LocalComm(dom, halos, 3)
        vs.seaice_init_velocity_solver(dom)
        vs.seaice_run_velocity_solver(dom)
    finally:
        vs.seaice_mesh_pool_destroy(dom)
    for pool, name, loc, n in (("velocity_solver", "uVelocity", "vertex", mesh.nVertices),
                               ("velocity_solver", "vVelocity", "vertex", mesh.nVertices),
                               ("velocity_variational", "stress11", "cell", mesh.nCells),
                               ("velocity_variational", "strain12", "cell", mesh.nCells),
                               ("velocity_solver", "divergence", "cell", mesh.nCells)):
        g = parallel.gather_global(dom.blocklist, halos, pool, name, n, loc).astype(np.float64)
        r = parallel.gather_global(ref.blocklist, rhalos, pool, name, n, loc).astype(np.float64)
        err = np.abs(g - r).max() / max(np.abs(r).max(), 1e-300)
        assert err <= TOL, (name, err)


def test_special_velocity_boundaries_on_three_blocks(oracle):
    """seaice_set_special_boundaries_velocity follows the halo update of every iteration
    (velocity_solver.F:2449-2456): on several blocks the library's application inside the iteration is switched
    off and the host calls it after its exchange.  Sources are neighbouring vertices, so every vertex that
    matters finds its source in its block; the result is the single-block oracle's."""
    mesh = get_mesh("hex_24_20")
    nV, nC = mesh.nVertices, mesh.nCells
    cov = mesh["cellsOnVertex"]
    rim = (cov > nC).any(axis=0)
    voe = mesh["verticesOnEdge"] - 1
    typ_g = np.zeros(nV, dtype=np.int32)
    src_g = np.zeros(nV, dtype=np.int32)
    typ_g[rim] = 3
    rng = np.random.default_rng(3)
    inner = np.flatnonzero(~rim)
    picked = set()
    for k, v in enumerate(rng.choice(inner, size=50, replace=False)):
        nbs = [int(b if a == v else a) for a, b in voe.T if a == v or b == v]
        nbs = [w for w in nbs if not rim[w] and w not in picked and typ_g[w] == 0]
        if not nbs or any(src_g[w] == v + 1 for w in range(nV) if typ_g[w] in (1, 2)):
            continue
        typ_g[v] = 1 if k % 2 == 0 else 2
        src_g[v] = nbs[0] + 1
        picked.add(v)
        picked.add(nbs[0])        # a source is never marked itself: no chains, whose order would be per block
    assert (typ_g == 1).sum() > 5 and (typ_g == 2).sum() > 5
    mv_g = np.where(rim, 0, 1).astype(np.int32)
    mc_g = np.ones(nC, dtype=np.int32)
    mc_g[::17] = 0
    cfg = {"config_elastic_subcycle_number": 20, "config_dt": 600.0, "config_use_special_boundaries_velocity": True,
           "config_use_special_boundaries_velocity_masks": True}

    def mark(b, gv, gc):
        sb = b.structs["special_boundaries"]
        n_v, n_c = b.dimensions["nVertices"], b.dimensions["nCells"]
        sb["vertexBoundaryType"][:n_v] = typ_g[gv]
        sb["vertexBoundarySource"][:n_v] = src_g[gv]
        sb["solveVelocitySpecialBoundaries"][:n_v] = mv_g[gv]
        sb["solveStressSpecialBoundaries"][:n_c] = mc_g[gc]

    ref = cases.make_domain(mesh, None, configs=cfg, open_water_band=None)
    mark(ref.blocklist[0], np.arange(nV), np.arange(nC))
    oracle.seaice_init_velocity_solver(ref)
    oracle.seaice_run_velocity_solver(ref)
    dom, halos, _ = parallel.make_partitioned_domain(mesh, None, 3, configs=cfg, open_water_band=None)
    for b, h in zip(dom.blocklist, halos):
        mark(b, h.globalID["vertex"], h.globalID["cell"])
    vs.seaice_mesh_pool_create(dom, device=0)
    try:
        dom.comm = parallel.LocalComm(dom, halos, 3)
        vs.seaice_init_velocity_solver(dom)
        vs.seaice_run_velocity_solver(dom)
    finally:
        vs.seaice_mesh_pool_destroy(dom)
    u = ref.blocklist[0].structs["velocity_solver"]["uVelocity"][:nV]
    assert np.abs(u).max() > 0 and np.all(u[typ_g == 3] == 0.0)
    sel = np.flatnonzero(typ_g == 2)
    assert np.abs(u[sel]).max() > 0 and np.allclose(u[sel], -u[src_g[sel] - 1], rtol=0, atol=0)
    for pool, name, loc, n in (("velocity_solver", "uVelocity", "vertex", nV),
                               ("velocity_solver", "vVelocity", "vertex", nV),
                               ("velocity_variational", "stress11", "cell", nC),
                               ("velocity_variational", "stress12", "cell", nC),
                               ("velocity_solver", "solveVelocity", "vertex", nV)):
        g = parallel.gather_global(dom.blocklist, halos, pool, name, n, loc).astype(np.float64)
        r = ref.blocklist[0].structs[pool][name][..., :n].astype(np.float64)
        err = np.abs(g - r).max() / max(np.abs(r).max(), 1e-300)
        assert err <= TOL, (name, err)


def _two_connected_blocks(max_cell_planes=24):
    mesh = get_mesh("QU480")
    dom, halos, _ = parallel.make_partitioned_domain(mesh, "QU480", 2, configs={"config_elastic_subcycle_number": 6})
    vs.seaice_mesh_pool_create(dom, device=0)
    parallel.CommBase(dom, halos, 2)  # gives the exchange lists to the two contexts
    for r, b in enumerate(dom.blocklist):
        b.ctx.comm_create(r, 2, max_cell_planes, 4)
    handles = [b.ctx.comm_handle() for b in dom.blocklist]
    for b in dom.blocklist:
        b.ctx.comm_connect(handles)
    return mesh, dom, halos


def test_library_exchange_between_two_blocks_of_one_process():
    """sih_halo_exchange_start / _finish (messages stored straight into the neighbour's receive buffer, sequence
    numbers awaited on the device) fill the halos with the owners' values: floats and ints, cell and vertex
    fields in one message, several exchanges in a row so that both parities of the buffers are used."""
    mesh, dom, halos = _two_connected_blocks()
    try:
        rng = np.random.default_rng(5)
        glob = {"uVelocity": rng.standard_normal(mesh.nVertices), "vVelocity": rng.standard_normal(mesh.nVertices),
                "stress11": rng.standard_normal((6, mesh.nCells)), "icePressure": rng.standard_normal(mesh.nCells)}
        gint = rng.integers(0, 5, mesh.nVertices).astype(np.int32)
        for rep in range(5):
            for b, h in zip(dom.blocklist, halos):
                d = b.dimensions
                for name, g in glob.items():
                    loc = parallel.field_loc(name)
                    n, nown = d["nVertices" if loc == "vertex" else "nCells"], h.nOwned[loc]
                    a = np.full(g.shape[:-1] + (n + 1,), -777.0, order="F")  # halo entries hold rubbish
                    a[..., :nown] = (g + rep)[..., h.globalID[loc][:nown]]
                    b.ctx.put(name, a)
                a = np.full(d["nVertices"] + 1, -7, dtype=np.int32)
                a[:h.nOwned["vertex"]] = (gint + rep)[h.globalID["vertex"][:h.nOwned["vertex"]]]
                b.ctx.put("solveVelocity", a)
            groups = (["uVelocity", "vVelocity", "stress11", "icePressure"], ["solveVelocity"])
            for ns in groups:
                for b in dom.blocklist:
                    b.ctx.halo_exchange_start(ns)
                for b in dom.blocklist:
                    b.ctx.halo_exchange_finish(ns)
            for b, h in zip(dom.blocklist, halos):
                b.ctx.comm_status()
                d = b.dimensions
                for name, g in glob.items():
                    loc = parallel.field_loc(name)
                    n = d["nVertices" if loc == "vertex" else "nCells"]
                    a = np.zeros(g.shape[:-1] + (n + 1,), order="F")
                    b.ctx.get(name, a)
                    assert np.array_equal(a[..., :n], (g + rep)[..., h.globalID[loc][:n]]), (rep, name)
                a = np.zeros(d["nVertices"] + 1, dtype=np.int32)
                b.ctx.get("solveVelocity", a)
                assert np.array_equal(a[:d["nVertices"]], (gint + rep)[h.globalID["vertex"][:d["nVertices"]]]), rep
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_a_silent_neighbour_is_reported_and_every_kernel_still_ends():
    """The device-side wait for a neighbour's message is bounded: with a limit of 20 ms and a neighbour that
    never sends, the halo stays as it was, sih_comm_status reports it, and the stream drains normally."""
    from mpas_seaice_amd import lib as sihlib
    mesh, dom, halos = _two_connected_blocks()
    try:
        b0 = dom.blocklist[0]
        n = b0.dimensions["nVertices"]
        u = np.arange(n + 1, dtype=np.float64)
        b0.ctx.put("uVelocity", u)
        b0.ctx.comm_set_timeout(20.0)
        b0.ctx.halo_exchange(["uVelocity", "vVelocity"])  # block 1 never starts its side
        with pytest.raises(sihlib.SeaiceHipError, match="did not arrive"):
            b0.ctx.comm_status()
        back = np.zeros(n + 1)
        b0.ctx.get("uVelocity", back)
        assert np.array_equal(back[:n], u[:n])
        with pytest.raises(sihlib.SeaiceHipError, match="larger than the plane counts"):
            b0.ctx.halo_exchange(["stress11", "stress22", "stress12", "strain11", "strain22", "strain12",
                                  "replacementPressure", "principalStress1", "principalStress2"])
    finally:
        vs.seaice_mesh_pool_destroy(dom)

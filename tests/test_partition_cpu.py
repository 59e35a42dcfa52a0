"""Block decomposition + halo exchange, on the CPU: bit-identical to the single-block answer."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from mpas_seaice_amd import cases, parallel, partition
from util import get_mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_structure():
    mesh = get_mesh("geo8")
    meshes, halos = partition.partition_mesh(mesh, 3)
    assert sum(m.nCellsSolve for m in meshes) == mesh.nCells
    assert sum(m.nVerticesSolve for m in meshes) == mesh.nVertices
    assert sum(m.nEdgesSolve for m in meshes) == mesh.nEdges
    for r, (m, h) in enumerate(zip(meshes, halos)):
        # every owned vertex sees its three cells inside the block
        cov = m["cellsOnVertex"][:, :m.nVerticesSolve]
        assert (cov <= m.nCells).all()
        # send / recv lists pair up
        for loc in ("cell", "vertex", "edge"):
            for p, idx in h.recv[loc].items():
                other = halos[p].send[loc][r]
                assert len(other) == len(idx)
                assert np.array_equal(h.globalID[loc][idx - 1], halos[p].globalID[loc][other - 1])
                assert (idx > h.nOwned[loc]).all() and (other <= halos[p].nOwned[loc]).all()


@pytest.mark.parametrize("nblocks", [2, 5])
def test_multiblock_oracle_flow_is_bit_identical(oracle, nblocks):
    mesh = get_mesh("geo8")
    cfg = {"config_elastic_subcycle_number": 15, "config_dt": 7200.0}
    ref = cases.make_domain(mesh, None, configs=cfg, open_water_band=(-0.2, 0.3))
    oracle.seaice_init_velocity_solver(ref)
    oracle.seaice_run_velocity_solver(ref)
    dom, halos, _ = parallel.make_partitioned_domain(mesh, None, nblocks, configs=cfg, open_water_band=(-0.2, 0.3))
    comm = parallel.LocalComm(dom, halos, nblocks)
    dom.halo = parallel.oracle_halo_callback(comm)
    dom.allreduce_min = comm.allreduce_min
    oracle.seaice_init_velocity_solver(dom)
    oracle.seaice_run_velocity_solver(dom)
    for pool, name, loc, n in (("velocity_solver", "uVelocity", "vertex", mesh.nVertices),
                               ("velocity_solver", "vVelocity", "vertex", mesh.nVertices),
                               ("velocity_variational", "stress22", "cell", mesh.nCells),
                               ("velocity_solver", "shear", "cell", mesh.nCells),
                               ("velocity_solver", "oceanStressCellV", "cell", mesh.nCells)):
        g = parallel.gather_global(dom.blocklist, halos, pool, name, n, loc)
        r = ref.blocklist[0].structs[pool][name][..., :n]
        assert np.array_equal(g, r), name


@pytest.mark.parametrize("extra", [[], ["upwind"]])
def test_two_process_gloo_halo_exchange(extra):
    """world_size 2 over gloo: each rank owns one block and runs the oracle flow on it (velocity solver,
    incremental remap; with "upwind" one more step of the upwind advection)."""
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), "cpu", "geo8"] + extra
    env = dict(os.environ, OMP_NUM_THREADS="2")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count(": ok") == 2, out.stdout


def test_the_benchmark_tries_a_thinner_halo_only_where_it_makes_the_blocks_fit():
    """bench.py on N GPUs: 8 halo layers unless the blocks are then too large for the persistent kernel and fewer
    layers make all of them fit (QU60 on 4 GPUs on the real thing; QU240 here with the room declared accordingly)"""
    mesh = get_mesh("QU240")
    sizes = {}
    for k in (8, 4, 2):
        _, halos = partition.partition_mesh(mesh, 4, num_halos=k)
        sizes[k] = max(len(h.globalID["cell"]) for h in halos)
    assert sizes[8] > sizes[4] > sizes[2]
    assert parallel.thinner_halo_that_fits(mesh, 4, 8, sizes[8]) == 8          # fits as it is
    assert parallel.thinner_halo_that_fits(mesh, 4, 8, sizes[8] - 1) == 4      # the deepest halo that fits
    assert parallel.thinner_halo_that_fits(mesh, 4, 8, sizes[4] - 1) == 2
    assert parallel.thinner_halo_that_fits(mesh, 4, 8, sizes[2] - 1) == 8      # nothing fits: no change
    assert parallel.thinner_halo_that_fits(mesh, 4, 2, sizes[2] - 1) == 2      # nothing below the asked depth


def test_plain_multi_rank_bench_command_ends_promptly_without_a_gpu():
    """`python bench.py --gpus 2` with no launcher: the parent starts two rank processes by itself; here, without a
    GPU, both must say so and the command must end with an error code instead of waiting for anything."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SIH_BENCH_NO_RETRY"] = "1"
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mesh", "QU480", "--steps", "1",
                          "--warmup", "0"], capture_output=True, text=True, timeout=240, env=env)
    import torch
    if torch.cuda.is_available():
        assert out.returncode == 0, out.stderr[-2000:]
    else:
        assert out.returncode != 0 and "needs a GPU" in out.stderr, out.stderr[-2000:]
        assert time.time() - t0 < 200

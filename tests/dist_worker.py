"""Worker for the multi-process halo-exchange tests (launched by torch.distributed.run).

mode cpu : every rank runs the CPU oracle flow on its own block, exchanging halos over gloo.
mode gpu : every rank drives the HIP library on its own block (all ranks share GPU 0), exchanging
           halos over gloo through host staging, or with "p2p" through the library's own transport
           (receive buffers mapped into the neighbours through HIP IPC; gloo only carries the handles).
Each rank also computes the single-block oracle answer and checks its owned entities against it.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from mpas_seaice_amd import cases, parallel  # noqa: E402
from oracle import evp_flow as ora  # noqa: E402
from oracle import ir_flow as ira  # noqa: E402
from oracle import upwind_flow as upw  # noqa: E402
from util import get_mesh  # noqa: E402


def main():
    mode = sys.argv[1]
    meshname = sys.argv[2] if len(sys.argv) > 2 else "geo8"
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ora.set_num_threads(2)
    mesh = get_mesh(meshname)
    cfg = {"config_elastic_subcycle_number": 12, "config_dt": 7200.0}
    if "long" in sys.argv[3:]:  # more exchanges in a step than one launch of the persistent kernel may hold
        cfg["config_elastic_subcycle_number"] = 40
    if "overlap" in sys.argv[3:]:  # exchange hidden behind interior work; only the device path has it
        cfg["config_overlap_halo_exchange"] = True
    if "graph" in sys.argv[3:]:  # chunks of the subcycle loop (iterations + exchange) replayed from HIP graphs
        cfg["config_subcycle_graph"] = True
    pcfg = dict(cfg)
    if "deep" in sys.argv[3:]:  # three halo layers, one exchange per three iterations
        pcfg.update(config_num_halos=3, config_halo_exchange_interval=3)
    ref = cases.make_domain(mesh, None, configs=cfg, open_water_band=(-0.2, 0.3))
    ora.seaice_init_velocity_solver(ref)
    ira.seaice_init_advection(ref)
    ora.seaice_run_velocity_solver(ref)
    reinit = "reinit" in sys.argv[3:]  # two velocity-solver steps, the device's tables rebuilt between them
    if reinit:
        ora.seaice_run_velocity_solver(ref)
    ira.seaice_run_advection(ref)
    upwind = "upwind" in sys.argv[3:]  # one more step with the upwind advection, surface temperature in kelvin
    if upwind:
        ref.blocklist[0].structs["tracers"]["surfaceTemperature"][..., :mesh.nCells] += 273.15
        upw.seaice_init_advection_upwind(ref)
        upw.seaice_run_advection_upwind(ref)
    dom, halo, _ = parallel.make_partitioned_domain(mesh, None, world, configs=pcfg, only_block=rank,
                                                    open_water_band=(-0.2, 0.3))
    if mode == "cpu":
        comm = parallel.DistComm(dom, halo, world, rank)
        dom.halo = parallel.oracle_halo_callback(comm)
        dom.allreduce_min = comm.allreduce_min
        ora.seaice_init_velocity_solver(dom)
        ira.seaice_init_advection(dom)
        ora.seaice_run_velocity_solver(dom)
        ira.seaice_run_advection(dom)
        if upwind:
            dom.blocklist[0].structs["tracers"]["surfaceTemperature"][..., :dom.blocklist[0].dimensions["nCells"]] += 273.15
            upw.seaice_init_advection_upwind(dom)
            upw.seaice_run_advection_upwind(dom)
        tol = 0.0
    else:
        from mpas_seaice_amd import velocity_solver as vs
        torch.cuda.set_device(0)
        vs.seaice_mesh_pool_create(dom, device=0)
        peer = "p2p" in sys.argv[3:]  # the library's own transport: messages stored into the neighbour's buffer
        dom.comm = (parallel.PeerComm if peer else parallel.DistComm)(dom, halo, world, rank)
        from mpas_seaice_amd import advection as adv
        vs.seaice_init_velocity_solver(dom)
        adv.seaice_init_advection(dom)
        if "statusonly" in sys.argv[3:]:
            # The way the Fortran shim drives a step on several ranks: pre-cycle with its exchanges between launches
            # (both ranks take part), then the subcycle loop with the exchange inside the kernel, then the transport's
            # status -- and nothing else that waits for the stream.  Rank 1 never starts its loop: rank 0's launch gives
            # up within its time limit, and the status call alone must say so.
            from mpas_seaice_amd import lib as sihlib
            agreed = dom.comm.agree_on_exchange_inside(dom)
            for b in dom.blocklist:
                vs._upload(dom, b)
            vs.velocity_solver_pre_subcycle(dom)
            dom.comm.check(dom)
            dist.barrier()
            if rank == 0:
                ctx0 = dom.blocklist[0].ctx
                ctx0.subcycle_exchange(12, last=False, interval=3, overlap=False, use_graph=False)
                try:
                    dom.comm.check(dom)
                    print(f"rank 0: no error from the status call (agreed {agreed})", flush=True)
                except sihlib.SeaiceHipError as e:
                    print(f"rank 0 status call reported: {e}", flush=True)
            dist.barrier()
            vs.seaice_mesh_pool_destroy(dom)
            dist.destroy_process_group()
            return
        if "silent" in sys.argv[3:]:
            # rank 1 never runs its dynamics: rank 0's kernel must give up within its time limit, end normally and
            # leave an error for the host (the time limit comes from SIH_PERSIST_TIMEOUT_MS)
            from mpas_seaice_amd import lib as sihlib
            if rank == 0:
                dom.comm.agree_on_exchange_inside(dom)
                try:
                    vs.seaice_run_velocity_solver(dom)
                    dom.blocklist[0].ctx.sync()
                    print("rank 0: no error", flush=True)
                except sihlib.SeaiceHipError as e:
                    print(f"rank 0 reported: {e}", flush=True)
            else:
                dom.comm.agree_on_exchange_inside(dom)
            dist.barrier()
            vs.seaice_mesh_pool_destroy(dom)
            dist.destroy_process_group()
            return
        vs.seaice_run_velocity_solver(dom)
        if reinit:
            # every rank rebuilds the persistent kernel's tables after the first step's exchanges: the tags and the
            # bank parity of the words in the receive regions must go on from where they were
            dom.blocklist[0].ctx.static_changed()
            vs.seaice_run_velocity_solver(dom)
        if os.environ.get("SIH_TEST_DUMP"):
            sv, vv = dom.blocklist[0].structs["velocity_solver"], dom.blocklist[0].structs["velocity_variational"]
            np.savez(os.path.join(os.environ["SIH_TEST_DUMP"], f"rank{rank}.npz"), u=sv["uVelocity"], v=sv["vVelocity"],
                     s11=vv["stress11"], s22=vv["stress22"], s12=vv["stress12"], div=sv["divergence"])
        adv.seaice_run_advection(dom)
        if upwind:
            from mpas_seaice_amd import advection_upwind
            dom.blocklist[0].structs["tracers"]["surfaceTemperature"][..., :dom.blocklist[0].dimensions["nCells"]] += 273.15
            advection_upwind.seaice_init_advection_upwind(dom)
            advection_upwind.seaice_run_advection_upwind(dom)
        ctx = dom.blocklist[0].ctx
        print(f"rank {rank} persistent kernel {int(ctx.info('persistentKernel'))} launches "
              f"{int(ctx.info('persistentLaunches'))} with exchange {int(ctx.info('persistentExchangeLaunches'))}"
              f" tile cells {int(ctx.info('persistentTileCells'))}", flush=True)
        if peer:
            dom.comm.check(dom)
            dist.barrier()  # no rank unmaps its buffer while a neighbour may still write to it
        vs.seaice_mesh_pool_destroy(dom)
        tol = 1e-10
    b = dom.blocklist[0]
    rb = ref.blocklist[0]
    worst = 0.0
    for pool, name, loc in (("velocity_solver", "uVelocity", "vertex"), ("velocity_solver", "vVelocity", "vertex"),
                            ("velocity_variational", "stress11", "cell"), ("velocity_variational", "stress12", "cell"),
                            ("velocity_solver", "oceanStressCellU", "cell"), ("velocity_solver", "divergence", "cell"),
                            ("velocity_solver", "solveVelocity", "vertex"), ("tracers", "iceAreaCategory", "cell"),
                            ("tracers", "iceVolumeCategory", "cell"), ("tracers", "iceEnthalpy", "cell"),
                            ("tracers", "snowEnthalpy", "cell"), ("tracers", "surfaceTemperature", "cell"),
                            ("tracers", "snowVolumeCategory", "cell")):
        nown = halo.nOwned[loc]
        g = halo.globalID[loc][:nown]
        mine = b.structs[pool][name][..., :nown]
        theirs = rb.structs[pool][name][..., g]
        den = max(np.abs(theirs).max(), 1e-300)
        err = float(np.abs(mine.astype(np.float64) - theirs).max() / den)
        worst = max(worst, err)
        assert err <= tol, (rank, name, err)
    # halo copies of the velocities must equal the owners' values too
    gv = halo.globalID["vertex"]
    u = b.structs["velocity_solver"]["uVelocity"][:len(gv)]
    err = float(np.abs(u - rb.structs["velocity_solver"]["uVelocity"][gv]).max())
    assert err <= tol * max(np.abs(u).max(), 1e-300) + 0.0, (rank, "halo u", err)
    print(f"rank {rank}/{world} mode {mode}: ok, worst rel err {worst:.2e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Time the subcycle kernels of ONE block of an N-way partition of QU60 (no exchange, so the numbers are the
compute floor of a many-GPU run): plain chunks, graph-replayed chunks, and the piece sequences of the
overlapped exchange."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mpas_seaice_amd import mesh as meshmod, parallel, velocity_solver as vs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mesh", default="QU60")
ap.add_argument("--blocks", type=int, default=8)
ap.add_argument("--interval", type=int, default=8)
ap.add_argument("--block", type=int, default=3)
ap.add_argument("--chunks", type=int, default=200)
args = ap.parse_args()

mesh = meshmod.make_named_mesh(args.mesh)
k = args.interval
cfg = {"config_num_halos": max(2, k), "config_halo_exchange_interval": k, "config_overlap_halo_exchange": True}
dom, halo, _ = parallel.make_partitioned_domain(mesh, args.mesh, args.blocks, configs=cfg, only_block=args.block)
vs.seaice_mesh_pool_create(dom)
b = dom.blocklist[0]
ctx = b.ctx
# exchange lists are needed for the subsets; no transport is attached
L = {loc: parallel._Lists(halo, loc, args.blocks) for loc in ("cell", "vertex")}
for loc in ("cell", "vertex"):
    ctx.halo_set_lists(parallel.LOC_ID[loc], L[loc].send_idx, L[loc].recv_idx, L[loc].send_counts, L[loc].recv_counts)
vs.seaice_init_velocity_solver(dom)
vs._upload(dom, b)
ctx.pre_subcycle(-1)
ctx.overlap_setup()
ctx.sync()
CA, CI, CR, VA, VS, VR = range(6)


def timed(fn, n):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / n * 1e6


out = {"cells_in_block": b.dimensions["nCells"], "owned": b.dimensions["nCellsSolve"], "interval": k}
out["plain_us_per_subcycle"] = timed(lambda: ctx.subcycle(k), args.chunks) / k
out["graph_us_per_subcycle"] = timed(lambda: ctx.subcycle_graph(k), args.chunks) / k
out["persistent_kernel"] = ctx.info("persistentKernel")
out["persistent_per_cu"], out["persistent_reason"] = ctx.info("persistentPerCU"), ctx.info("persistentReason")
out["persistent_tiles"], out["persistent_arrived"] = ctx.info("persistentTiles"), ctx.info("persistentArrived")
if out["persistent_kernel"]:
    out["persistent_us_per_subcycle"] = timed(lambda: ctx.subcycle_persistent(k), args.chunks) / k
    out["persistent_us_per_subcycle_120"] = timed(lambda: ctx.subcycle_persistent(120), 20) / 120
out["fused_kernel"] = ctx.info("fusedKernel")
if out["fused_kernel"]:
    out["fused_us_per_subcycle"] = timed(lambda: ctx.subcycle_fused(k), args.chunks) / k
    out["fused_us_per_subcycle_119"] = timed(lambda: ctx.subcycle_fused(119), 20) / 119
ctx.set_persistent(0)
out["two_kernel_graph_us_per_subcycle"] = timed(lambda: ctx.subcycle_graph(k), args.chunks) / k
out["two_kernel_graph_us_per_subcycle_119"] = timed(lambda: ctx.subcycle_graph(119), 20) / 119
ctx.set_persistent(-1)
body = [CR, VA] + [CA, VA] * (k - 2) + [CA, VS]
tail = [VR, CI]


def overlapped():
    ctx.subcycle_parts(body, True)
    ctx.subcycle_parts(tail, True)


out["overlap_pieces_us_per_subcycle"] = timed(overlapped, args.chunks) / k


def overlapped_plain():
    ctx.subcycle_parts(body, False)
    ctx.subcycle_parts(tail, False)


out["overlap_pieces_no_graph_us_per_subcycle"] = timed(overlapped_plain, args.chunks) / k
for name, parts in (("CA", [CA]), ("CI", [CI]), ("CR", [CR]), ("VA", [VA]), ("VS", [VS]), ("VR", [VR])):
    out["piece_us_" + name] = timed(lambda p=parts: ctx.subcycle_parts(p * 8, False), 50) / 8
names = ["uVelocity", "vVelocity", "stress11", "stress22", "stress12"]
import torch  # noqa: E402
nfC, nfV = 3 * b.dimensions["maxEdges"], 2
ns = int(sum(L["cell"].send_counts) * nfC + sum(L["vertex"].send_counts) * nfV)
nr = int(sum(L["cell"].recv_counts) * nfC + sum(L["vertex"].recv_counts) * nfV)
sbuf = torch.empty(max(ns, 1), dtype=torch.float64, device="cuda")
rbuf = torch.zeros(max(nr, 1), dtype=torch.float64, device="cuda")
order = ["stress11", "stress22", "stress12", "uVelocity", "vVelocity"]
torch.cuda.synchronize()
out["pack_unpack_us"] = timed(lambda: (ctx.halo_pack(2, order, sbuf.data_ptr()), ctx.halo_unpack(2, order, rbuf.data_ptr())),
                              args.chunks)
out["message_bytes"] = ns * 8
print(json.dumps(out))
vs.seaice_mesh_pool_destroy(dom)

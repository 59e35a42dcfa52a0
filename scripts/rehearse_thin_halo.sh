#!/bin/bash
# Rehearsal, on a box with ONE GPU, of bench.py's choice of a thinner halo where that lets every block stay resident
# (on the real thing: QU60 on 4 GPUs).  QU240 in 4 ranks with the room declared small enough to trigger it: first the
# case where the thin halo is kept, then one where the persistent kernel is refused and the run falls back to 8 layers.
set -e
export SIH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 SIH_HALO_TIMEOUT_MS=3000
P=29740
for CASE in keep fallback; do
  P=$((P+1))
  if [ $CASE = keep ]; then EXTRA="SIH_BENCH_ROOM_CELLS=4000"; else EXTRA="SIH_BENCH_ROOM_CELLS=4000 SIH_PERSIST_CAPACITY=1 SIH_PERSIST_SMALL_TILES=0"; fi
  env $EXTRA timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 4 --mesh QU240 --steps 5 --warmup 2 > gpurun_out/thin_$CASE.json 2> gpurun_out/thin_$CASE.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/thin_$CASE.json").read().strip().splitlines()[-1])
print("$CASE:", "us/iter", round(d["ms_per_step"]/120*1e3,2), d["roofline"]["kernel"][:30], "inside", d["exchange"]["inside_the_subcycle_kernel"], "cells", d["exchange"]["block_cells_with_halo"], "|", d["config"]["parallelism"][:60], "|", d["config"]["parallelism"][-150:])
PY
done

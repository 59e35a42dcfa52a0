#!/bin/bash
# Rehearsal of bench.py's multi-rank path on a box with ONE GPU: the ranks share the device (gloo rendezvous, the
# library's own transport over HIP IPC), on meshes small enough for all ranks' persistent kernels to be resident
# together.  Not a scaling measurement: it checks the flow end to end and compares the two exchange modes.
set -e
export SIH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
P=29540
for R in 2 4; do
  for X in 1 0; do
    P=$((P+1))
    SIH_PERSIST_EXCHANGE=$X timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port $P bench.py --gpus $R --mesh QU240 --steps 10 --warmup 3 > gpurun_out/reh_${R}_$X.json 2> gpurun_out/reh_${R}_$X.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/reh_${R}_$X.json").read().strip().splitlines()[-1])
print("ranks $R inside $X:", "ms_per_step", round(d["ms_per_step"],4), "us/iter", round(d["ms_per_step"]/120*1e3,2), d["exchange"]["inside_the_subcycle_kernel"], d["exchange"]["inside_note"], "chain us", round(d["exchange"]["us_per_exchange_in_a_chain_of_50"],1))
PY
  done
done

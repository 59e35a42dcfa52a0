#!/bin/bash
# Rehearsal of bench.py's 2- and 4-rank paths at full size on a box with ONE GPU (the ranks share the device): blocks
# of QU60 too large for the persistent kernel, which run the fused one-launch-per-iteration kernel between their
# exchanges.  Not a scaling measurement (the ranks take turns on one device): it checks the flow end to end.
set -e
export SIH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
P=29640
for R in 2 4; do
  P=$((P+1))
  SIH_HALO_TIMEOUT_MS=3000 timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port $P bench.py --gpus $R --steps 3 --warmup 1 > gpurun_out/reh60_$R.json 2> gpurun_out/reh60_$R.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/reh60_$R.json").read().strip().splitlines()[-1])
print("ranks $R:", "ms_per_step", round(d["ms_per_step"],3), "us/iter", round(d["ms_per_step"]/120*1e3,2), d["roofline"]["kernel"][:40], "inside", d["exchange"]["inside_the_subcycle_kernel"], "cells", d["exchange"]["block_cells_with_halo"], d["exchange"]["transport"])
PY
done

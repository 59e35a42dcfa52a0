#!/bin/bash
# Repeats the multi-rank runs with the halo exchange inside the persistent kernel (ranks sharing one GPU, every rank
# checked against the single-block oracle) over tile sizes, intervals and rank counts; prints one line per run.
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2 SIH_PERSIST_EXCHANGE=1 SIH_HALO_TIMEOUT_MS=5000
P=29800
for rep in 1 2 3; do
for small in 0 1; do
for cfg in "QU240 2 p2p long" "QU240 3 p2p long" "QU240 4 p2p long" "QU240 3 p2p deep long" "QU480 4 p2p long" "QU240 4 p2p deep long"; do
  set -- $cfg
  mesh=$1; R=$2; shift 2
  P=$((P+1))
  if [ $small = 1 ]; then export SIH_PERSIST_CAPACITY=1 SIH_PERSIST_CAPACITY_SMALL=1024; else unset SIH_PERSIST_CAPACITY SIH_PERSIST_CAPACITY_SMALL; fi
  ok=$(timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port $P tests/dist_worker.py gpu $mesh "$@" 2>&1 | grep -o ": ok" | wc -l)
  echo "rep $rep small $small $mesh ranks $R $@: ok on $ok of $R"
done
done
done

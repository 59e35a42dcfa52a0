#!/bin/bash
# Repeats the multi-rank runs with the halo exchange inside the persistent kernel (ranks sharing one GPU, every rank
# checked against the single-block oracle) over tile sizes, intervals and rank counts; prints one line per run.
# For code believed to be correct only: the script ends, with a non-zero exit code, at the FIRST run that is not ok on
# every rank or that had to be ended by its time limit, and starts nothing on the device after it.
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2 SIH_PERSIST_EXCHANGE=1 SIH_HALO_TIMEOUT_MS=5000
P=29800
mkdir -p gpurun_out
for rep in 1 2 3; do
for small in 0 1; do
for cfg in "QU240 2 p2p long" "QU240 3 p2p long" "QU240 4 p2p long" "QU240 3 p2p deep long" "QU480 4 p2p long" "QU240 4 p2p deep long"; do
  set -- $cfg
  mesh=$1; R=$2; shift 2
  P=$((P+1))
  if [ $small = 1 ]; then export SIH_PERSIST_CAPACITY=1 SIH_PERSIST_CAPACITY_SMALL=1024; else unset SIH_PERSIST_CAPACITY SIH_PERSIST_CAPACITY_SMALL; fi
  log=gpurun_out/soak_last.log
  timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port $P tests/dist_worker.py gpu $mesh "$@" > $log 2>&1
  rc=$?
  ok=$(grep -c ": ok" $log)
  echo "rep $rep small $small $mesh ranks $R $@: ok on $ok of $R (exit code $rc)"
  if [ $rc -ne 0 ] || [ "$ok" != "$R" ]; then
    echo "---- stopping at the first run that is not ok; the end of its output:"
    tail -n 30 $log
    exit 1
  fi
done
done
done

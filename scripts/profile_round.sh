#!/bin/bash
# The profile runs whose summaries go to profiles/: kernel trace of bench.py, the two HBM counter passes (each in a
# run of its own), kernel trace of the one-block timing of an 8-way partition.  Run on the GPU box from the repo root.
set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-hbm-only > $O/stats.log 2>&1
echo stats done
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-hbm-only > $O/fetch.log 2>&1
echo fetch done
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-hbm-only > $O/write.log 2>&1
echo write done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/block8 -o b -- python3 scripts/bench_block.py --blocks 8 --interval 8 --chunks 50 > $O/block8.log 2>&1
echo block8 done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/qu120 -o b -- python3 scripts/bench_block.py --mesh QU120 --blocks 1 --block 0 --interval 8 --chunks 50 > $O/qu120.log 2>&1
echo qu120 done
find $O -name "*.csv" | xargs ls -la

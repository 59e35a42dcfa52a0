#!/bin/bash
# maxEdges 7: bench.py on variable-resolution meshes (var256: 655 362 cells, a block that is bound by HBM alone; var128:
# 163 842 cells, the size of QU60), with the kernel trace of the larger run, and the one-block timing of an 8-way
# partition of var128 (the persistent kernel's tiles of 32 cells).  Run on the GPU box from the repo root.
set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/m7
mkdir -p $O
timeout -k 10 400 python3 bench.py --mesh var256 --steps 10 --warmup 2 --no-hbm-only > $O/bench_var256.json 2> $O/bench_var256.err
echo var256 done
timeout -k 10 300 python3 bench.py --mesh var128 --steps 10 --warmup 2 --no-hbm-only > $O/bench_var128.json 2> $O/bench_var128.err
echo var128 done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 bench.py --mesh var256 --steps 5 --warmup 1 --no-cpu-baseline --no-hbm-only > $O/stats.log 2>&1
echo stats done
timeout -k 10 300 python3 scripts/bench_block.py --mesh var128 --blocks 8 --interval 8 --chunks 50 > $O/block8_var128.json 2> $O/block8_var128.err
echo block8 done
timeout -k 10 300 python3 scripts/bench_block.py --mesh var128 --blocks 8 --interval 2 --chunks 50 > $O/block8_var128_h2.json 2> $O/block8_var128_h2.err
echo block8 h2 done
find $O -name "*stats*.csv" | xargs ls -la

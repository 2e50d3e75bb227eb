#!/bin/bash
# round-3 profile collection on the GPU box: the default bench line, its rocprofv3 kernel statistics, and the PMC traffic of the
# dominant kernel (separate counter passes).  usage: tools/profile_r03.sh <commit>
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03
mkdir -p $OUT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o c4 -- python3 bench.py --no-cpu-baseline --no-other-configs --no-mixed-extra > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/r03_c4_bench_kernel_stats.csv
rm -rf $OUT/kt
echo "kernel stats done"
bash tools/pmc_collect.sh "$1" > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_r03/r03_pmc_schur_inner_c4.json $OUT/
rm -rf gpurun_out/pmc_r03/fetch gpurun_out/pmc_r03/write
echo "pmc done"

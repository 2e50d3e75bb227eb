#!/bin/bash
# HBM traffic of the dominant kernel for bench.py's roofline.traffic (run on the GPU box through gpurun, ~6 min):
# two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with trace domains) over the C4 bench command,
# reduced by tools/pmc_inner.py to profiles/r03_pmc_schur_inner_c4.json together with the commit and the launch geometry
# (algorithmic bytes per launch of the same run), which bench.py checks against the live kernel before it prints `traffic`.
#   usage: tools/pmc_collect.sh <commit>
set -e
export TMPDIR=/tmp
OUT=gpurun_out/pmc_r03
mkdir -p $OUT
ARGS="bench.py --no-cpu-baseline --no-other-configs --no-mixed-extra"
python3 $ARGS > $OUT/bench_plain.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/bench_fetch.json
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/bench_write.json
F=$(find $OUT/fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/write -name '*counter_collection.csv' | head -1)
python3 tools/pmc_inner.py "$F" "$W" $OUT/r03_pmc_schur_inner_c4.json \
  "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over $ARGS (C4 full solve, MI355X, round 3)" "$1" $OUT/bench_plain.json > /dev/null
cat $OUT/r03_pmc_schur_inner_c4.json

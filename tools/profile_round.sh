#!/bin/bash
# Per-round profile collection on the GPU box (through gpurun, ~18 min):  usage: tools/profile_round.sh <round tag, e.g. r04> <commit>
#  1. HBM traffic of the dominant kernel: two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with a trace
#     domain) over the C4 bench command, reduced by tools/pmc_inner.py to <tag>_pmc_schur_inner_c4.json with the commit and the launch
#     geometry of the same build.  The file is copied into profiles/ of THIS tree before step 3, so that the stored default bench line
#     quotes counters of the build it ran on (round 3 ran the bench first and so always quoted the previous collection).
#  2. rocprofv3 --kernel-trace --stats over the same command: per-kernel time of the C4 solve.
#  3. the driver's default command (python3 bench.py), whole line with its extras.
set -e
TAG=${1:-r04}; COMMIT=${2:-unknown}; STAGE=${3:-all}      # stage: all | c4 (steps 1-2) | rest (step 4 + the default bench line): two gpurun calls
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="bench.py --no-cpu-baseline --no-other-configs --no-mixed-extra"
if [ "$STAGE" != "rest" ]; then
python3 $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err
echo "plain run done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/bench_write.err
echo "write pass done"
F=$(find $OUT/fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/write -name '*counter_collection.csv' | head -1)
python3 tools/pmc_inner.py "$F" "$W" $OUT/${TAG}_pmc_schur_inner_c4.json \
  "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over $ARGS (C4 full solve, MI355X, round ${TAG#r})" "$COMMIT" $OUT/bench_plain.json > /dev/null
rm -rf $OUT/fetch $OUT/write
cp $OUT/${TAG}_pmc_schur_inner_c4.json profiles/${TAG}_pmc_schur_inner_c4.json
echo "pmc done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o c4 -- python3 $ARGS > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_c4_bench_kernel_stats.csv
rm -rf $OUT/kt
echo "kernel stats done"
fi
if [ "$STAGE" = "c4" ]; then exit 0; fi
#  4. (round 5) the other configurations the bench line quotes, on the same build: MFMA-busy counters of the dense factorization kernel at
#     C2 (one --pmc pass, no trace domain; installed in profiles/ BEFORE the default bench run, which cites it), rocprofv3 kernel statistics
#     of the C2 solve and of the C3 batch kernel, kernel timelines of one C1 and one n = 1000 solve.
C2ARGS="bench.py --workload C2 --steps 3 --warmup 1 --no-cpu-baseline --no-mixed-extra"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 $C2ARGS > $OUT/bench_c2_mfma.json 2> $OUT/bench_c2_mfma.err
python3 tools/pmc_mfma_util.py "$(find $OUT/mfma -name '*counter_collection.csv' | head -1)" $OUT/${TAG}_c2_mid_factor_mfma_util.json \
  "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over $C2ARGS (MI355X, round ${TAG#r})" "$COMMIT" > $OUT/mfma_summary.txt
rm -rf $OUT/mfma
cp $OUT/${TAG}_c2_mid_factor_mfma_util.json profiles/${TAG}_c2_mid_factor_mfma_util.json
echo "c2 mfma done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt2 -o c2 -- python3 $C2ARGS > $OUT/bench_c2_profiled.json 2> $OUT/bench_c2_profiled.err
cp $(find $OUT/kt2 -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_c2_kernel_stats.csv
rm -rf $OUT/kt2
python3 $C2ARGS > $OUT/${TAG}_c2_bench_run.json 2> $OUT/bench_c2.err
echo "c2 stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt3 -o c3 -- python3 bench.py --workload C3 --steps 4 --stream-depth 1 --no-cpu-baseline > $OUT/bench_c3_profiled.json 2> $OUT/bench_c3_profiled.err
cp $(find $OUT/kt3 -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_c3_batch_kernel_stats.csv
rm -rf $OUT/kt3
echo "c3 stats done"
bash tools/mid_trace.sh C1 > $OUT/${TAG}_mid_trace_c1.txt 2>&1
bash tools/mid_trace.sh n1000 > $OUT/${TAG}_mid_trace_n1000.txt 2>&1
python3 tools/mid_latency.py > $OUT/${TAG}_mid_size_latency.txt 2>&1
echo "mid traces done"
python3 bench.py > $OUT/${TAG}_c4_bench_default_run.json 2> $OUT/bench_default.err
echo "bench done"

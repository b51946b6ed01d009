#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Keeps only small summaries under gpurun_out/prof_summary/ (the raw traces are large).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_summary
RAW=/tmp/prof_raw
rm -rf $RAW; mkdir -p $OUT $RAW
# (--no-video: the configs[4] leg launches the same scan kernels on other shapes; kept out so that the per-kernel averages
#  of the summary are those of the timed region)
ARGS="${BENCH_ARGS:---steps 3 --warmup 1 --no-cpu-baseline --no-video --no-sharded-leg}"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/kt -- python3 bench.py $ARGS > $OUT/bench_under_kernel_trace.json 2> $RAW/kt.err
f=$(find $RAW/kt -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
t=$(find $RAW/kt -name '*kernel_trace.csv' | head -1)
if [ -n "$t" ]; then head -1 "$t" > $OUT/kernel_trace_cbh.csv; grep -E "cbh|rocprim|hipcub" "$t" >> $OUT/kernel_trace_cbh.csv; fi
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $RAW/pmc_$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-sharded-leg > $OUT/bench_under_pmc_$c.json 2> $RAW/pmc_$c.err
  p=$(find $RAW/pmc_$c -name '*counter_collection.csv' | head -1)
  if [ -n "$p" ]; then head -1 "$p" > $OUT/pmc_${c}_cbh.csv; grep -E "cbh" "$p" >> $OUT/pmc_${c}_cbh.csv; fi
done
tail -3 $RAW/*.err | cut -c1-300
ls -la $OUT; du -sh $OUT

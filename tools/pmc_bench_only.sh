cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_summary; RAW=/tmp/prof_raw; mkdir -p $OUT $RAW
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $RAW/pmc_$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-sharded-leg > $OUT/bench_under_pmc_$c.json 2> $RAW/pmc_$c.err
  p=$(find $RAW/pmc_$c -name '*counter_collection.csv' | head -1)
  if [ -n "$p" ]; then head -1 "$p" > $OUT/pmc_${c}_cbh.csv; grep -E "cbh" "$p" >> $OUT/pmc_${c}_cbh.csv; fi
done
ls -la $OUT

#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 --kernel-trace --stats of one python tool; keeps the kernel stats summary.
#   tools/profile_cmd.sh <name> <script.py> [args...]   ->  gpurun_out/prof_<name>_kernel_stats.csv
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAW=/tmp/prof_raw_$name
rm -rf $RAW; mkdir -p $RAW gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -- python3 "$@" > gpurun_out/prof_${name}.out 2> $RAW/err.log
f=$(find $RAW -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/prof_${name}_kernel_stats.csv
tail -2 $RAW/err.log | cut -c1-300

#!/bin/bash
# Run on the GPU box (via gpurun): separate rocprofv3 --pmc passes (one counter each) of one python tool; prints the
# per-kernel sums of our kernels.   tools/pmc_cmd.sh <kernel-name-substring> "<C1 C2 ...>" <script.py> [args...]
pat=$1; shift
counters=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in $counters; do
  rm -rf /tmp/pmcc_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmcc_$c -- python3 "$@" > /dev/null 2> /tmp/pmcc_$c.err
  p=$(find /tmp/pmcc_$c -name '*counter_collection.csv' | head -1)
  if [ -n "$p" ]; then python3 - "$p" "$c" "$pat" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if sys.argv[3] in k:
        name = k.split("(")[0][-40:]
        acc[name][0] += float(r["Counter_Value"]); acc[name][1] += 1
for k, (v, n) in acc.items(): print(sys.argv[2], k, "sum", v, "launches", n, "per_launch", v / n)
PY
  else echo "$c: no data"; tail -2 /tmp/pmcc_$c.err | cut -c1-200; fi
done

#!/bin/bash
# MFMA-pipe utilisation of the scan kernels: separate --pmc passes (one counter each) over a short bench run.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-video --dht 3,6 > /dev/null 2> /tmp/pmc_$c.err
  p=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  if [ -n "$p" ]; then python3 - "$p" "$c" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_hamm64_mfma" in k:
        name = "mfma3" if "mfma3" in k else "mfma_pre"
        acc[name][0] += float(r["Counter_Value"]); acc[name][1] += 1
for k, (v, n) in acc.items(): print(sys.argv[2], k, "sum", v, "rows", n)
PY
  else echo "$c: no data"; tail -2 /tmp/pmc_$c.err | cut -c1-200; fi
done

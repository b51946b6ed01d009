#!/bin/bash
# Matrix-pipe counters of the 256-bit scan kernels (tools/knn_only.py 100000 2: one needle image = k_hamm256_small<16>,
# 64 needle images = k_hamm256_mfma3<12,2>), one counter per pass.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for c in SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS; do
  rm -rf /tmp/pk_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pk_$c -- python3 tools/knn_only.py 100000 2 > /dev/null 2> /tmp/pk_$c.err
  p=$(find /tmp/pk_$c -name '*counter_collection.csv' | head -1)
  [ -n "$p" ] && python3 tools/pmc_sum.py "$p" "$c" k_hamm256 || { echo "$c: no data"; tail -2 /tmp/pk_$c.err | cut -c1-200; }
done

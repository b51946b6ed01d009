#!/bin/bash
# Which kernels hash an image of a given geometry and what they issue (kernel trace + one counter per pass).
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
# usage: tools/pmc_geo.sh ["W H" ...]   (default: the >= 2 MP geometries of profiles/r04_pmc_geo.md)
[ $# -eq 0 ] && set -- "3840 2160" "4000 3000" "1920 1080"
for geo in "$@"; do
  set -- $geo
  echo "== $1 x $2"
  rm -rf /tmp/kt_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$1 -- python3 tools/hash_geo_only.py $1 $2 2> /tmp/kt_$1.err | tail -1
  f=$(find /tmp/kt_$1 -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && grep -E "cbh" "$f" | cut -d, -f1-4 | sed 's/(anonymous namespace):://; s/void cbh:://; s/cbh:://' | cut -c1-140
  for c in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE; do
    rm -rf /tmp/pg_$c
    rocprofv3 --pmc $c --output-format csv -d /tmp/pg_$c -- python3 tools/hash_geo_only.py $1 $2 > /dev/null 2> /tmp/pg_$c.err
    p=$(find /tmp/pg_$c -name '*counter_collection.csv' | head -1)
    [ -n "$p" ] && python3 tools/pmc_sum.py "$p" "$c" k_blur k_area k_tile k_dcthash k_band_area || { echo "$c: no data"; tail -1 /tmp/pg_$c.err | cut -c1-160; }
  done
done

#!/bin/bash
# Matrix-pipe / VALU counters of the shipped 64-bit scan kernels, one counter per pass.
#   run A: bench.py --dht 3,7  (one PRE launch with rare candidates, one FULL3 launch; 10^12 pairs each)
#   run B: bench.py --dht 6    (one PRE launch at its last threshold: a candidate in every third group)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
C="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"
for run in "A 3,7" "B 6"; do
  set -- $run
  echo "== run $1: bench.py --dht $2"
  for c in $C; do
    rm -rf /tmp/p_$1_$c
    rocprofv3 --pmc $c --output-format csv -d /tmp/p_$1_$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-video --no-orb --no-features --no-sharded-leg --dht $2 > /dev/null 2> /tmp/p_$1_$c.err
    p=$(find /tmp/p_$1_$c -name '*counter_collection.csv' | head -1)
    [ -n "$p" ] && python3 tools/pmc_sum.py "$p" "$c" k_hamm64_mfma k_dcthash_256 || { echo "$c: no data"; tail -2 /tmp/p_$1_$c.err | cut -c1-200; }
  done
done

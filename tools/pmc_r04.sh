#!/bin/bash
# Round 4: matrix-pipe counters of the four shipped MFMA scan kernels + the new hash kernel, one counter per pass.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
C="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
echo "== hamm64 (bench --dht 3,6: one PRE launch + one FULL3 launch at 1M x 1M)"
for c in $C; do
  rm -rf /tmp/p1_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/p1_$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-video --no-orb --no-features --no-sharded-leg --dht 3,6 > /dev/null 2> /tmp/p1_$c.err
  p=$(find /tmp/p1_$c -name '*counter_collection.csv' | head -1)
  [ -n "$p" ] && python3 tools/pmc_sum.py "$p" "$c" k_hamm64_mfma k_dcthash_256 || { echo "$c: no data"; tail -2 /tmp/p1_$c.err | cut -c1-200; }
done
echo "== hamm256 (tools/knn_only.py 100000 2)"
for c in $C; do
  rm -rf /tmp/p2_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/p2_$c -- python3 tools/knn_only.py 100000 2 > /dev/null 2> /tmp/p2_$c.err
  p=$(find /tmp/p2_$c -name '*counter_collection.csv' | head -1)
  [ -n "$p" ] && python3 tools/pmc_sum.py "$p" "$c" k_hamm256 || { echo "$c: no data"; tail -2 /tmp/p2_$c.err | cut -c1-200; }
done

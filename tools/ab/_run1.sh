cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 tools/hash_geo_only.py 854 480 2>/dev/null | tail -1
f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); grep -E "cbh" "$f" | cut -d, -f1-4 | sed 's/(anonymous namespace):://; s/void cbh:://' | cut -c1-100

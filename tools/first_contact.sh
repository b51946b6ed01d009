#!/bin/bash
# First contact with a multi-GPU node (see tools/first_contact.py): one JSON object with pass/fail per leg.
#   tools/first_contact.sh [--out profiles/first_contact.json]
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
python3 -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1 || { echo '{"ok": false, "error": "build() failed"}'; exit 1; }
python3 tools/first_contact.py "$@"

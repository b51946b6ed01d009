#!/bin/bash
# Same-box A/B of BUILDS of the library on dctHash64 by geometry: three alternating rounds of tools/hash_sizes.py (8 GB
# batches) per library; "cur" = cbird_amd/libcbird_hip.so, any other name N = cbird_amd/libcbird_hip.so.N (a copy of an
# earlier build: it travels with the gpurun snapshot and is git-ignored).  Pipe the output through `grep "GB/s\|^lib"`.
#   bash tools/ab/lib_ab_hash.sh 400x300,533x400,256x256 base cur
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}" || exit 1
G=$1; shift
for r in 1 2 3; do
  for l in "$@"; do
    echo lib $l
    if [ "$l" = cur ]; then unset CBH_LIB_PATH; else export CBH_LIB_PATH=$PWD/cbird_amd/libcbird_hip.so.$l; fi
    python tools/hash_sizes.py bytes=8e9 geos=$G ab=hash_band_area:1
  done
done

G=$1; shift
for r in 1 2 3; do for l in "$@"; do echo lib $l; if [ $l = cur ]; then unset CBH_LIB_PATH; else export CBH_LIB_PATH=$PWD/cbird_amd/libcbird_hip.so.$l; fi; python tools/hash_sizes.py bytes=8e9 geos=$G ab=hash_band_area:1; done; done

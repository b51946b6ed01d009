#!/bin/bash
# Run on the GPU box (via gpurun): the round's closing measurements on the last build -- the default bench line, the
# kernel trace + PMC passes of bench.py (tools/profile_bench.sh), dctHash64 by geometry with k_band_area off / on in large
# and small batches.  Outputs under gpurun_out/; tools/collect_profiles.py $R turns prof_summary/ into profiles/${R}_*.
R=${ROUND:-r06}
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out
python3 bench.py > gpurun_out/${R}_bench_1gpu.json 2> gpurun_out/${R}_bench_1gpu.err; tail -c 400 gpurun_out/${R}_bench_1gpu.json | head -c 400; echo
GEOS=64x64,128x128,160x120,200x150,300x200,320x240,400x300,480x360,533x400,600x400,640x480,720x540,800x600,854x480,900x600,960x540,1024x768,1280x720,1366x768,1600x900,1920x1080,2560x1440,3840x2160,4000x3000,256x256,512x512
python3 tools/hash_sizes.py bytes=8e9 geos=$GEOS ab=hash_band_area:0:1 2>/dev/null > gpurun_out/${R}_hash_sizes_8gb.txt
python3 tools/hash_sizes.py bytes=2e8 geos=$GEOS ab=hash_band_area:0:1 2>/dev/null > gpurun_out/${R}_hash_sizes_200mb.txt
tail -3 gpurun_out/${R}_hash_sizes_8gb.txt
bash tools/profile_bench.sh > gpurun_out/${R}_profile_bench.log 2>&1; tail -3 gpurun_out/${R}_profile_bench.log

#!/bin/bash
# Run on the GPU box (via gpurun): the whole `-m gpu` suite, the hash fuzzers in large and small batches, the scan soaks and
# the leak soak in one call.  Outputs under gpurun_out/soak/.
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out/soak
python -m pytest tests -m gpu -x -q > gpurun_out/soak/pytest_gpu.log 2>&1; tail -3 gpurun_out/soak/pytest_gpu.log | grep -E "passed|failed|error" 
python tools/fuzz_hash_sizes.py --cases 400 --seed 3031 > gpurun_out/soak/fuzz_hash_sizes.json 2> gpurun_out/soak/fuzz_hash_sizes.err; tail -c 300 gpurun_out/soak/fuzz_hash_sizes.json
python tools/fuzz_hash_sizes.py --cases 300 --seed 3032 --pixels 6000000 > gpurun_out/soak/fuzz_hash_sizes_small.json 2>> gpurun_out/soak/fuzz_hash_sizes.err; tail -c 200 gpurun_out/soak/fuzz_hash_sizes_small.json
python tools/fuzz_hash.py --cases 1500 --seed 3033 > gpurun_out/soak/fuzz_hash.json 2> gpurun_out/soak/fuzz_hash.err; tail -c 200 gpurun_out/soak/fuzz_hash.json
python tools/soak_scan.py 300000 60 > gpurun_out/soak/soak_scan.txt 2>&1; tail -2 gpurun_out/soak/soak_scan.txt
python tools/soak_sweep.py > gpurun_out/soak/soak_sweep.txt 2>&1; tail -2 gpurun_out/soak/soak_sweep.txt
python tools/leak_soak.py > gpurun_out/soak/leak_soak.txt 2>&1; tail -2 gpurun_out/soak/leak_soak.txt

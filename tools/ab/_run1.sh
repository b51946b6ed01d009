python bench.py --steps 10 --warmup 3 > gpurun_out/bench_r05_b.json 2> gpurun_out/bench_r05_b.err; tail -c 300 gpurun_out/bench_r05_b.err
python - <<EOF
import json
d=json.loads([l for l in open("gpurun_out/bench_r05_b.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["matches_expected"], d["full_identity"]["equal"], d["hash_identity"]["equal"])
print([(s["dht"], s["scan_kernel_ms"]) for s in d["dht_sweep"]])
print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline_full3"]["frac"], d["roofline_hash"]["frac"], d["roofline_hash"]["avg_launch_ms"])
EOF

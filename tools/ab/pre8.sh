export CBH_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 2954$i bench.py --gpus 8 --images 80001 --steps 1 --warmup 1 --no-cpu-baseline --no-features --no-sharded-leg --dht 2,5,8 --orb-images 4001 --video-clips 2000 > gpurun_out/pre8_$i.out 2> gpurun_out/pre8_$i.err
echo "run $i rc=$? stdout_bytes=$(wc -c < gpurun_out/pre8_$i.out) lines=$(wc -l < gpurun_out/pre8_$i.out) json_lines=$(grep -c '^{' gpurun_out/pre8_$i.out)"
head -c 300 gpurun_out/pre8_$i.out; echo; tail -5 gpurun_out/pre8_$i.err | cut -c1-300
done

mkdir -p gpurun_out/r2
export PANDORA_BENCH_BACKEND=gloo
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 100 --warmup 5 2> gpurun_out/r2/two_rank.err | tail -1 > gpurun_out/r2/two_rank.json
python tools/show_bench.py gpurun_out/r2/two_rank.json

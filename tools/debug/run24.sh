cd "$GRAFT_REPO_ROOT"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
ULTRA_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 5 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err; tail -c 1500 gpurun_out/bench_2rank.json; tail -3 gpurun_out/bench_2rank.err

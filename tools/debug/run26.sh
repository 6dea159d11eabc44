cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(command -v python3)")
python tools/pretrain_bench.py --steps 12 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pretrain -o pre -- "$PY" tools/pretrain_bench.py --steps 12 > gpurun_out/pretrain.log 2>&1
find gpurun_out/pretrain -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_pretrain_kernel_stats.csv \;
rm -rf gpurun_out/pretrain
tail -1 gpurun_out/pretrain.log

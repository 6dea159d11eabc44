cd "$GRAFT_REPO_ROOT"
bash tools/refresh_profiles.sh r02 > gpurun_out/refresh.log 2>&1
python tools/run_configs.py --with-stress > gpurun_out/r02_configs.md 2> gpurun_out/configs.err; tail -5 gpurun_out/r02_configs.md
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(command -v python3)")
for W in wn18rr fb15k237; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/train_$W -o train -- "$PY" tools/train_bench.py --workload S-$W --steps 20 > gpurun_out/train_$W.log 2>&1
  find gpurun_out/train_$W -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_train_${W}_kernel_stats.csv \;
  tail -2 gpurun_out/train_$W.log
  python tools/train_bench.py --workload S-$W --steps 30 --graphed 2>&1 | tail -1
done
rm -rf gpurun_out/train_wn18rr gpurun_out/train_fb15k237 gpurun_out/refresh/bench

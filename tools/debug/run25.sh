cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/bench_r2e.json 2> gpurun_out/bench_r2e.err; tail -c 400 gpurun_out/bench_r2e.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r2e.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["metrics_hip"], d["metrics_hip_after_finetune"], d["roofline_hbm"]["frac"], d["composition"]["finetune_ms_per_step"])
PY

"""One evaluation batch of the headline workload replayed as a hipGraph N times: for a kernel trace of the step alone
(rocprofv3 --kernel-trace -- python3 tools/debug/step_trace.py), then tools/debug/step_trace_report.py on the csv."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from ultra_torchdrug_amd.engine import GraphedPredict
from ultra_torchdrug_amd.data import DEFAULT_SEED

dev = torch.device("cuda:0")
task, triples, fact_mask, n_fact = bench.transductive_task(os.environ.get("WORKLOAD", "S-fb15k237"), dev, 2048, DEFAULT_SEED)
bench.prepare_plans(task)
test = torch.from_numpy(triples[n_fact:]).to(dev)
with torch.no_grad():
    task.predict(test[:16])
    g = GraphedPredict(task, test[:16], warmup=0)
    for i in range(10):
        g(test[16 * i:16 * i + 16])
    torch.cuda.synchronize()
    marker = torch.zeros(1, device=dev)
    marker.add_(1.0)                      # a recognisable kernel in front of the timed replays
    torch.cuda.synchronize()
    for i in range(20):
        g(test[16 * (i % 100):16 * (i % 100) + 16])
    torch.cuda.synchronize()

"""Diagnosis: one GraphedTrainStep with a GradientReducer (1-rank RCCL group) against the eager step, per-parameter."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import copy
import torch
import torch.distributed as dist
from ultra_torchdrug_amd import engine
from ultra_torchdrug_amd.data import synthetic_triples
from ultra_torchdrug_amd.graph import Graph
from ultra_torchdrug_amd.task import build_ultra

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29544", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")


def build():
    triples, n, r = synthetic_triples("S-tiny", 1024)
    torch.manual_seed(1024)
    task = build_ultra(r)
    task.preprocess(Graph(torch.from_numpy(triples), num_node=n, num_relation=r))
    task.num_negative = 16
    return task.to(dev).train(), triples


task, triples = build()
state = copy.deepcopy(task.state_dict())
batches = [torch.from_numpy(triples[i:i + 8]).to(dev) for i in (0, 8, 16)]
for mode in ("none", "in_graph", "after_replay"):
    twin, _ = build()
    twin.load_state_dict(state)
    task.load_state_dict(state)
    opt_g = torch.optim.AdamW(twin.parameters(), lr=1e-3)
    reducer = None if mode == "none" else engine.GradientReducer(twin, overlap=True, single_rank=True)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        step = engine.GraphedTrainStep(twin, opt_g, batches[0], reducer=reducer, reduce_in_graph=(mode == "in_graph"))
    print(mode, "reduce_in_graph =", step.reduce_in_graph, "| warnings:", [str(w.message)[:160] for w in caught if "GraphedTrainStep" in str(w.message)])
    opt_e = torch.optim.AdamW(task.parameters(), lr=1e-3)
    for i, b in enumerate(batches):
        lg = step(b)[0].item()
        g_grads = {k: p.grad.detach().clone() for k, p in twin.named_parameters() if p.grad is not None}
        task._static_negative = step.last_negatives.clone()
        le = engine.train_step(task, opt_e, b)[0].item()
        task._static_negative = None
        e_grads = {k: p.grad.detach().clone() for k, p in task.named_parameters() if p.grad is not None}
        bad = [(k, float((g_grads[k] - e_grads[k]).abs().max())) for k in e_grads if k in g_grads and not torch.equal(g_grads[k], e_grads[k])]
        missing = [k for k in e_grads if k not in g_grads] + [k for k in g_grads if k not in e_grads]
        pbad = [k for (k, a), (_, c) in zip(task.named_parameters(), twin.named_parameters()) if not torch.equal(a, c)]
        print("  step", i, "loss graphed %.6f eager %.6f" % (lg, le), "| grads differing:", bad[:4], len(bad), "| only on one side:", missing[:4],
              "| params differing after the step:", len(pbad), pbad[:3])
    if reducer is not None:
        print("  launched_from_hooks", reducer.launched_from_hooks, "of", len(reducer.buckets))
        reducer.remove_hooks()
dist.destroy_process_group()

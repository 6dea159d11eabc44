"""Child process of tests/test_model_gpu.py::test_captured_collectives_in_a_child_process (not a test module itself).

Builds a one-rank `nccl` (= RCCL) group on cuda:0, captures a training step with the bucket all-reduces INSIDE the hipGraph
(engine.GraphedTrainStep(reduce_in_graph=True)), runs three replays against the same three eager steps and prints one JSON
line.  On runtimes where the replay of a captured collective aborts, this process dies instead of the test session."""
import copy
import json
import os
import sys
import warnings

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build():
    from ultra_torchdrug_amd.data import synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.task import build_ultra
    triples, n, r = synthetic_triples("S-tiny", 1024)
    graph = Graph(torch.from_numpy(triples), num_node=n, num_relation=r)
    torch.manual_seed(1024)
    task = build_ultra(r)
    task.preprocess(graph)
    task.num_negative = 16
    return task, triples


def main():
    from ultra_torchdrug_amd import engine
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29537", rank=0, world_size=1, device_id=dev)
    task, triples = build()
    task.to(dev).train()
    state = copy.deepcopy(task.state_dict())
    twin, _ = build()
    twin.to(dev).train()
    twin.load_state_dict(state)
    batches = [torch.from_numpy(triples[i:i + 8]).to(dev) for i in (0, 8, 16)]
    opt_g = torch.optim.AdamW(twin.parameters(), lr=1e-3)
    reducer = engine.GradientReducer(twin, overlap=True, single_rank=True)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        step = engine.GraphedTrainStep(twin, opt_g, batches[0], reducer=reducer, reduce_in_graph=True)
    fell_back = any("reduced outside the graphs" in str(w.message) for w in caught)
    losses_g, negs = [], []
    for b in batches:
        losses_g.append(step(b)[0].item())
        negs.append(step.last_negatives.clone())
    torch.cuda.synchronize()
    opt_e = torch.optim.AdamW(task.parameters(), lr=1e-3)
    losses_e = []
    for b, neg in zip(batches, negs):
        task._static_negative = neg
        losses_e.append(engine.train_step(task, opt_e, b)[0].item())
    task._static_negative = None
    torch.cuda.synchronize()
    equal = all(torch.equal(a, b) for a, b in zip(task.parameters(), twin.parameters()))
    print(json.dumps({"in_graph": bool(step.reduce_in_graph), "fell_back": fell_back, "losses_equal": losses_g == losses_e,
                      "parameters_equal": equal, "launched_from_hooks": reducer.launched_from_hooks,
                      "buckets": len(reducer.buckets)}), flush=True)
    reducer.remove_hooks()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Child process of tests/test_configs_gpu.py::test_multi_graph_graphed_steps_on_two_ranks_that_draw_different_graphs (not a
test module itself): ONE rank of a two-rank job in which the ranks train on DIFFERENT graphs in most steps and meet each graph
for the first time on different steps (multi-graph pre-training, /root/reference/ultra/engine.py:23-34).

    python multigraph_ranks_child.py RANK WORLD PORT BACKEND

BACKEND gloo: both ranks share cuda:0 (a one-GPU box; gloo reduces CUDA tensors through the host); nccl: one GPU per rank.
Runs the same hop sequence twice -- engine.GraphedMultiGraphTrainStep with a GradientReducer, then eager engine.train_step
with a reducer on the negatives the replays drew -- and prints one JSON line: bucket all-reduces per step, whether graphed and
eager parameters are equal, and a checksum of the final parameters (the parent compares the ranks')."""
import copy
import json
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

# rank 0 and rank 1 meet each graph for the first time on different steps, and share a graph on one step only
ORDERS = (["fb15k237", "fb15k237", "wn18rr", "codexm", "wn18rr", "fb15k237"],
          ["codexm", "wn18rr", "wn18rr", "fb15k237", "codexm", "fb15k237"])


def main():
    rank, world, port, backend = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    kwargs = dict(device_id=dev) if backend == "nccl" else {}
    dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world, **kwargs)
    from test_configs_gpu import _multi_graph_task
    from ultra_torchdrug_amd import engine
    B = 16
    task = _multi_graph_task(scale=8, num_negative=32).to(dev).train()          # same seed: same weights on both ranks
    state = copy.deepcopy(task.state_dict())
    twin = _multi_graph_task(scale=8, num_negative=32).to(dev).train()
    twin.load_state_dict(state)
    gen = torch.Generator().manual_seed(100 + rank)                              # per-rank draws (script/run_full.py:102-107)
    batches = []
    for gid in ORDERS[rank % len(ORDERS)]:
        fact = task.contexts[gid]["fact_graph"].edge_list
        batches.append((fact[torch.randperm(len(fact), generator=gen)[:B].to(dev)], gid))

    opt_g = torch.optim.AdamW(twin.parameters(), lr=1e-3)
    reducer_g = engine.GradientReducer(twin, overlap=True)
    graphed = engine.GraphedMultiGraphTrainStep(twin, opt_g, B, reducer=reducer_g)
    after_init = reducer_g.total_launched
    per_step, negatives, losses_g, sums, said = [], [], [], [], []
    for batch in batches:
        before, sent_before = reducer_g.total_launched, reducer_g.collectives
        loss, _ = graphed(batch)
        per_step.append(reducer_g.total_launched - before)
        # what crossed the phase boundaries of THIS step: the groups this rank sent, in order, and -- after the round -- the flat
        # buffer every rank's optimizer read: averaged gradients, so bit-identical on all ranks whatever graph each rank was on
        # (a phase of an FB15k237-shaped step is several times longer than a WN18RR-shaped one: if a group of one rank ever paired
        # with another group of the other, sizes or contents would differ here)
        torch.cuda.synchronize()
        said.append(reducer_g.sent[-(reducer_g.collectives - sent_before):])
        flat = reducer_g._flat_all
        sums.append([float(flat.double().sum()), float(flat.double().abs().sum()), int(flat.view(torch.int32).long().sum())])
        negatives.append(graphed.steps[batch[1]].last_negatives.clone())
        losses_g.append(float(loss))
    torch.cuda.synchronize()

    opt_e = torch.optim.AdamW(task.parameters(), lr=1e-3)
    reducer_e = engine.GradientReducer(task, overlap=True)
    losses_e = []
    for batch, neg in zip(batches, negatives):
        task._static_negative = neg
        loss, _ = engine.train_step(task, opt_e, batch, reducer=reducer_e)
        losses_e.append(float(loss))
    task._static_negative = None
    torch.cuda.synchronize()
    equal = all(torch.equal(a, b) for a, b in zip(task.parameters(), twin.parameters()))
    flat = torch.cat([p.detach().reshape(-1).double() for p in twin.parameters()])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    # what ran the collectives, and on how many DIFFERENT devices (the parent asserts nccl + distinct devices on a multi-GPU node)
    props = torch.cuda.get_device_properties(dev)
    ident = torch.tensor([(int(getattr(props, "pci_domain_id", 0)) << 32) | (int(getattr(props, "pci_bus_id", 0)) << 16)
                          | int(getattr(props, "pci_device_id", dev.index or 0))], dtype=torch.int64, device=dev)
    idents = [torch.zeros_like(ident) for _ in range(world)]
    dist.all_gather(idents, ident)
    mine = torch.tensor(sums, dtype=torch.float64, device=dev)
    theirs = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(theirs, mine)
    print(json.dumps({"rank": rank, "groups_sent_per_step": said, "groups": [list(g) for g in reducer_g.groups],
                      "reduced_buffers_equal_on_all_ranks_per_step": [bool(all(torch.equal(theirs[0][i], t[i]) for t in theirs[1:]))
                                                                      for i in range(len(sums))],
                      "reduced_buffer_nonzero_per_step": [s[1] > 0 for s in sums],
                      "backend": dist.get_backend(), "distinct_devices": len({int(t.item()) for t in idents}),
                      "modes": graphed.modes, "buckets": len(reducer_g.buckets), "warm_launches": after_init, "per_step": per_step,
                      "captured": sorted(graphed.steps), "graphed_equals_eager": equal, "losses_equal": losses_g == losses_e,
                      "ranks_hold_equal_parameters": bool(all(torch.equal(both[0], b) for b in both[1:])),
                      "finite": bool(torch.isfinite(flat).all())}), flush=True)
    reducer_g.remove_hooks()
    reducer_e.remove_hooks()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Fine-tuning step time (config 3 of BASELINE.json: WN18RR-shaped, fp32, rspmm fwd+bwd through autograd).

    python tools/train_bench.py [--workload S-wn18rr] [--batch 16] [--steps 10] [--rebuild]

--rebuild drops the batch's positive edges the reference's way (new graph + re-sort every step) instead of
zero-weighting them on the cached plans.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="S-wn18rr")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rebuild", action="store_true")
    ap.add_argument("--graphed", action="store_true", help="forward + backward replayed as one hipGraph (engine.GraphedTrainStep)")
    args = ap.parse_args()
    from ultra_torchdrug_amd import engine
    from ultra_torchdrug_amd.data import synthetic_triples, DEFAULT_SEED
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.task import build_ultra
    dev = torch.device("cuda:0")
    triples, n, r = synthetic_triples(args.workload, DEFAULT_SEED)
    torch.manual_seed(DEFAULT_SEED)
    task = build_ultra(r)
    task.preprocess(Graph(torch.from_numpy(triples), num_node=n, num_relation=r))
    task.to(dev).train()
    if args.rebuild:
        task.model._removal_by_zero_weight = lambda: False
    opt = torch.optim.AdamW(task.parameters(), lr=5e-4, fused=True)       # (bench.make_optimizer)
    data = torch.from_numpy(triples).to(dev)
    rng = np.random.default_rng(0)
    losses = []
    graphed = None
    if args.graphed:
        idx = torch.from_numpy(rng.choice(len(triples), args.batch, replace=False)).to(dev)
        graphed = engine.GraphedTrainStep(task, opt, data[idx])
    for i in range(3 + args.steps):
        if i == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        idx = torch.from_numpy(rng.choice(len(triples), args.batch, replace=False)).to(dev)
        loss, _ = graphed(data[idx]) if graphed is not None else engine.train_step(task, opt, data[idx])
        loss = loss.clone()
        losses.append(loss)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    und = task.model._undirected(task.fact_graph)
    E = und.relcsr.n_edges
    E_rel = task.rel_graphs[0].relcsr.n_edges
    msgs = (6 * E + 6 * E_rel) * args.batch * 3          # forward + d_input + d_relation
    print("%s B=%d %s: %.2f ms/step, %.2e edge messages/s (fwd+bwd), loss %.4f -> %.4f"
          % (args.workload, args.batch, ("rebuild" if args.rebuild else "zero-weight") + (", hipGraph" if args.graphed else ""), ms, msgs / (ms * 1e-3),
             losses[0].item(), losses[-1].item()))


if __name__ == "__main__":
    main()

"""Multi-graph pre-training step (config 4 of BASELINE.json: pretrain_3g-shaped, batch 64 per GPU).

    python tools/pretrain_bench.py [--steps 10] [--batch 64]
    python -m torch.distributed.run --nproc-per-node N ... tools/pretrain_bench.py      (one rank per GPU, RCCL)

Three seeded synthetic graphs of the sizes of FB15k237, WN18RR and CoDEx-M under ONE set of weights; every step each
rank draws a graph (probability ~ #fact edges) and 64 of its fact edges (ultra/engine.py:23-34), runs the fine-tuning
step on it and all-reduces the flat gradient buffer (engine.allreduce_gradients).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    from ultra_torchdrug_amd import engine
    from ultra_torchdrug_amd.data import synthetic_kg, DEFAULT_SEED
    from ultra_torchdrug_amd.task import build_ultra
    rank, world = engine.init_distributed()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    torch.manual_seed(DEFAULT_SEED)
    task = build_ultra(237)
    for i, name in enumerate(["S-fb15k237", "S-wn18rr", "S-codexm"]):
        task.add_context(str(i), synthetic_kg(name))
    task.to(dev).train()
    opt = torch.optim.AdamW(task.parameters(), lr=5e-4, fused=True)       # (bench.make_optimizer)
    gen = torch.Generator().manual_seed(DEFAULT_SEED + rank)          # seed + rank, script/run_full.py:102-107
    msgs, seen = 0, []
    for step in range(3 + args.steps):
        if step == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            msgs = 0
        batch, gid = engine.sample_edges_from_graph(task, args.batch, gen)
        loss, _ = engine.train_step(task, opt, (batch, gid))
        ctx = task.contexts[gid]
        und = task.model._undirected(ctx["fact_graph"])
        msgs += (6 * und.relcsr.n_edges + 6 * ctx["rel_graphs"][0].relcsr.n_edges) * args.batch * 3
        seen.append(gid)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print("pretrain-3g-shaped, %d rank(s), batch %d/GPU: %.1f ms/step, %.2e edge messages/s per rank (fwd+bwd), "
              "graphs drawn %s, last loss %.4f" % (world, args.batch, 1e3 * dt / args.steps, msgs / dt,
                                                  "".join(seen[3:]), loss.item()))


if __name__ == "__main__":
    main()

"""Step time of the captured fine-tuning step with a gradient reducer on a ONE-rank RCCL group (the collectives are issued, the
reduction is the identity): mode `phased` (three graphs, bucket all-reduces between them on the side stream) against mode `after`
(one graph, buckets after the replay: round 3) and against no reducer at all.  VERDICT r3 item 3: "step time <= today's + 2 %".

    python tools/phased_step_time.py [--workload S-wn18rr] [--batch 16] [--steps 60]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="S-wn18rr")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=60)
    args = ap.parse_args()
    from ultra_torchdrug_amd import engine
    from ultra_torchdrug_amd.data import DEFAULT_SEED, synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.task import build_ultra
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29551", rank=0, world_size=1, device_id=dev)
    triples, n, r = synthetic_triples(args.workload, DEFAULT_SEED)
    data = torch.from_numpy(triples).to(dev)
    rng = np.random.default_rng(0)
    batches = [data[torch.from_numpy(rng.choice(len(triples), args.batch, replace=False)).to(dev)] for _ in range(args.steps + 5)]
    out = {}
    for mode in ("no reducer", "after", "phased"):
        torch.manual_seed(DEFAULT_SEED)
        task = build_ultra(r)
        task.preprocess(Graph(torch.from_numpy(triples), num_node=n, num_relation=r))
        task.to(dev).train()
        opt = torch.optim.AdamW(task.parameters(), lr=5e-4, fused=True)       # (bench.make_optimizer)
        reducer = None if mode == "no reducer" else engine.GradientReducer(task, overlap=True, single_rank=True)
        step = engine.GraphedTrainStep(task, opt, batches[0], reducer=reducer, phased=(mode == "phased"))
        assert step.mode == {"no reducer": "single", "after": "after", "phased": "phased"}[mode], step.mode
        for b in batches[:5]:
            step(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches[5:]:
            step(b)
        torch.cuda.synchronize()
        out[mode] = 1e3 * (time.perf_counter() - t0) / args.steps
        if reducer is not None:
            reducer.remove_hooks()
        del step, task, opt
        torch.cuda.empty_cache()
    print("%s B=%d, one-rank RCCL group, ms per step: %s" % (args.workload, args.batch,
                                                             ", ".join("%s %.3f" % kv for kv in out.items())))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

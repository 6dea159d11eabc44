"""Config 5 as BASELINE.json names it -- *inference* on S-stress (10 M nodes / 100 M edges / 1 k relations, 64d): the WHOLE
`predict` of the shipped 6 x 64d Ultra (relation stack + 6 entity layers + epilogues + score head) and the filtered ranking,
on a graph generated on the device (/root/reference/ultra/task.py:228-263, ultra/model.py:101-143,182-194).

    python tools/stress_predict.py [--batch 1] [--nodes N --triples T --relations R] [--reps 5] [--json PATH]

Prints one JSON object (stage times from stream events around each stage of one eager `predict`, whole-call wall time over
`--reps` calls, memory high-water mark).  The per-kernel table comes from running this under `rocprofv3 --kernel-trace --stats`.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def build_stress_task(dev, n_node, n_triple, n_rel, seed=1024, log=None):
    """ultra_torchdrug_amd.data.stress_task, timed."""
    from ultra_torchdrug_amd.data import stress_task
    t0 = time.perf_counter()
    task, gen = stress_task(dev, n_node, n_triple, n_rel, seed)
    torch.cuda.synchronize()
    if log is not None:
        log["task_build_s"] = time.perf_counter() - t0
        log["relation_graph_edges"] = int(task.rel_graphs[0].num_edge)
    return task, gen


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--triples", type=int, default=50_000_000)
    ap.add_argument("--relations", type=int, default=500)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    import ultra_torchdrug_amd as U
    U.require_library()
    dev = torch.device("cuda:0")
    out = {"workload": "S-stress N=%d triples=%d R=%d B=%d" % (args.nodes, args.triples, 2 * args.relations, args.batch)}
    task, gen = build_stress_task(dev, args.nodes, args.triples, args.relations, log=out)
    print("[stress_predict] task built: %s" % out, file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    und = task.model._undirected(task.fact_graph)
    csr = und.relcsr
    _ = csr.fwd
    _ = csr.frontier_index
    _ = task.rel_graphs[0].relcsr.fwd
    torch.cuda.synchronize()
    out["plan_build_s"] = time.perf_counter() - t0
    out["E"] = int(csr.n_edges)
    print("[stress_predict] plans built: %.1f s, E = %d" % (out["plan_build_s"], out["E"]), file=sys.stderr, flush=True)
    B = args.batch
    batch = torch.stack([torch.randint(0, args.nodes, (B,), device=dev, generator=gen),
                         torch.randint(0, args.nodes, (B,), device=dev, generator=gen),
                         torch.randint(0, args.relations, (B,), device=dev, generator=gen)], dim=1)
    with torch.no_grad():
        torch.cuda.reset_peak_memory_stats()
        pred = task.predict(batch)
        torch.cuda.synchronize()
        out["pred_shape"] = list(pred.shape)
        out["pred_finite"] = bool(torch.isfinite(pred).all())
        ranks = task.rank_batch(batch, pred)
        torch.cuda.synchronize()
        out["ranks"] = ranks.tolist()
        times = []
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pred = task.predict(batch)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ranks2 = task.rank_batch(batch, pred)
            torch.cuda.synchronize()
            times.append((1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)))
        assert torch.equal(ranks, ranks2)
        out["predict_ms"] = sorted(t[0] for t in times)[len(times) // 2]
        out["rank_ms"] = sorted(t[1] for t in times)[len(times) // 2]
        out["peak_memory_GB"] = torch.cuda.max_memory_allocated() / 1e9
        out["entity_edge_messages_per_s"] = 6 * out["E"] * 2 * B / (out["predict_ms"] * 1e-3)
    line = json.dumps(out)
    print(line)
    if args.json:
        with open(args.json, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()

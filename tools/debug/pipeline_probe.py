"""Do the relation stack of batch i+1 and the entity stack of batch i overlap when replayed on two streams?
(probe for a two-stage evaluation pipeline: times R then E on one stream vs R and E on two streams)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from ultra_torchdrug_amd.data import DEFAULT_SEED

dev = torch.device("cuda:0")
wl = os.environ.get("WORKLOAD", "S-fb15k237")
task, triples, fact_mask, n_fact = bench.transductive_task(wl, dev, 2048, DEFAULT_SEED)
bench.prepare_plans(task)
test = torch.from_numpy(triples[n_fact:]).to(dev)
batch = test[:16].clone()
model = task.model
model.check_indices = False
with torch.no_grad():
    for _ in range(2):
        rel = task.relation_representations(batch[:, 2])
        pred = model.score_both_sides(task.fact_graph, rel[0], batch)
    assert pred is not None
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    static_rel = rel[0].clone()
    gR, gE = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gR, stream=sa, capture_error_mode="thread_local"):
        out_rel = task.relation_representations(batch[:, 2])[0]
    with torch.cuda.graph(gE, stream=sb, capture_error_mode="thread_local"):
        out_pred = model.score_both_sides(task.fact_graph, static_rel, batch)
    torch.cuda.synchronize()

    def timed(fn, n=300):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def serial():
        with torch.cuda.stream(sa):
            gR.replay()
            gE.replay()

    def only_r():
        with torch.cuda.stream(sa):
            gR.replay()

    def only_e():
        with torch.cuda.stream(sb):
            gE.replay()

    def concurrent():
        with torch.cuda.stream(sa):
            gR.replay()
        with torch.cuda.stream(sb):
            gE.replay()

    for rnd in range(2):
        print("%s  R %.3f ms  E %.3f ms  serial %.3f ms  two streams %.3f ms" % (wl, timed(only_r), timed(only_e), timed(serial), timed(concurrent)), flush=True)

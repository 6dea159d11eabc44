"""Does WHEN a GraphedPredict is captured (what the allocator holds by then) change its replay time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from ultra_torchdrug_amd.engine import GraphedPredict
from ultra_torchdrug_amd.data import DEFAULT_SEED

dev = torch.device("cuda:0")
task, triples, fact_mask, n_fact = bench.transductive_task("S-fb15k237", dev, 2048, DEFAULT_SEED)
bench.prepare_plans(task)
test = torch.from_numpy(triples[n_fact:]).to(dev)
B = 16
nb = len(test) // B


def timed(g, n=300):
    for i in range(20):
        g(test[(i % nb) * B:(i % nb) * B + B])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        g(test[(i % nb) * B:(i % nb) * B + B])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    task.predict(test[:B])
    g1 = GraphedPredict(task, test[:B], warmup=0)
    print("g1 (first capture)            %.4f ms" % timed(g1), flush=True)
    g2 = GraphedPredict(task, test[:B], warmup=0)
    print("g2 (captured with g1 alive)   %.4f ms" % timed(g2), flush=True)
    print("g1 again                      %.4f ms" % timed(g1), flush=True)
    junk = [torch.empty(int(37e6) + 4096 * k, device=dev) for k in range(6)]
    g3 = GraphedPredict(task, test[:B], warmup=0)
    print("g3 (after odd allocations)    %.4f ms" % timed(g3), flush=True)
    del junk, g1, g2
    torch.cuda.empty_cache()
    g4 = GraphedPredict(task, test[:B], warmup=0)
    print("g4 (after empty_cache)        %.4f ms" % timed(g4), flush=True)
    print("g3 again                      %.4f ms" % timed(g3), flush=True)
    print("g4 with 50 replays            %.4f ms" % timed(g4, 50), flush=True)

"""Which stage of the fused inference sequence replays differently from eager execution?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from ultra_torchdrug_amd.data import DEFAULT_SEED

dev = torch.device("cuda:0")
task, triples, fact_mask, n_fact = bench.transductive_task("S-fb15k237", dev, 2048, DEFAULT_SEED)
bench.prepare_plans(task)
test = torch.from_numpy(triples[n_fact:]).to(dev)
B = 16
model = task.model
model.check_indices = False


def capture(fn, *static):
    with torch.no_grad():
        fn(*static)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = fn(*static)
    return g, out


with torch.no_grad():
    # stage 1: relation representations
    r_static = test[:B, 2].clone()
    g1, rel_out = capture(lambda r: task.relation_representations(r)[0], r_static)
    bad = 0
    for i in range(1, 9):
        r = test[16 * i:16 * i + 16, 2]
        r_static.copy_(r)
        g1.replay()
        torch.cuda.synchronize()
        bad += int(not torch.equal(rel_out, task.relation_representations(r)[0]))
    print("relation stack: %d of 8 replays differ from eager" % bad)
    # stage 2: entity stack on given relation representations
    b_static = test[:B].clone()
    rep_static = task.relation_representations(b_static[:, 2])[0].contiguous().clone()
    g2, score_out = capture(lambda b, rep: model.score_both_sides(task.fact_graph, rep, b), b_static, rep_static)
    bad = 0
    for i in range(1, 9):
        b = test[16 * i:16 * i + 16]
        rep = task.relation_representations(b[:, 2])[0].contiguous()
        b_static.copy_(b); rep_static.copy_(rep)
        g2.replay()
        torch.cuda.synchronize()
        bad += int(not torch.equal(score_out, model.score_both_sides(task.fact_graph, rep, b)))
    print("entity stack (score_both_sides): %d of 8 replays differ from eager" % bad)
    # stage 3: the whole predict
    p_static = test[:B].clone()
    g3, pred_out = capture(lambda b: task.predict(b), p_static)
    bad = 0
    for i in range(1, 9):
        b = test[16 * i:16 * i + 16]
        p_static.copy_(b)
        g3.replay()
        torch.cuda.synchronize()
        bad += int(not torch.equal(pred_out, task.predict(b)))
    print("predict: %d of 8 replays differ from eager" % bad)

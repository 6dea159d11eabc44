"""CPU, world_size = 2, gloo: the N > 1 path of the engine -- query sharding, the int64 rank all-gather, the flat
gradient all-reduce and the packed metric reduce.  The rspmm operator is played by the CPU oracle (test
infrastructure); what is under test is everything that crosses ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(seed=3):
    from ultra_torchdrug_amd.data import synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.task import build_ultra
    triples, n, r = synthetic_triples((120, 700, 4), seed)
    torch.manual_seed(seed)
    task = build_ultra(r, hidden_dims=(16,) * 2, input_dim=16, rel_hidden=16, rel_layers=2, num_negative=8)
    task.preprocess(Graph(torch.from_numpy(triples), num_node=n, num_relation=r))
    return task, torch.from_numpy(triples)


def _worker(rank, world, port, out_dir):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), OMP_NUM_THREADS="2")
    torch.set_num_threads(2)
    from oracle_ops import oracle_rspmm
    from ultra_torchdrug_amd import engine
    engine.init_distributed("gloo")
    assert engine.get_world_size() == world
    task, triples = _build()
    test = triples[:37]                                   # odd count: ranks get 19 and 18 queries
    with oracle_rspmm(0):
        task.eval()
        metric, ranking = engine.evaluate(task, test, batch_size=8)
        # training: each rank its own batch, then ONE flat all-reduce
        task.train()
        torch.manual_seed(100 + rank)
        batch = triples[40 + 8 * rank: 48 + 8 * rank]
        opt = torch.optim.AdamW(task.parameters(), lr=5e-4)
        loss, tmetric = engine.train_step(task, opt, batch)
    grads = torch.cat([p.grad.reshape(-1) for p in task.parameters() if p.grad is not None])
    gathered = [torch.zeros_like(grads) for _ in range(world)]
    dist.all_gather(gathered, grads)
    assert torch.equal(gathered[0], gathered[1])           # after the all-reduce every rank holds the same gradients
    unused = sorted(k for k, p in task.named_parameters() if p.grad is None)

    # three more steps, three ways: flat blocking all-reduce | bucketed reducer launched from the backward hooks
    # (overlap) | the same buckets launched after backward -- identical parameters afterwards, on both ranks
    finals = {}
    for mode in ("flat", "overlap", "after"):
        twin, _ = _build()                                 # same seed: same initial weights in every mode
        twin.train()
        opt_t = torch.optim.AdamW(twin.parameters(), lr=5e-4)
        reducer = None if mode == "flat" else engine.GradientReducer(twin, overlap=(mode == "overlap"))
        if reducer is not None:      # never-used parameters are excluded statically, everything else is in a bucket
            in_buckets = {n for b in reducer.buckets for n in b["names"]}
            assert in_buckets == {k for k, _ in twin.named_parameters()} - set(unused)
            assert [b["name"] for b in reducer.buckets][:2] == ["model.mlp", "model.layers.1"]
        with oracle_rspmm(0):
            for s in range(3):
                b = triples[100 + 16 * s + 8 * rank: 108 + 16 * s + 8 * rank]
                twin._static_negative = torch.randint(0, 120, (8, 8), generator=torch.Generator().manual_seed(7 * s + rank))
                engine.train_step(twin, opt_t, b, reducer=reducer)
                if reducer is not None and mode == "overlap":
                    assert reducer._launched == 0 and reducer._next == 0          # finish() reset the state
        finals[mode] = torch.cat([p.detach().reshape(-1) for p in twin.parameters()])
    for mode in ("overlap", "after"):
        diff = (finals[mode] - finals["flat"]).abs().max().item()
        assert torch.equal(finals[mode], finals["flat"]), (mode, diff)
    both = [torch.zeros_like(finals["flat"]) for _ in range(world)]
    dist.all_gather(both, finals["overlap"])
    assert torch.equal(both[0], both[1])
    # a backward under paused() fires no hook; reduce_all() then packs from the gradients of THAT backward (what a
    # graph replay followed by reduce_all() relies on), and a second backward without finish() is refused
    twin, _ = _build()
    twin.train()
    reducer = engine.GradientReducer(twin, overlap=True)
    with oracle_rspmm(0):
        twin._static_negative = torch.randint(0, 120, (8, 8), generator=torch.Generator().manual_seed(50 + rank))
        stale, _ = twin(triples[200 + 8 * rank: 208 + 8 * rank])
        twin.zero_grad()
        stale.backward()                                    # hooks live: every bucket is in flight ...
        assert reducer._from_hooks == len(reducer.buckets)
        reducer.finish()                                    # ... and retired
        assert reducer.launched_from_hooks == len(reducer.buckets)
        twin.zero_grad()
        fresh, _ = twin(triples[300 + 8 * rank: 308 + 8 * rank])
        with reducer.paused():
            fresh.backward()
        assert reducer._from_hooks == 0
        local = torch.cat([p.grad.reshape(-1).clone() for b in reducer.buckets for p in b["params"]])
        reducer.reduce_all()
        reduced = torch.cat([p.grad.reshape(-1) for b in reducer.buckets for p in b["params"]])
        both_local = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(both_local, local)
        assert torch.allclose(reduced, (both_local[0] + both_local[1]) / 2, rtol=0, atol=1e-7)     # the NEW gradients
        again, _ = twin(triples[300 + 8 * rank: 308 + 8 * rank])
        again.backward()
        more, _ = twin(triples[300 + 8 * rank: 308 + 8 * rank])
        with pytest.raises(RuntimeError, match="second backward"):
            more.backward()
        reducer.abandon()
        reducer.remove_hooks()

    # multi-graph evaluation (ultra/engine.py:100-159): per-graph loop, metrics averaged over the graphs
    multi, _ = _build()
    from ultra_torchdrug_amd.data import synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    other, n2, r2 = synthetic_triples((90, 500, 4), 11)
    multi.add_context("second", Graph(torch.from_numpy(other), num_node=n2, num_relation=r2))
    multi.eval()
    sets = {"default": triples[:21], "second": torch.from_numpy(other[:13])}
    with oracle_rspmm(0):
        mean, per_graph, rankings = engine.evaluate_all(multi, sets, batch_size=8)
    assert multi.split == "default" and rankings["default"].shape == (21, 2) and rankings["second"].shape == (13, 2)

    # multi-graph TRAINING with the ranks on different graphs in the same step (ultra/engine.py:23-34: every rank draws its
    # own graph): with a reducer every step sends one round of buckets in bucket order whatever graph a rank is on, so the
    # sequences of collectives agree and both ranks end with the same parameters; warm() is one more round, nothing else
    multi.train()
    opt_m = torch.optim.AdamW(multi.parameters(), lr=5e-4)
    reducer = engine.GradientReducer(multi, overlap=True)
    assert reducer.warm() == len(reducer.groups) == 3 and reducer.total_launched == len(reducer.buckets) and reducer.rounds == 0
    hops = (["default", "second", "second", "default"], ["second", "second", "default", "default"])[rank]
    pools = {"default": triples, "second": torch.from_numpy(other)}
    with oracle_rspmm(0):
        for s, gid in enumerate(hops):
            before = reducer.total_launched
            multi._static_negative = torch.randint(0, 90, (8, 8), generator=torch.Generator().manual_seed(31 * s + rank))
            engine.train_step(multi, opt_m, (pools[gid][50 * s + 8 * rank: 50 * s + 8 * rank + 8], gid), reducer=reducer)
            assert reducer.total_launched - before == len(reducer.buckets) and reducer.rounds == s + 1
            assert reducer.collectives == 3 * (s + 2)           # warm() + one all-reduce per GROUP and step
            # what this rank SAID in this step: groups 0, 1, 2 in order, each its whole slice of the flat buffer -- the same
            # sequence on the rank that is on the other graph (compared below), and the same reduced buffer on both
            assert [g for g, _ in reducer.sent[-3:]] == [0, 1, 2]
            assert [n for _, n in reducer.sent[-3:]] == [sum(reducer.buckets[b]["numel"] for b in grp) for grp in reducer.groups]
            said = torch.tensor(reducer.sent[-3:], dtype=torch.int64)
            both_said = [torch.zeros_like(said) for _ in range(world)]
            dist.all_gather(both_said, said)
            assert torch.equal(both_said[0], both_said[1])
            flat = reducer._flat_all.clone()
            both_flat = [torch.zeros_like(flat) for _ in range(world)]
            dist.all_gather(both_flat, flat)
            assert torch.equal(both_flat[0], both_flat[1]) and float(flat.abs().sum()) > 0
    multi._static_negative = None
    reducer.remove_hooks()
    hop_params = torch.cat([p.detach().reshape(-1) for p in multi.parameters()])
    hop_both = [torch.zeros_like(hop_params) for _ in range(world)]
    dist.all_gather(hop_both, hop_params)
    assert torch.equal(hop_both[0], hop_both[1]) and torch.isfinite(hop_params).all()
    torch.save(dict(ranking=ranking, mrr=metric["mrr"], loss=loss, tloss=tmetric["binary cross entropy"],
                    unused=unused, grads=grads, multi_mean=mean, multi_rankings=rankings,
                    multi_mrr={k: float(v["mrr"]) for k, v in per_graph.items()}),
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_evaluate_and_train_step(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, "rank%d.pt" % r)) for r in range(world))
    assert torch.equal(r0["ranking"], r1["ranking"]) and r0["ranking"].shape == (37, 2)

    # single-process reference of the same evaluation
    sys.path.insert(0, HERE)
    from oracle_ops import oracle_rspmm
    task, triples = _build()
    task.eval()
    with torch.no_grad(), oracle_rspmm(0):
        want = torch.cat([task.rank_batch(triples[:37][i:i + 8]) for i in range(0, 37, 8)])
    # batches are composed differently (strided shards), scores of a query do not depend on its batch mates
    assert torch.equal(r0["ranking"], want)
    assert abs(float(r0["mrr"]) - float((1.0 / want.float()).mean())) < 1e-6
    # parameters that never get a gradient are exactly the ones the reference needs find_unused_parameters for
    assert any(k.startswith("model.dist_embed") for k in r0["unused"])
    assert any(k.startswith("rel_models.0.model.mlp") for k in r0["unused"])
    assert torch.equal(r0["grads"], r1["grads"]) and torch.isfinite(r0["grads"]).all()
    assert abs(float(r0["tloss"]) - float(r1["tloss"])) < 1e-7     # packed metric reduce: same mean on both ranks
    # evaluate_all: both ranks hold every graph's full ranking; the mean is the unweighted mean over the graphs
    for name in ("default", "second"):
        assert torch.equal(r0["multi_rankings"][name], r1["multi_rankings"][name])
        want_mrr = float((1.0 / r0["multi_rankings"][name].float()).mean())
        assert abs(r0["multi_mrr"][name] - want_mrr) < 1e-6
    assert abs(r0["multi_mean"]["mrr"] - (r0["multi_mrr"]["default"] + r0["multi_mrr"]["second"]) / 2) < 1e-9
    assert r0["multi_mean"] == r1["multi_mean"]
    from ultra_torchdrug_amd.data import synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    other, n2, r2 = synthetic_triples((90, 500, 4), 11)
    task.add_context("second", Graph(torch.from_numpy(other), num_node=n2, num_relation=r2)).copy()
    with torch.no_grad(), oracle_rspmm(0):
        task.use("second")
        want2 = torch.cat([task.rank_batch(torch.from_numpy(other[:13])[i:i + 8]) for i in range(0, 13, 8)])
    assert torch.equal(r0["multi_rankings"]["second"], want2)


def test_gather_variable_single_process():
    from ultra_torchdrug_amd import engine
    x = torch.arange(10).view(5, 2)
    assert torch.equal(engine.gather_variable(x), x)
    assert torch.equal(engine.shard_indices(10, 1, 4), torch.tensor([1, 5, 9]))

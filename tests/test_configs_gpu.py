"""GPU parity at the sizes of BASELINE.json's configs 3, 4 and 5 (the ones no small-graph test reaches).

* config 5 -- S-stress AT SIZE (10 M nodes / 100 M edges / 1 k relations, 64d, B = 1): the oracle cannot run the whole
  graph in test time, so the bar is (i) equality with the oracle on a 50 000-row subset (row-subset CSR, sources
  re-labelled), forward and ``d_input``; (ii) ``d_relation`` of two relations against the oracle; (iii) the
  size-independent properties of the operator on the WHOLE output: run-to-run equality, exact homogeneity
  (``f(2x) == 2 f(x)``: scaling by two commutes with every fp32 rounding) and additivity within the summation bound.
* config 3 -- S-wn18rr (N = 40 943, E = 173 670, R = 22) with B = 16 (F = 1 024), DistMult and TransE messages:
  forward + backward EQUAL to the oracle on a 64-column tile (columns are independent, so a tile of the full-width
  launch must equal the oracle run on that tile alone).
* config 4 -- multi-graph pre-training (ultra/task.py:637-890, ultra/engine.py:23-34): one step per graph context
  drawn with ``engine.sample_edges_from_graph``, B = 64, loss and every parameter gradient against the same step
  with the CPU oracle as operator backend (three graphs with the relation vocabularies of FB15k237 / WN18RR /
  CoDEx-M at 1/8 of their entity and triple counts: the CPU side of the comparison runs autograd over
  ``(N, 64, 64)`` activations); and one step per context at FULL size on the HIP path alone (finite loss, every
  trainable parameter receives a gradient, the negatives are strict non-edges).
"""
import numpy as np
import pytest
import torch

from oracle_ops import oracle_rspmm

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


# ------------------------------------------------------------------------------------------------ config 5
def test_stress_graph_at_full_size(oracle):
    from ultra_torchdrug_amd import RelCSR, functional as UF
    dev = _dev()
    N, TRIPLES, R_BASE, F = 10_000_000, 50_000_000, 500, 64
    gen = torch.Generator(device=dev).manual_seed(1024)
    h = torch.randint(0, N, (TRIPLES,), device=dev, generator=gen)
    t = torch.randint(0, N, (TRIPLES,), device=dev, generator=gen)
    r = torch.randint(0, R_BASE, (TRIPLES,), device=dev, generator=gen)
    # rspmm sees the graph with inverse edges: E = 100 M, R = 1 000 (SURVEY.md 8d, S-stress)
    dst, src, rel = torch.cat([t, h]), torch.cat([h, t]), torch.cat([r, r + R_BASE])
    del h, t, r
    R = 2 * R_BASE
    csr = RelCSR(dst, src, rel, None, N, N, R)
    del dst, src, rel
    E = csr.n_edges
    assert E > 99_900_000 and csr.unit_weight                 # uniform triples: (almost) no duplicates to merge
    x = torch.randn(N, F, device=dev, generator=gen)
    relation = torch.randn(R, F, device=dev, generator=gen)

    out = UF.rspmm_forward(csr, relation, x, "add", "mul")
    again = UF.rspmm_forward(csr, relation, x, "add", "mul")
    assert torch.equal(out, again), "two launches on the same inputs differ"
    del again
    doubled = UF.rspmm_forward(csr, relation, 2 * x, "add", "mul")
    assert torch.equal(doubled, 2 * out), "f(2x) != 2 f(x)"
    del doubled
    y = torch.randn(N, F, device=dev, generator=gen)
    out_y = UF.rspmm_forward(csr, relation, y, "add", "mul")
    both = UF.rspmm_forward(csr, relation, x + y, "add", "mul")
    bound = UF.rspmm_forward(csr, relation.abs(), x.abs() + y.abs(), "add", "mul")
    assert ((both - (out + out_y)).abs() <= 2e-6 * bound + 1e-6).all(), "additivity beyond the summation bound"
    del out_y, both, bound, y

    # ---- (i) oracle on a row subset: forward
    rows = torch.arange(3_000_000, 3_050_000, device=dev)                     # 50 000 consecutive destination rows
    lo = int(torch.searchsorted(csr.dst, rows[0]))
    hi = int(torch.searchsorted(csr.dst, rows[-1] + 1))
    sub_dst, sub_src, sub_rel = csr.dst[lo:hi] - rows[0], csr.src[lo:hi], csr.rel_id[lo:hi]
    uniq, inverse = torch.unique(sub_src, return_inverse=True)
    csr_o = oracle.coalesce_csr(sub_dst.cpu().numpy(), inverse.cpu().numpy(), sub_rel.cpu().numpy(), None,
                                len(rows), len(uniq), R)
    want = oracle.rspmm_forward(csr_o, relation.cpu().numpy(), x[uniq].cpu().numpy(), "add", "mul", piece=csr.piece_len)
    assert np.array_equal(out[rows].cpu().numpy(), want), "forward differs from the oracle on the row subset"
    assert int(torch.bincount(sub_dst).max()) <= csr.piece_len               # no split rows: == the sequential order too
    for sum_op in ("min", "max"):
        got = UF.rspmm_forward(csr, relation, x, sum_op, "add")
        want = oracle.rspmm_forward(csr_o, relation.cpu().numpy(), x[uniq].cpu().numpy(), sum_op, "add")
        assert np.array_equal(got[rows].cpu().numpy(), want), sum_op
        del got

    # ---- (ii) the WHOLE output against the reference's own ATen definition (ultra/layer.py:249-255,275-285: gather, (*|+),
    # scatter), no oracle in between: sums within the bound of two summation orders, min / max EQUAL.
    # The definition's (E, F) messages are 25.6 GB each at this size: evaluated over four slices of the edge list.
    with torch.no_grad():
        want_sum, scale = torch.zeros(N, F, device=dev), torch.zeros(N, F, device=dev)
        fmax = torch.finfo(torch.float32).max
        want_min, want_max = torch.full((N, F), fmax, device=dev), torch.full((N, F), -fmax, device=dev)
        for e0 in range(0, E, 25_000_000):
            sl = slice(e0, min(e0 + 25_000_000, E))
            d_, s_, r_ = csr.dst[sl], csr.src[sl], csr.rel_id[sl]
            message = relation[r_] * x[s_]                                     # layer.py:249-255 (distmult), unit weights
            want_sum.index_add_(0, d_, message)                                # scatter_add (layer.py:275-276)
            scale.index_add_(0, d_, message.abs())
            message = relation[r_] + x[s_]                                     # transe, for the order-free aggregates
            index = d_.unsqueeze(-1).expand_as(message)
            want_min.scatter_reduce_(0, index, message, reduce="amin", include_self=True)
            want_max.scatter_reduce_(0, index, message, reduce="amax", include_self=True)
            del message, index
        assert ((out - want_sum).abs() <= 1e-5 * scale + 1e-6).all(), "sum differs from the ATen definition beyond the bound"
        del want_sum, scale
        assert torch.equal(UF.rspmm_forward(csr, relation, x, "min", "add"), want_min), "min differs from the ATen definition"
        assert torch.equal(UF.rspmm_forward(csr, relation, x, "max", "add"), want_max), "max differs from the ATen definition"
        del want_min, want_max

    # ---- backward at size: d_input on a subset of source rows, d_relation of two relations
    grad = torch.randn(N, F, device=dev, generator=gen)
    d_x, d_rel = UF.rspmm_backward(csr, relation, x, None, grad, "add", "mul")
    d_x2, d_rel2 = UF.rspmm_backward(csr, relation, x, None, grad, "add", "mul")
    assert torch.equal(d_x, d_x2) and torch.equal(d_rel, d_rel2)
    del d_x2, d_rel2
    by_src = csr.by_src                                                        # rows = source nodes, node_a = dst
    s_rows = torch.arange(7_000_000, 7_050_000, device=dev)
    row64 = by_src.row.long()
    lo = int(torch.searchsorted(row64, s_rows[0]))
    hi = int(torch.searchsorted(row64, s_rows[-1] + 1))
    e_src, e_dst, e_rel = row64[lo:hi] - s_rows[0], by_src.node_a[lo:hi].long(), by_src.rel[lo:hi].long()
    uniq_d, inv_d = torch.unique(e_dst, return_inverse=True)
    # the oracle's backward walks the FORWARD CSR (rows = destinations): hand it the sub-graph dst' -> src'
    csr_b = oracle.coalesce_csr(inv_d.cpu().numpy(), e_src.cpu().numpy(), e_rel.cpu().numpy(), None,
                                len(uniq_d), len(s_rows), R)
    x_sub = x[s_rows].cpu().numpy()
    out_sub = np.zeros((len(uniq_d), F), dtype=np.float32)                     # unused for sum = add
    want_d_rel_sub, want_d_x = oracle.rspmm_backward(csr_b, relation.cpu().numpy(), x_sub, out_sub,
                                                     grad[uniq_d].cpu().numpy(), "add", "mul", piece=csr.piece_len)
    assert np.array_equal(d_x[s_rows].cpu().numpy(), want_d_x), "d_input differs from the oracle on the row subset"
    for rid in (17, 983):
        sel = (csr.rel_id == rid).nonzero().flatten()                          # forward order = (dst, src) order
        e_d, e_s = csr.dst[sel], csr.src[sel]
        ud, idd = torch.unique(e_d, return_inverse=True)
        us, ids = torch.unique(e_s, return_inverse=True)
        csr_r = oracle.coalesce_csr(idd.cpu().numpy(), ids.cpu().numpy(), np.zeros(len(sel), dtype=np.int64), None,
                                    len(ud), len(us), 1)
        want_rel, _ = oracle.rspmm_backward(csr_r, relation[rid:rid + 1].cpu().numpy(), x[us].cpu().numpy(),
                                            np.zeros((len(ud), F), dtype=np.float32), grad[ud].cpu().numpy(),
                                            "add", "mul", piece=csr.piece_len)
        assert np.array_equal(d_rel[rid].cpu().numpy(), want_rel[0]), "d_relation[%d] differs from the oracle" % rid


# ------------------------------------------------------------------------------------------------ config 3
@pytest.mark.parametrize("mul", ["mul", "add"])
def test_wn18rr_shape_forward_backward_equal_oracle_on_a_column_tile(oracle, mul):
    from graphs import kg_graph
    from ultra_torchdrug_amd import RelCSR, functional as UF
    dev = _dev()
    n, triples, base_rel = 40943, 86835, 11
    g = kg_graph(1024, n, triples, base_rel)
    R, F = 2 * base_rel, 16 * 64
    t = lambda a: torch.from_numpy(a).to(dev)
    csr = RelCSR(t(g["dst"]), t(g["src"]), t(g["rel"]), None, n, n, R)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], None, n, n, R)
    assert csr.n_edges == csr_o.n_edges
    gen = torch.Generator(device=dev).manual_seed(3)
    relation = torch.randn(R, F, device=dev, generator=gen).requires_grad_()
    x = torch.randn(n, F, device=dev, generator=gen).requires_grad_()
    grad = torch.randn(n, F, device=dev, generator=gen)
    out = UF.generalized_rspmm(csr, relation, x, sum="add", mul=mul)
    out.backward(grad)
    for tile in (0, 9):
        cols = slice(64 * tile, 64 * tile + 64)
        rel_t = np.ascontiguousarray(relation.detach()[:, cols].cpu().numpy())
        x_t = np.ascontiguousarray(x.detach()[:, cols].cpu().numpy())
        g_t = np.ascontiguousarray(grad[:, cols].cpu().numpy())
        want = oracle.rspmm_forward(csr_o, rel_t, x_t, "add", mul, piece=csr.piece_len)
        assert np.array_equal(out.detach()[:, cols].cpu().numpy(), want)
        want_d_rel, want_d_x = oracle.rspmm_backward(csr_o, rel_t, x_t, want, g_t, "add", mul, piece=csr.piece_len)
        assert np.array_equal(x.grad[:, cols].cpu().numpy(), want_d_x)
        assert np.array_equal(relation.grad[:, cols].cpu().numpy(), want_d_rel)


# ------------------------------------------------------------------------------------------------ config 4
PRETRAIN_3G = {"fb15k237": (14541, 272115, 237), "wn18rr": (40943, 86835, 11), "codexm": (17050, 185584, 51)}


def _multi_graph_task(scale, seed=1024, **kwargs):
    from ultra_torchdrug_amd.data import synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.task import build_ultra
    torch.manual_seed(seed)
    task = build_ultra(max(v[2] for v in PRETRAIN_3G.values()), **kwargs)
    for i, (name, (n, triples, r)) in enumerate(PRETRAIN_3G.items()):
        tr, n_, r_ = synthetic_triples((n // scale, triples // scale, r), seed + i)
        task.add_context(name, Graph(torch.from_numpy(tr), num_node=n_, num_relation=r_))
    return task


def test_multi_graph_pretraining_step_matches_oracle_path():
    """pretrain_3g.yaml: batch 64 per GPU, 128 strict negatives, bce + self-adversarial weights; a graph id travels
    with every batch (task.py:722-731).  One step on EACH of the three contexts: HIP vs CPU-oracle backend."""
    from ultra_torchdrug_amd import engine
    task = _multi_graph_task(scale=8, num_negative=128)
    task.train()
    gen = torch.Generator().manual_seed(7)
    batches = {}
    for _ in range(200):                                       # sample_edges_from_graph picks the graph at random
        batch, gid = engine.sample_edges_from_graph(task, 64, gen)
        batches.setdefault(gid, batch)
        if len(batches) == len(PRETRAIN_3G):
            break
    assert set(batches) == set(PRETRAIN_3G), "the sampler never drew one of the graphs"

    def step(dev, gid, neg):
        task.to(dev)
        task.zero_grad()
        task._static_negative = neg.to(dev)                    # same negatives on both sides
        try:
            loss, _ = task((batches[gid].to(dev), gid))
            loss.backward()
        finally:
            task._static_negative = None
        return loss.item(), {k: p.grad.detach().cpu().clone() for k, p in task.named_parameters() if p.grad is not None}

    dev = _dev()
    for gid in PRETRAIN_3G:
        task.to(dev).use(gid)
        torch.manual_seed(5)
        neg = task._strict_negative(*batches[gid].to(dev).t()).cpu()             # drawn by the HIP sampler
        loss_gpu, grads_gpu = step(dev, gid, neg)
        with oracle_rspmm(None):
            loss_cpu, grads_cpu = step(torch.device("cpu"), gid, neg)
        assert abs(loss_cpu - loss_gpu) <= 1e-5 * max(1.0, abs(loss_cpu)), gid
        assert grads_cpu.keys() == grads_gpu.keys()
        for k in grads_cpu:
            scale = grads_cpu[k].abs().max().item() + 1e-8
            assert (grads_cpu[k] - grads_gpu[k]).abs().max().item() <= 2e-4 * scale + 1e-6, (gid, k)


def test_multi_graph_pretraining_step_at_full_size():
    """The three pre-training graphs at their real sizes, B = 64 per step (F = 4 096), HIP path: finite loss, every
    trainable parameter gets a gradient, negatives are strict (no negative completes its query in the fact graph)."""
    from ultra_torchdrug_amd import engine
    dev = _dev()
    task = _multi_graph_task(scale=1, num_negative=128).to(dev).train()
    opt = torch.optim.AdamW(task.parameters(), lr=5e-4)
    gen = torch.Generator().manual_seed(11)
    seen = set()
    for _ in range(12):
        batch, gid = engine.sample_edges_from_graph(task, 64, gen)
        if gid in seen:
            continue
        seen.add(gid)
        batch = batch.to(dev)
        task.use(gid)
        h, t, r = batch.t()
        neg = task._strict_negative(h, t, r)
        fact = task.fact_graph
        half = len(batch) // 2
        assert neg.shape == (64, 128) and int(neg.min()) >= 0 and int(neg.max()) < fact.num_node
        bad_t = fact.match(torch.stack([h[:half, None].expand(-1, 128), neg[:half], r[:half, None].expand(-1, 128)],
                                       dim=-1).flatten(0, 1))[1]
        bad_h = fact.match(torch.stack([neg[half:], t[half:, None].expand(-1, 128), r[half:, None].expand(-1, 128)],
                                       dim=-1).flatten(0, 1))[1]
        assert int(bad_t.sum()) == 0 and int(bad_h.sum()) == 0, "a sampled negative is a fact edge"
        loss, metric = engine.train_step(task, opt, (batch, gid))
        assert torch.isfinite(loss)
        missing = [k for k, p in task.named_parameters()
                   if p.grad is None and "dist_embed" not in k and "rel_models.0.model.mlp" not in k]
        assert not missing, missing
    assert seen == set(PRETRAIN_3G)


def test_multi_graph_graphed_steps_equal_eager_steps():
    """engine.GraphedMultiGraphTrainStep: one captured step per graph context over ONE set of parameters and ONE optimizer
    (each capture re-binds the gradient tensors its backward writes).  A sequence of steps that hops between the three
    graphs must leave exactly the parameters of the same eager steps (the graphs draw their own negatives: replayed)."""
    import copy
    from ultra_torchdrug_amd import engine
    dev = _dev()
    task = _multi_graph_task(scale=8, num_negative=32).to(dev).train()
    state = copy.deepcopy(task.state_dict())
    twin = _multi_graph_task(scale=8, num_negative=32).to(dev).train()
    twin.load_state_dict(state)
    gen = torch.Generator().manual_seed(13)
    order = ["fb15k237", "wn18rr", "fb15k237", "codexm", "wn18rr", "codexm", "fb15k237"]
    batches = []
    for gid in order:
        fact = task.contexts[gid]["fact_graph"].edge_list
        batches.append((fact[torch.randperm(len(fact), generator=gen)[:16].to(dev)], gid))
    opt_g = torch.optim.AdamW(twin.parameters(), lr=1e-3)
    graphed = engine.GraphedMultiGraphTrainStep(twin, opt_g, 16)
    opt_e = torch.optim.AdamW(task.parameters(), lr=1e-3)
    for batch in batches:
        loss_g, _ = graphed(batch)
        negatives = graphed.steps[batch[1]].last_negatives.clone()
        task._static_negative = negatives
        loss_e, _ = engine.train_step(task, opt_e, batch)
        task._static_negative = None
        assert loss_g.item() == loss_e.item(), batch[1]
    assert set(graphed.steps) == set(PRETRAIN_3G)
    for (k, a), (_, b) in zip(task.named_parameters(), twin.named_parameters()):
        assert torch.equal(a, b), k


def test_multi_graph_graphed_steps_send_one_round_of_buckets_per_step_over_rccl():
    """Config 4's fastest mode WITH the gradient reducer on the real backend (a one-rank `nccl` = RCCL group): every context
    is captured in the constructor with the hooks paused -- the only collectives of the constructor are the ``warm()`` round
    -- and every call, first uses of a graph and ragged eager batches included, sends exactly one all-reduce per bucket
    (round 3 captured lazily with live hooks, `warmup` extra rounds on the rank that met a graph first).
    Parameters after the hop sequence equal those of eager steps with a reducer on the same negatives."""
    import copy
    import os
    import torch.distributed as dist
    from ultra_torchdrug_amd import engine
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    dev = _dev()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1, device_id=dev)
    try:
        task = _multi_graph_task(scale=8, num_negative=32).to(dev).train()
        state = copy.deepcopy(task.state_dict())
        twin = _multi_graph_task(scale=8, num_negative=32).to(dev).train()
        twin.load_state_dict(state)
        gen = torch.Generator().manual_seed(17)
        order = [("wn18rr", 16), ("fb15k237", 16), ("wn18rr", 16), ("codexm", 9), ("codexm", 16), ("fb15k237", 16)]   # 9: ragged
        batches = []
        for gid, n in order:
            fact = task.contexts[gid]["fact_graph"].edge_list
            batches.append((fact[torch.randperm(len(fact), generator=gen)[:n].to(dev)], gid))
        opt_g = torch.optim.AdamW(twin.parameters(), lr=1e-3)
        reducer = engine.GradientReducer(twin, overlap=True, single_rank=True)
        graphed = engine.GraphedMultiGraphTrainStep(twin, opt_g, 16, reducer=reducer)
        n_buckets = len(reducer.buckets)
        assert set(graphed.steps) == set(PRETRAIN_3G)                       # all captured up front, none lazily
        assert reducer.total_launched == n_buckets and reducer.rounds == 0  # the constructor: warm() and nothing else
        negatives = []
        for batch in batches:
            before, rounds = reducer.total_launched, reducer.rounds
            torch.manual_seed(len(negatives))                               # (the ragged eager step draws with torch.rand)
            graphed(batch)
            assert reducer.total_launched - before == n_buckets and reducer.rounds == rounds + 1, batch[1]
            step = graphed.steps[batch[1]] if len(batch[0]) == 16 else twin      # a replay rewrites ITS capture's tensor
            negatives.append(step.last_negatives.clone())
        torch.cuda.synchronize()
        opt_e = torch.optim.AdamW(task.parameters(), lr=1e-3)
        reducer_e = engine.GradientReducer(task, overlap=True, single_rank=True)
        for batch, neg in zip(batches, negatives):
            task._static_negative = neg
            engine.train_step(task, opt_e, batch, reducer=reducer_e)
        task._static_negative = None
        for (k, a), (_, b) in zip(task.named_parameters(), twin.named_parameters()):
            assert torch.equal(a, b), k
        reducer.remove_hooks()
        reducer_e.remove_hooks()
    finally:
        dist.destroy_process_group()


def test_multi_graph_graphed_steps_on_two_ranks_that_draw_different_graphs():
    """TWO ranks (child processes; they share this box's GPU over gloo, one GPU each over RCCL where two are visible) run
    config 4's graphed step with a reducer while drawing DIFFERENT graphs -- each rank meets each graph for the first time on
    another step.  The job must finish (no diverging collective sequences: round 3's lazy capture paired one rank's warm-up
    rounds with the other's gradients and hung at the end), both ranks must hold the same parameters, every step must have
    sent one round of buckets, and the graphed steps must equal eager reducer steps on the same negatives."""
    import json
    import os
    import socket
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multigraph_ranks_child.py")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, child, str(r), "2", str(port), backend], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    reports = []
    try:
        for proc in procs:
            out, err = proc.communicate(timeout=900)
            lines = [line for line in out.splitlines() if line.startswith("{")]
            assert proc.returncode == 0 and lines, "rank failed (rc %s): %s" % (proc.returncode, err[-2000:])
            reports.append(json.loads(lines[-1]))
    finally:
        for proc in procs:                      # a hang is the failure this test exists for: end exactly these two
            if proc.poll() is None:
                proc.kill()
    for rep in reports:
        assert rep["captured"] == sorted(PRETRAIN_3G), rep
        assert rep["warm_launches"] == rep["buckets"], rep                  # constructor: one warm round only
        assert rep["per_step"] == [rep["buckets"]] * len(rep["per_step"]), rep
        assert rep["graphed_equals_eager"] and rep["losses_equal"] and rep["ranks_hold_equal_parameters"] and rep["finite"], rep
        # what ran the collectives: on a node with two GPUs this IS RCCL on two different devices (the first such run must say so)
        assert rep["backend"] == backend and rep["distinct_devices"] == (2 if backend == "nccl" else 1), rep
        assert len(set(rep["modes"].values())) == 1 and rep["modes"] == reports[0]["modes"], rep
        # the phase boundaries: on every step every rank sent the same groups in the same order (one all-reduce per group, whichever
        # graph -- i.e. however long a phase -- the rank was on), and the reduced buffer the optimizers read was the same on both
        n_groups = len(rep["groups"])
        for said in rep["groups_sent_per_step"]:
            assert [g for g, _ in said] == list(range(n_groups)), rep["groups_sent_per_step"]
        assert rep["groups_sent_per_step"] == reports[0]["groups_sent_per_step"], (rep["groups_sent_per_step"], reports[0]["groups_sent_per_step"])
        assert all(rep["reduced_buffers_equal_on_all_ranks_per_step"]) and all(rep["reduced_buffer_nonzero_per_step"]), rep

"""GPU tests of the first-layer frontier kernel (csrc/frontier.inc) and of the key-based per-step index kernels
(csrc/sampler.inc): filtered ranks, strict negatives and edge removal without dense masks / match() / host syncs.

Bars: the frontier result must be EQUAL to the full forward kernel on the dense boundary (and to the oracle in the
kernels' summation order); ranks, negatives and removed-edge weights are integer / exact work and must be EQUAL to
the reference formulation (``ultra/task.py:65-118,307-315``, ``ultra/model.py:57-74``) restated with dense masks.
"""
import zlib

import numpy as np
import pytest
import torch

from graphs import random_graph

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _t(a):
    return torch.from_numpy(np.asarray(a)).to(_dev())


# ------------------------------------------------------------------------------------------------ frontier
@pytest.mark.parametrize("case", ["uniform", "hub_split_rows_multi_relation", "weights", "kg_shape"])
def test_frontier_equals_full_kernel_on_dense_boundary(oracle, case):
    from ultra_torchdrug_amd import RelCSR, functional as UF
    dev = _dev()
    rng = np.random.default_rng(zlib.crc32(case.encode()) % 1000)
    if case == "uniform":
        n, r = 500, 9
        g = random_graph(seed=1, n_node=n, n_edge=6000, n_rel=r)
        opts = {}
    elif case == "hub_split_rows_multi_relation":
        # node 3 receives 2 000 edges (split into pieces of 64) and several source nodes reach it through MANY relations:
        # the runs (src -> 3, *) straddle piece boundaries with more than one term on either side
        n, r = 300, 24
        g = random_graph(seed=2, n_node=n, n_edge=5000, n_rel=r, hub_row=3, hub_edges=2000)
        for s in (7, 8, 9):
            extra = np.arange(r, dtype=np.int64)
            g["dst"] = np.concatenate([g["dst"], np.full(r, 3)])
            g["src"] = np.concatenate([g["src"], np.full(r, s)])
            g["rel"] = np.concatenate([g["rel"], extra])
        opts = dict(chunk_edges=16, piece_len=64)
    elif case == "weights":
        n, r = 400, 7
        g = random_graph(seed=3, n_node=n, n_edge=9000, n_rel=r, weights=True, skew=True)
        opts = {}
    else:
        from graphs import kg_graph
        n, r = 14541, 474
        g = kg_graph(1024, n, 272115, 237)
        opts = {}
    csr = RelCSR(_t(g["dst"]), _t(g["src"]), _t(g["rel"]), None if g["w"] is None else _t(g["w"]), n, n, r, **opts)
    B = 6
    F = B * 64
    relation = torch.from_numpy(rng.standard_normal((r, F)).astype(np.float32)).to(dev)
    value = torch.from_numpy(rng.standard_normal((B, 64)).astype(np.float32)).to(dev)
    if case == "hub_split_rows_multi_relation":
        nodes = [7, 8, 9, 3, 7, 250]            # sources of the multi-relation runs, the hub itself, a repeat
    else:
        deg_out = torch.bincount(csr.src, minlength=n)
        nodes = [int(deg_out.argmax()), int((deg_out == 0).nonzero()[0]) if (deg_out == 0).any() else 1, 0, n - 1, 5, 5]
    node = torch.tensor(nodes, dtype=torch.int32, device=dev)
    dense = torch.zeros(n, B, 64, device=dev)
    dense[node.long(), torch.arange(B, device=dev)] = value                  # model.py:106-107
    dense = dense.flatten(1)

    got = UF.rspmm_frontier(csr, relation, (node, value))
    want = UF.rspmm_forward(csr, relation, dense, "add", "mul", boundary=(node, value))
    assert torch.equal(got, want), "frontier differs from the full kernel: max |diff| %.3g" % (got - want).abs().max()
    want_dense_add = UF.rspmm_forward(csr, relation, dense, "add", "mul", add_rows=dense)
    assert torch.equal(got, want_dense_add)
    if n <= 1000:
        csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
        want_o = oracle.rspmm_forward(csr_o, relation.cpu().numpy(), dense.cpu().numpy(), "add", "mul", piece=csr.piece_len)
        assert np.array_equal(got.cpu().numpy(), want_o + dense.cpu().numpy())


@pytest.mark.parametrize("case", ["uniform", "hub_split_rows_multi_relation", "weights", "kg_shape_32_queries", "self_loops"])
@pytest.mark.parametrize("norm,relu,shortcut", [(True, True, True), (False, True, False), (True, False, True)])
def test_sparse_first_layer_equals_frontier_plus_dense_epilogue(monkeypatch, case, norm, relu, shortcut):
    """``ultra_first_layer_sparse_f32`` (round 4): the whole first layer with the epilogue on the rows the frontier reaches
    only, one constant vector broadcast everywhere else, must EQUAL ``ultra_rspmm_frontier_f32`` followed by the dense
    boundary-form epilogue bit for bit: isolated heads, repeated heads, self loops (the head's own row is then listed by a
    run, not by its spare slot), hubs with split rows, per-edge weights, and an FB15k237-shaped batch of 32 queries."""
    from ultra_torchdrug_amd import RelCSR, functional as UF
    dev = _dev()
    monkeypatch.setattr(UF, "SPARSE_FIRST_LAYER_MIN_ROWS", 0)      # (the size heuristic would send the small graphs to the dense form)
    rng = np.random.default_rng(zlib.crc32(case.encode()) % 1000)
    opts = {}
    if case == "uniform":
        n, r = 500, 9
        g = random_graph(seed=1, n_node=n, n_edge=6000, n_rel=r)
    elif case == "hub_split_rows_multi_relation":
        n, r = 300, 24
        g = random_graph(seed=2, n_node=n, n_edge=5000, n_rel=r, hub_row=3, hub_edges=2000)
        for s_ in (7, 8, 9):
            g["dst"] = np.concatenate([g["dst"], np.full(r, 3)])
            g["src"] = np.concatenate([g["src"], np.full(r, s_)])
            g["rel"] = np.concatenate([g["rel"], np.arange(r, dtype=np.int64)])
        opts = dict(chunk_edges=16, piece_len=64)
    elif case == "weights":
        n, r = 400, 7
        g = random_graph(seed=3, n_node=n, n_edge=9000, n_rel=r, weights=True, skew=True)
    elif case == "self_loops":
        n, r = 200, 5
        g = random_graph(seed=6, n_node=n, n_edge=3000, n_rel=r, skew=True)
        loops = np.array([0, 5, 5, 17, 199], dtype=np.int64)                   # node 5 loops through two relations
        g["dst"] = np.concatenate([g["dst"], loops])
        g["src"] = np.concatenate([g["src"], loops])
        g["rel"] = np.concatenate([g["rel"], np.array([1, 0, 3, 2, 4], dtype=np.int64)])
    else:
        from graphs import kg_graph
        n, r = 14541, 474
        g = kg_graph(1024, n, 272115, 237)
    csr = RelCSR(_t(g["dst"]), _t(g["src"]), _t(g["rel"]), None if g["w"] is None else _t(g["w"]), n, n, r, **opts)
    deg_out = torch.bincount(csr.src, minlength=n)
    if case == "kg_shape_32_queries":
        order = torch.argsort(deg_out, descending=True)
        nodes = order[:12].tolist() + order[2000:2016].tolist() + [int(order[0]), int(order[-1]), 0, n - 1]   # hubs, mid, repeat, leaf
    elif case == "self_loops":
        nodes = [0, 5, 17, 199, 5, 3, 1, 2]
    elif case == "hub_split_rows_multi_relation":
        nodes = [7, 8, 9, 3, 7, 250]
    else:
        nodes = [int(deg_out.argmax()), int((deg_out == 0).nonzero()[0]) if (deg_out == 0).any() else 1, 0, n - 1, 5, 5]
    Q = len(nodes)
    F = Q * 64
    relation = torch.from_numpy(rng.standard_normal((r, F)).astype(np.float32)).to(dev)
    value = torch.from_numpy(rng.standard_normal((Q, 64)).astype(np.float32)).to(dev)
    node = torch.tensor(nodes, dtype=torch.int32, device=dev)
    lin = torch.nn.Linear(128, 64).to(dev)
    ln = torch.nn.LayerNorm(64).to(dev) if norm else None
    with torch.no_grad():
        if ln is not None:
            ln.weight.copy_(torch.from_numpy(rng.standard_normal(64).astype(np.float32)))
            ln.bias.copy_(torch.from_numpy(rng.standard_normal(64).astype(np.float32)))
        args = (lin.weight, lin.bias, ln.weight if ln else None, ln.bias if ln else None, 1e-5, relu, shortcut)
        update = UF.rspmm_frontier(csr, relation, (node, value)).view(n, Q, 64)
        touched = int((update != 0).any(-1).sum())
        want = UF.combine_forward(None, update, *args, reuse_update=False, input_boundary=(node, value))
        got = UF.first_layer_forward(csr, relation, (node, value), *args)
        assert got is not None, "the sparse first layer declined a shape it should take"
        assert got.shape == want.shape and torch.equal(got, want), "max |diff| %.3g" % (got - want).abs().max()
        again = UF.first_layer_forward(csr, relation, (node, value), *args)          # the row list is rebuilt every launch
        assert torch.equal(again, want)
        # round 6: vocabularies beyond LDS (S-stress: 1 000 relations) take frontier_kernel -- relation rows through L2 -- which lists
        # its rows too; knob bit 0 selects it here
        lib = UF._lib.load()
        lib.ultra_rspmm_force_general_path(1)
        try:
            general = UF.first_layer_forward(csr, relation, (node, value), *args)
        finally:
            lib.ultra_rspmm_force_general_path(0)
        assert general is not None and torch.equal(general, want), "the L2-row frontier kernel's listed first layer differs"
    run_prefix, max_runs = csr.frontier_runs
    assert run_prefix.dtype == torch.int32 and run_prefix.shape == (csr.n_edges,) and int(run_prefix[-1]) >= max_runs > 0
    assert touched < n * Q                                                            # (some rows really are the constant)


def test_first_layer_uses_the_frontier_and_predict_is_unchanged():
    """task.predict with and without the first-layer shortcut: identical scores (the shortcut is bit-compatible)."""
    from ultra_torchdrug_amd import layer
    from ultra_torchdrug_amd.data import synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.task import build_ultra
    triples, n, r = synthetic_triples((1200, 9000, 12), 1024)
    torch.manual_seed(1024)
    task = build_ultra(r)
    task.preprocess(Graph(torch.from_numpy(triples), num_node=n, num_relation=r))
    task.to(_dev()).eval()
    batch = _t(triples[:16])
    calls = []
    real = layer.backend.get().rspmm_frontier
    from ultra_torchdrug_amd import functional as UF
    UF.rspmm_frontier = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    real_first = UF.first_layer_forward                  # the entity stack's first layer: frontier + sparse epilogue, fused

    def counted_first(*a, **k):
        out = real_first(*a, **k)
        if out is not None:
            calls.append(1)
        return out
    UF.first_layer_forward = counted_first
    try:
        with torch.no_grad():
            with_frontier = task.predict(batch)
            layer.FRONTIER_FIRST_LAYER = False
            n_calls = len(calls)
            without = task.predict(batch)
    finally:
        layer.FRONTIER_FIRST_LAYER = True
        UF.rspmm_frontier = real
        UF.first_layer_forward = real_first
    assert n_calls == 2 and len(calls) == 2            # entity stack + relation stack, first layer each
    assert torch.equal(with_frontier, without)


@pytest.mark.parametrize("bad", [float("inf"), float("nan")])
def test_frontier_non_finite_relation_table_takes_the_full_kernel(bad):
    """``inf * 0 = NaN``: with a non-finite relation entry the full kernel writes NaN at EVERY destination of an edge of that
    relation (its zero source rows included), the frontier kernel only along the boundary nodes' out-edges.  The layer
    tests the table on eager calls and takes the full kernel then, so the first layer keeps the reference's result;
    ``functional.rspmm_frontier`` itself documents finiteness as its precondition."""
    from ultra_torchdrug_amd import RelCSR, functional as UF, layer
    from ultra_torchdrug_amd.graph import Graph
    dev = _dev()
    n, r, B = 400, 6, 4
    g = random_graph(seed=5, n_node=n, n_edge=5000, n_rel=r)
    graph = Graph(torch.stack([_t(g["src"]), _t(g["dst"]), _t(g["rel"])], dim=1), num_node=n, num_relation=r)
    rng = np.random.default_rng(3)
    conv = layer.GeneralizedRelationalConvNBFMod(64, 64, r, 64, message_func="distmult", aggregate_func="sum",
                                                 layer_norm=True, project=False).to(dev).eval()
    table = torch.from_numpy(rng.standard_normal((B, r, 64)).astype(np.float32)).to(dev)
    table[1, 2, 5] = bad                                                      # query 1, relation 2, column 5
    conv.relation = table
    node = torch.tensor([3, 9, 27, 81], dtype=torch.int32, device=dev)
    value = torch.from_numpy(rng.standard_normal((B, 64)).astype(np.float32)).to(dev)
    boundary = torch.zeros(n, B, 64, device=dev)
    boundary[node.long(), torch.arange(B, device=dev)] = value
    graph.query, graph.boundary, graph.boundary_sparse = value, boundary, (node, value)
    relation_input = table.transpose(0, 1).flatten(1).contiguous()
    want = UF.rspmm_forward(graph.relcsr, relation_input, boundary.flatten(1), "add", "mul", boundary=(node, value))
    assert torch.isnan(want).any()
    calls = []
    real = UF.rspmm_frontier
    UF.rspmm_frontier = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            got = conv.message_and_aggregate(graph, boundary, input_is_boundary=True)
            table[1, 2, 5] = 0.25                                             # finite again: the shortcut is back
            conv.message_and_aggregate(graph, boundary, input_is_boundary=True)
    finally:
        UF.rspmm_frontier = real
    assert len(calls) == 1, "the non-finite table must not go through the frontier kernel"
    assert torch.equal(torch.isnan(got.flatten(1)), torch.isnan(want))
    assert torch.equal(torch.nan_to_num(got.flatten(1), nan=0.0), torch.nan_to_num(want, nan=0.0))
    # what the guard protects against: the bare kernel reaches fewer rows
    table[1, 2, 5] = bad
    bare = UF.rspmm_frontier(graph.relcsr, table.transpose(0, 1).flatten(1).contiguous(), (node, value))
    assert int(torch.isnan(bare).any(1).sum()) < int(torch.isnan(want).any(1).sum())


# ------------------------------------------------------------------------------------------------ sampler
def _kg(n=700, triples=6000, r=9, seed=4):
    from ultra_torchdrug_amd.data import synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    tr, n, r = synthetic_triples((n, triples, r), seed)
    tr = np.concatenate([tr, tr[:50]])                  # duplicate fact edges: filters must stay DISTINCT lists
    return Graph(torch.from_numpy(tr).to(_dev()), num_node=n, num_relation=r), tr


def test_filtered_rank_from_keys_equals_dense_mask_ranking():
    from ultra_torchdrug_amd import functional as UF
    from ultra_torchdrug_amd.task import build_ultra
    graph, tr = _kg()
    task = build_ultra(graph.num_relation)
    task.preprocess(graph)
    dev = _dev()
    batch = _t(tr[:33])
    gen = torch.Generator(device=dev).manual_seed(1)
    pred = (torch.randn(33, 2, graph.num_node, device=dev, generator=gen) * 4).round() / 4        # many exact ties
    mask, target = task.target(batch)
    want = task.get_ranking(pred, (mask, target))
    got = task.rank_batch(batch, pred=pred)
    assert got.dtype == torch.int64 and torch.equal(got, want)
    task.filtered_ranking = False
    assert torch.equal(task.rank_batch(batch, pred=pred), task.get_ranking(pred, (mask, target)))
    # the lists the older entry point takes describe the same filter
    task.filtered_ranking = True
    (ptr, node), _ = task.target_lists(batch)
    assert torch.equal(UF.filtered_rank(pred.flatten(0, 1), target.flatten(), ptr, node).view(-1, 2), want)


def test_filtered_rank_over_many_candidates_counts_in_slices():
    """Rows of more than 64 K candidates are counted by several workgroups each (one launch writes 1 - filtered count, one adds
    the slices with integer atomics): the same int64 ranks as the dense-mask formula (ultra/task.py:307-315), ragged last slice,
    ties, a strided (B, 2, N) view and the unfiltered form included."""
    from ultra_torchdrug_amd import functional as UF
    from ultra_torchdrug_amd.data import synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    dev = _dev()
    n, r = 3 * 32768 + 4321, 7
    tr, n, r = synthetic_triples((n, 40000, r), 6, alpha=0.0)
    tr[:3000, 0], tr[:3000, 2] = 11, 2                       # one (head, relation) with thousands of known tails
    graph = Graph(torch.from_numpy(tr).to(dev), num_node=n, num_relation=r)
    batch = _t(tr[[0, 5, 2999, 3500, 39999]])
    h, t, rel = batch.t()
    gen = torch.Generator(device=dev).manual_seed(3)
    pred = (torch.randn(5, 2, n, device=dev, generator=gen) * 8).round() / 8
    for side, (anchor, target) in enumerate(((h, t), (t, h))):
        keys = graph.completion_keys(side)
        got = UF.filtered_rank_keys(pred[:, side], target, keys, anchor, rel, r, n)
        pos = pred[:, side].gather(1, target.unsqueeze(1))
        mask = torch.ones(5, n, dtype=torch.bool, device=dev)
        for q in range(5):
            known = graph.edge_list[(graph.edge_list[:, side] == anchor[q]) & (graph.edge_list[:, 2] == rel[q]), 1 - side]
            mask[q, known] = False
        want = ((pos <= pred[:, side]) & mask).sum(-1) + 1
        assert torch.equal(got, want), side
        assert torch.equal(UF.filtered_rank_keys(pred[:, side], target, None, anchor, rel, r, n), (pos <= pred[:, side]).sum(-1) + 1)
        assert torch.equal(UF.filtered_rank_keys(pred[:, side], target, keys, anchor, rel, r, n), got)        # replay: same output buffer pattern


def test_strict_negatives_from_keys_equal_the_reference_sampler():
    """Same uniform numbers in, same negatives out as mask.nonzero() + variadic_sample (task.py:102-118)."""
    from ultra_torchdrug_amd import functional as UF
    from ultra_torchdrug_amd.task import build_ultra, variadic_sample
    graph, tr = _kg()
    task = build_ultra(graph.num_relation, num_negative=32)
    task.preprocess(graph)
    dev = _dev()
    batch = _t(tr[100:164])
    h, t, r = batch.t()
    half = len(batch) // 2
    torch.manual_seed(9)
    got = task._strict_negative(h, t, r)
    # the reference formulation, fed the same random stream
    torch.manual_seed(9)
    t_mask = task._calculate_t_mask(task.fact_graph, h[:half], r[:half])
    rand = torch.rand(half, 32, device=dev)
    sizes = t_mask.sum(dim=-1)
    idx = (rand * sizes.unsqueeze(-1)).long() + (sizes.cumsum(0) - sizes).unsqueeze(-1)
    neg_t = t_mask.nonzero()[:, 1][idx]
    h_mask = task._calculate_h_mask(task.fact_graph, t[half:], r[half:])
    rand = torch.rand(len(batch) - half, 32, device=dev)
    sizes = h_mask.sum(dim=-1)
    idx = (rand * sizes.unsqueeze(-1)).long() + (sizes.cumsum(0) - sizes).unsqueeze(-1)
    neg_h = h_mask.nonzero()[:, 1][idx]
    assert got.shape == (64, 32) and torch.equal(got, torch.cat([neg_t, neg_h]))
    assert t_mask.gather(1, got[:half]).all() and h_mask.gather(1, got[half:]).all()        # strict: never a fact
    # extremes of the uniform numbers and a row whose every entity but one is a known completion
    n, nr = graph.num_node, graph.num_relation
    star = torch.stack([torch.zeros(n - 1, dtype=torch.long), torch.arange(1, n), torch.zeros(n - 1, dtype=torch.long)], 1)
    from ultra_torchdrug_amd.graph import Graph
    g2 = Graph(star.to(dev), num_node=n, num_relation=nr)
    edge = torch.tensor([[0.0, 0.5, 0.99999994]], device=dev)
    out = UF.strict_negatives(g2.completion_keys(0), torch.zeros(1, dtype=torch.long, device=dev),
                              torch.zeros(1, dtype=torch.long, device=dev), nr, n, edge)
    assert out.tolist() == [[0, 0, 0]]                  # node 0 is the only non-completion of (0, 0, ?)
    out = UF.strict_negatives(g2.completion_keys(0), torch.ones(1, dtype=torch.long, device=dev),
                              torch.zeros(1, dtype=torch.long, device=dev), nr, n, edge)
    assert out.tolist() == [[0, n // 2, n - 1]]         # nothing known about (1, 0, ?): plain floor(rand * n)


def test_native_edge_removal_equals_mask_and_reweight_path():
    """Graph.without_triples (one native call) against the reference's way (model.py:57-74: match + edge_mask, then
    undirected): the weights of all three plans, the forward result and both gradients."""
    from ultra_torchdrug_amd import functional as UF
    from ultra_torchdrug_amd.graph import Graph
    graph, tr = _kg()
    dev = _dev()
    n, r = graph.num_node, graph.num_relation
    und = graph.undirected(add_inverse=True)
    pos = _t(tr[:20])
    gen = torch.Generator(device=dev).manual_seed(2)
    # the pattern grid of a training batch: positives in column 0, (strict) negatives elsewhere -- those match nothing
    h = pos[:, 0:1].repeat(1, 5)
    t = pos[:, 1:2].repeat(1, 5)
    rr = pos[:, 2:3].repeat(1, 5)
    t[:, 1:] = torch.randint(0, n, (20, 4), device=dev, generator=gen)
    edge_index = graph.match(torch.stack([h, t, rr], dim=-1).flatten(0, 1))[0]
    keep = torch.ones(graph.num_edge, dtype=torch.bool, device=dev)
    keep[edge_index] = False
    assert int((~keep).sum()) >= 20
    want_graph = und.reweighted(und.edge_weight * keep.repeat_interleave(2))
    got_graph = und.without_triples(h, t, rr, r)
    for plan in ("fwd", "by_src", "by_rel"):
        a, b = getattr(got_graph.relcsr, plan), getattr(want_graph.relcsr, plan)
        assert torch.equal(a.weight, b.weight), plan
        assert a.chunks.data_ptr() == getattr(und.relcsr, plan).chunks.data_ptr()        # plans shared, not rebuilt
    F = 128
    relation = torch.randn(2 * r, F, device=dev, generator=gen).requires_grad_()
    x = torch.randn(n, F, device=dev, generator=gen).requires_grad_()
    grad = torch.randn(n, F, device=dev, generator=gen)
    outs = []
    for g in (got_graph, want_graph, Graph(und.edge_list[keep.repeat_interleave(2)], None, n, 2 * r)):
        relation.grad = x.grad = None
        out = UF.generalized_rspmm(g.relcsr, relation, x)
        out.backward(grad)
        outs.append((out.detach(), relation.grad.clone(), x.grad.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    for a, b in zip(outs[0], outs[2]):                  # really removing the edges: another plan, hence another
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)     # chunk / piece structure and order of fp32 additions
    # reweighting a reweighted graph keeps working through the backward (plans come from the base object)
    twice = want_graph.reweighted(want_graph.edge_weight * 0.5)
    relation.grad = x.grad = None
    UF.generalized_rspmm(twice.relcsr, relation, x).backward(grad)
    torch.testing.assert_close(x.grad, 0.5 * outs[1][2], rtol=1e-6, atol=1e-6)


def test_graphed_train_step_captures_negatives_and_edge_removal():
    """With the sampler and the edge removal on the device, a whole fine-tuning step (negatives, edge removal, forward,
    backward) replays as ONE hipGraph; same losses and parameters as eager steps fed the same random stream."""
    import copy
    from ultra_torchdrug_amd import engine
    from ultra_torchdrug_amd.data import synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.task import build_ultra
    triples, n, r = synthetic_triples("S-tiny", 1024)
    torch.manual_seed(1024)
    task = build_ultra(r, num_negative=16)
    task.preprocess(Graph(torch.from_numpy(triples), num_node=n, num_relation=r))
    dev = _dev()
    task.to(dev).train()
    twin = copy.deepcopy(task)
    batches = [_t(triples[i:i + 8]) for i in (0, 8, 16, 24)]
    opt_g = torch.optim.AdamW(twin.parameters(), lr=1e-3)
    step = engine.GraphedTrainStep(twin, opt_g, batches[0])
    losses_g, negs = [], []
    for b in batches[1:]:
        losses_g.append(step(b)[0].item())
        negs.append(step.last_negatives.clone())
    opt_e = torch.optim.AdamW(task.parameters(), lr=1e-3)
    losses_e = []
    for b, neg in zip(batches[1:], negs):
        task._static_negative = neg                    # the graph drew its own negatives: replay them eagerly
        losses_e.append(engine.train_step(task, opt_e, b)[0].item())
    task._static_negative = None
    assert losses_g == losses_e
    for (k, a), (_, b) in zip(task.named_parameters(), twin.named_parameters()):
        assert torch.equal(a, b), k
    assert not torch.equal(negs[0], negs[1])            # the captured RNG advances between replays


# ------------------------------------------------------------------------------------------------ rowgroup / raw CSR
@pytest.mark.parametrize("case", ["short_rows", "weights_many_relations", "relations_beyond_lds", "ragged_with_empty_rows"])
def test_rowgroup_kernel_and_raw_csr_entry_match_oracle(oracle, case):
    """One row per 16-lane group (csrc/rowgroup.inc): (i) as the kernel run_plan picks for big-graph plans without split
    rows (forced on small graphs with wide_ids=True), against the chunked kernels (knob bit 3) and the oracle; (ii) as
    the raw-CSR entry ultra_rspmm_fwd_f32, whose order is the strictly sequential reference order for every row."""
    from ultra_torchdrug_amd import RelCSR, _lib, functional as UF
    dev = _dev()
    if case == "short_rows":
        n, r, F = 3000, 40, 256
        g = random_graph(seed=5, n_node=n, n_edge=30000, n_rel=r)
    elif case == "weights_many_relations":
        n, r, F = 700, 900, 128          # relation table beyond LDS: its first 624 / 312 rows from LDS, the rest through L2
        g = random_graph(seed=6, n_node=n, n_edge=9000, n_rel=r, weights=True)
    elif case == "relations_beyond_lds":
        n, r, F = 400, 700, 256          # 64-lane groups hold 156 of 700 relation rows: below a quarter, all rows through L2
        g = random_graph(seed=8, n_node=n, n_edge=5000, n_rel=r)
    else:
        n, r, F = 500, 5, 192
        g = random_graph(seed=7, n_node=n, n_edge=6000, n_rel=r, isolated=120, hub_row=9, hub_edges=100)
    rng = np.random.default_rng(1)
    relation = rng.standard_normal((r, F)).astype(np.float32)
    x = rng.standard_normal((n, F)).astype(np.float32)
    grad = rng.standard_normal((n, F)).astype(np.float32)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
    csr = RelCSR(_t(g["dst"]), _t(g["src"]), _t(g["rel"]), None if g["w"] is None else _t(g["w"]), n, n, r,
                 wide_ids=True, piece_len=512, chunk_edges=128)
    assert csr.fwd.packed_src_shift == 32 and csr.fwd.row_ptr is not None and csr.fwd.n_pieces == 0
    lib = _lib.load()
    rel_t, x_t, g_t = _t(relation), _t(x), _t(grad)
    for s in ("add", "min", "max"):
        for m in ("mul", "add"):
            want = oracle.rspmm_forward(csr_o, relation, x, s, m, piece=0)
            got = UF.rspmm_forward(csr, rel_t, x_t, s, m)
            lib.ultra_rspmm_force_general_path(8)
            try:
                chunked = UF.rspmm_forward(csr, rel_t, x_t, s, m)
            finally:
                lib.ultra_rspmm_force_general_path(0)
            assert np.array_equal(got.cpu().numpy(), want), (s, m)
            assert torch.equal(got, chunked), (s, m)
            raw = UF.rspmm_forward_csr(csr.fwd.row_ptr, csr.fwd.node_a[:csr.n_edges], csr.fwd.rel,
                                       None if csr.unit_weight else csr.weight, rel_t, x_t, s, m)
            assert np.array_equal(raw.cpu().numpy(), want), ("raw", s, m)
    # wide groups (32 / 64 lanes per row: column tiles of 128 / 256, what big DRAM-resident inputs get): same bits
    if F % 128 == 0:
        lib.ultra_rspmm_force_general_path(16)
        try:
            for s in ("add", "max"):
                wide = UF.rspmm_forward(csr, rel_t, x_t, s, "mul")
                assert np.array_equal(wide.cpu().numpy(), oracle.rspmm_forward(csr_o, relation, x, s, "mul", piece=0)), s
            d_x_w, _ = UF.rspmm_backward(csr, rel_t, x_t, None, g_t, "add", "mul", need_relation=False)
        finally:
            lib.ultra_rspmm_force_general_path(0)
        d_x_n, _ = UF.rspmm_backward(csr, rel_t, x_t, None, g_t, "add", "mul", need_relation=False)
        assert torch.equal(d_x_w, d_x_n)
    # fused boundary epilogues ride along as in the chunked kernels
    node = torch.tensor([1, 17, 3, 9][:F // 64], dtype=torch.int32, device=dev)
    value = torch.randn(F // 64, 64, device=dev)
    with_b = UF.rspmm_forward(csr, rel_t, x_t, "add", "mul", boundary=(node, value))
    lib.ultra_rspmm_force_general_path(8)
    try:
        assert torch.equal(with_b, UF.rspmm_forward(csr, rel_t, x_t, "add", "mul", boundary=(node, value)))
    finally:
        lib.ultra_rspmm_force_general_path(0)
    # d_input through the same kernel (rows = source nodes, gathers output_grad)
    for m in ("mul", "add"):
        out = oracle.rspmm_forward(csr_o, relation, x, "add", m, piece=0)
        d_rel_o, d_x_o = oracle.rspmm_backward(csr_o, relation, x, out, grad, "add", m, piece=csr.piece_len)
        d_x, d_rel = UF.rspmm_backward(csr, rel_t, x_t, None, g_t, "add", m)
        assert np.array_equal(d_x.cpu().numpy(), d_x_o) and np.array_equal(d_rel.cpu().numpy(), d_rel_o), m


# ------------------------------------------------------------------------------------------------ fused epilogue backward
@pytest.mark.parametrize("rows", [1, 31, 33, 1000, 65536 + 7, 232656])
@pytest.mark.parametrize("ln,relu,shortcut", [(True, True, True), (False, True, False), (True, False, False)])
def test_one_pass_combine_backward_equals_three_kernel_backward(monkeypatch, rows, ln, relu, shortcut):
    """csrc/combine_fused_bwd.inc (one pass over the rows) against the three-kernel backward of combine_train.inc:
    d_input / d_update carry the SAME bits (same k order of the fmaf chains); the parameter gradients are sums over
    the rows taken in another order (tiles dealt round-robin to waves instead of contiguous row ranges): fp32
    tolerance 2e-5 of each gradient's scale, the bar of the autograd test in tests/test_rspmm_gpu.py."""
    from ultra_torchdrug_amd import functional as UF
    dev = _dev()
    gen = torch.Generator(device=dev).manual_seed(rows)
    x = torch.randn(rows, 64, device=dev, generator=gen)
    u = torch.randn(rows, 64, device=dev, generator=gen) * 2
    gout = torch.randn(rows, 64, device=dev, generator=gen)
    torch.manual_seed(rows + 1)
    lin, norm = torch.nn.Linear(128, 64).to(dev), torch.nn.LayerNorm(64).to(dev)
    with torch.no_grad():
        norm.weight.copy_(torch.rand(64, device=dev) + 0.5)
        norm.bias.copy_(torch.randn(64, device=dev) * 0.1)
    results = {}
    for mode in ("fused", "split"):
        monkeypatch.setenv("ULTRA_COMBINE_BWD", mode)
        leaves = [t.detach().clone().requires_grad_() for t in (x, u, lin.weight, lin.bias, norm.weight, norm.bias)]
        out = UF.combine(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4] if ln else None, leaves[5] if ln else None,
                         norm.eps, relu, shortcut)
        out.backward(gout)
        results[mode] = [t.grad for t in (leaves if ln else leaves[:4])]
    names = ["d_input", "d_update", "d_weight", "d_bias", "d_ln_weight", "d_ln_bias"]
    for name, a, b in zip(names, results["fused"], results["split"]):
        if name in ("d_input", "d_update"):
            assert torch.equal(a, b), name
        else:
            scale = b.abs().max().item() + 1e-12
            assert (a - b).abs().max().item() <= 2e-5 * scale + 1e-6, name


# ------------------------------------------------------------------------------------------------ grouped projections, training
@pytest.mark.parametrize("batch,n_rel,n_layers", [(3, 5, 1), (16, 22, 6), (16, 474, 6), (64, 102, 8), (1, 1, 2)])
def test_grouped_relation_projection_backward_matches_autograd_of_the_reference_chain(batch, n_rel, n_layers):
    """functional.relation_project_train (csrc/project_bwd.inc): forward = the inference tables bit for bit; backward =
    one launch for the gradients of all layers' four parameters and of the relation representations, against fp64
    autograd of the reference's own chain -- nn.Linear, relu, nn.Linear, transpose (ultra/layer.py:318-319,325-326) --
    to 2e-5 of each gradient's scale (the bar of the fused epilogue's backward).  One layer's table receives no gradient
    at all (a NULL entry of the C call)."""
    from ultra_torchdrug_amd import functional as UF
    dev = _dev()
    gen = torch.Generator(device=dev).manual_seed(batch * 1000 + n_rel)
    relation = torch.randn(batch, n_rel, 64, device=dev, generator=gen)
    weights = []
    for _ in range(n_layers):
        weights.append((torch.randn(64, 64, device=dev, generator=gen) * 0.2, torch.randn(64, device=dev, generator=gen) * 0.1,
                        torch.randn(64, 64, device=dev, generator=gen) * 0.2, torch.randn(64, device=dev, generator=gen) * 0.1))
    table_grads = [torch.randn(n_rel, batch * 64, device=dev, generator=gen) for _ in range(n_layers)]
    unused = n_layers - 1 if n_layers > 1 else None

    def run(dtype, fn):
        x = relation.to(dtype).detach().clone().requires_grad_()
        ws = [tuple(t.to(dtype).detach().clone().requires_grad_() for t in lw) for lw in weights]
        tables = fn(x, ws)
        loss = sum((t * g.to(dtype)).sum() for k, (t, g) in enumerate(zip(tables, table_grads)) if k != unused)
        loss.backward()
        return tables, x.grad, [[t.grad for t in lw] for lw in ws]

    chain = lambda x, ws: [torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(x, w1, b1)), w2, b2)
                           .transpose(0, 1).flatten(1) for w1, b1, w2, b2 in ws]
    tables, d_x, d_w = run(torch.float32, UF.relation_project_train)
    with torch.no_grad():
        plain = UF.relation_project(relation, weights)
    for a, b in zip(tables, plain):
        assert torch.equal(a.detach(), b)
    _, d_x64, d_w64 = run(torch.float64, chain)

    def close(got, want, what):
        scale = want.abs().max().item() + 1e-12
        assert (got.double() - want).abs().max().item() <= 2e-5 * scale, what

    close(d_x, d_x64, "d_relation")
    for l in range(n_layers):
        for k, name in enumerate(("w1", "b1", "w2", "b2")):
            if l == unused:        # no gradient reached this layer: exact zeros
                assert d_w[l][k] is None or not d_w[l][k].any(), (l, name)
            else:
                close(d_w[l][k], d_w64[l][k], (l, name))


def test_sparse_first_layer_refuses_a_row_list_that_cannot_hold_every_slot():
    """``ultra_first_layer_sparse_f32`` used to accept any ``row_list_len >= n_query``; rows whose slot did not fit kept
    their raw sums with no epilogue and the call returned ULTRA_OK.  The caller now states ``max_runs`` and a shorter list is
    ULTRA_ERR_BAD_SHAPE."""
    import ultra_torchdrug_amd as U
    from ultra_torchdrug_amd import RelCSR
    dev = _dev()
    n, r, Q = 300, 6, 4
    g = random_graph(seed=4, n_node=n, n_edge=4000, n_rel=r)
    csr = RelCSR(_t(g["dst"]), _t(g["src"]), _t(g["rel"]), None, n, n, r)
    lib = U.require_library()
    src_ptr, fwd_rank = csr.frontier_index
    run_prefix, max_runs = csr.frontier_runs
    relation = torch.randn(r, Q * 64, device=dev)
    node = torch.tensor([1, 2, 3, 4], dtype=torch.int32, device=dev)
    value = torch.randn(Q, 64, device=dev)
    weight, bias = torch.randn(64, 128, device=dev), torch.randn(64, device=dev)
    out = torch.empty(n, Q, 64, device=dev)
    offsets = torch.empty(Q + 1, dtype=torch.int32, device=dev)

    def call(list_len):
        row_list = torch.empty(max(list_len, 1), dtype=torch.int32, device=dev)
        return lib.ultra_first_layer_sparse_f32(
            csr.by_src.pointer, src_ptr.data_ptr(), fwd_rank.data_ptr(), run_prefix.data_ptr(), relation.data_ptr(), node.data_ptr(),
            value.data_ptr(), Q, weight.data_ptr(), bias.data_ptr(), None, None, 1e-5, 1, 0, out.data_ptr(), row_list.data_ptr(),
            list_len, max_runs, offsets.data_ptr(), n, r, torch.cuda.current_stream().cuda_stream)

    assert max_runs > 1
    if lib.ultra_first_layer_sparse_supported(n, r, Q):
        assert call(Q * (max_runs + 1)) == 0
        assert call(Q * (max_runs + 1) - 1) == 2 and call(Q) == 2          # ULTRA_ERR_BAD_SHAPE
    torch.cuda.synchronize()

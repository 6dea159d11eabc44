"""One entity layer of a Bellman-Ford in inference as ONE launch (csrc/layer_fused.hip, `functional.layer_forward`): the rspmm of
the row-per-group kernel with the 128 -> 64 epilogue inside its row loop.  SURVEY.md 8f-1 ("fusing into the rspmm row tile removes
one full read + write of (N, F) per layer"; /root/reference/ultra/layer.py:357-358,386-392, ultra/model.py:126-127).

The bar is EQUALITY with the two launches it replaces (`rspmm_forward(..., boundary=)` + `combine_forward`), which the other tests
hold to the oracle: every lane-group width (16 / 32 / 64 lanes per row), relation rows from LDS / partly from LDS / from L2, rows
of more than one 16-edge window, empty rows, a row count that leaves the last batch of four iterations ragged, per-edge weights,
with and without LayerNorm / relu / shortcut / boundary.
"""
import numpy as np
import pytest
import torch

from graphs import random_graph

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _layer_params(gen, dev, layer_norm=True):
    w = torch.randn(64, 128, device=dev, generator=gen) * 0.1
    b = torch.randn(64, device=dev, generator=gen) * 0.1
    g = (1 + 0.1 * torch.randn(64, device=dev, generator=gen)) if layer_norm else None
    beta = (0.1 * torch.randn(64, device=dev, generator=gen)) if layer_norm else None
    return w, b, g, beta


CASES = {
    # name: (n_node, n_edge, n_rel, n_query, knob, weights)
    "g16_rel_lds": (3001, 24000, 30, 1, 0, False),
    "g16_three_queries": (2500, 20000, 30, 3, 0, False),
    "g32_rel_part": (3003, 30000, 250, 2, 16, False),
    "g64_rel_l2": (2002, 16000, 1000, 4, 16, False),
    "g64_two_tiles": (1501, 12000, 40, 8, 16, False),
    "g32_weights": (2000, 18000, 60, 2, 16, True),
    "g16_rel_l2_weights": (1800, 15000, 900, 1, 0, True),
    "g64_thirty_two_queries": (700, 5000, 20, 32, 16, False),      # the score form's largest query count, 8 column tiles
    "g16_three_nodes": (3, 7, 2, 1, 0, False),                    # fewer rows than one batch of four iterations
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_fused_layer_equals_rspmm_plus_epilogue(case):
    from ultra_torchdrug_amd import RelCSR, _lib, functional as UF
    dev = _dev()
    n, e, r, n_query, knob, weights = CASES[case]
    g = random_graph(seed=len(case), n_node=n, n_edge=e, n_rel=r, weights=weights)
    if n > 60:
        g["dst"][:40] = 7                                 # a row of three windows
        g["dst"][40:60] = n - 1                           # the last row: two windows
    keep = g["dst"] != (11 if n > 60 else 1)              # an empty row
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    csr = RelCSR(t(g["dst"][keep]), t(g["src"][keep]), t(g["rel"][keep]), t(g["w"][keep]) if weights else None, n, n, r,
                 wide_ids=True, piece_len=512)
    assert csr.fwd.row_ptr is not None and csr.fwd.n_pieces == 0
    gen = torch.Generator(device=dev).manual_seed(5)
    F = 64 * n_query
    x = torch.randn(n, n_query, 64, device=dev, generator=gen)
    relation = torch.randn(r, F, device=dev, generator=gen)
    b_node = torch.randint(0, n, (n_query,), device=dev, generator=gen).to(torch.int32)
    b_node[0] = min(7, n - 1)
    b_value = torch.randn(n_query, 64, device=dev, generator=gen)
    lib = _lib.load()
    lib.ultra_rspmm_force_general_path(knob)
    try:
        for layer_norm, relu, shortcut, with_boundary in ((True, True, True, True), (False, True, False, True),
                                                          (True, False, True, False), (True, True, False, True)):
            w, b, gamma, beta = _layer_params(gen, dev, layer_norm)
            boundary = (b_node, b_value) if with_boundary else None
            update = UF.rspmm_forward(csr, relation, x.flatten(1), "add", "mul", boundary=boundary).view(n, n_query, 64)
            want = UF.combine_forward(x, update, w, b, gamma, beta, 1e-5, relu, shortcut)
            got = UF.layer_forward(csr, relation, x, boundary, w, b, gamma, beta, 1e-5, relu, shortcut)
            assert got is not None, "the fused entry declined a row-per-group plan"
            torch.cuda.synchronize()
            assert torch.equal(got, want), (case, layer_norm, relu, shortcut, with_boundary,
                                            float((got - want).abs().max()), int((got != want).any(-1).sum()))
            again = UF.layer_forward(csr, relation, x, boundary, w, b, gamma, beta, 1e-5, relu, shortcut)
            assert torch.equal(got, again)
            # the last layer's form: the score head inside the same launch == layer + ultra_score_forward_f32
            query = torch.randn(n_query, 64, device=dev, generator=gen)
            w1 = torch.randn(128, 128, device=dev, generator=gen) * 0.1
            b1 = torch.randn(128, device=dev, generator=gen) * 0.1
            w2 = torch.randn(1, 128, device=dev, generator=gen) * 0.1
            b2 = torch.randn(1, device=dev, generator=gen)
            want_score = UF.score_all_entities(want, query, w1, b1, w2, b2)
            got_score = UF.layer_score_forward(csr, relation, x, boundary, w, b, gamma, beta, 1e-5, relu, shortcut, query, w1, b1, w2, b2)
            assert got_score is not None and got_score.shape == want_score.shape
            torch.cuda.synchronize()
            assert torch.equal(got_score, want_score), (case, float((got_score - want_score).abs().max()), int((got_score != want_score).sum()))
            assert torch.equal(got_score, UF.layer_score_forward(csr, relation, x, boundary, w, b, gamma, beta, 1e-5, relu, shortcut, query,
                                                                 w1, b1, w2, b2))
    finally:
        lib.ultra_rspmm_force_general_path(0)


def test_fused_layer_keeps_nan_and_inf_where_the_two_launches_do():
    """Non-finite inputs (a damaged checkpoint): `inf` / `NaN` in the gathered rows and in the relation table travel through the
    sum, the Linear, LayerNorm and -- `torch.relu` keeps a NaN -- the activation exactly as in the two launches; the score head's
    relu likewise.  NaN pattern and every other value equal."""
    from ultra_torchdrug_amd import RelCSR, functional as UF
    dev = _dev()
    n, e, r, n_query = 2000, 9000, 12, 2
    g = random_graph(seed=21, n_node=n, n_edge=e, n_rel=r)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    csr = RelCSR(t(g["dst"]), t(g["src"]), t(g["rel"]), None, n, n, r, wide_ids=True, piece_len=512)
    gen = torch.Generator(device=dev).manual_seed(8)
    x = torch.randn(n, n_query, 64, device=dev, generator=gen)
    x[5, 0, 3] = float("nan")
    x[9, 1, 60] = float("inf")
    x[11, 0, :] = float("-inf")
    relation = torch.randn(r, 64 * n_query, device=dev, generator=gen)
    relation[4, 70] = float("inf")
    boundary = (torch.tensor([5, 40], dtype=torch.int32, device=dev), torch.randn(n_query, 64, device=dev, generator=gen))
    same = lambda a, b: torch.equal(a.isnan(), b.isnan()) and torch.equal(torch.nan_to_num(a, nan=0.0), torch.nan_to_num(b, nan=0.0))
    for layer_norm, relu in ((True, True), (False, True), (True, False)):
        w, b, gamma, beta = _layer_params(gen, dev, layer_norm)
        update = UF.rspmm_forward(csr, relation, x.flatten(1), "add", "mul", boundary=boundary).view(n, n_query, 64)
        want = UF.combine_forward(x, update, w, b, gamma, beta, 1e-5, relu, True)
        got = UF.layer_forward(csr, relation, x, boundary, w, b, gamma, beta, 1e-5, relu, True)
        assert got is not None and bool(want.isnan().any()) and not bool(want.isnan().all())
        assert same(got, want)
        if relu:        # ... and torch's own chain keeps a NaN at the same places (relu(NaN) = NaN)
            ref = torch.cat([x, update], dim=-1) @ w.t() + b
            if layer_norm:
                ref = torch.nn.functional.layer_norm(ref, (64,), gamma, beta, 1e-5)
            assert torch.equal((torch.relu(ref) + x).isnan(), want.isnan())
        query = torch.randn(n_query, 64, device=dev, generator=gen)
        w1 = torch.randn(128, 128, device=dev, generator=gen) * 0.1
        b1 = torch.randn(128, device=dev, generator=gen) * 0.1
        w2 = torch.randn(1, 128, device=dev, generator=gen) * 0.1
        b2 = torch.randn(1, device=dev, generator=gen)
        want_score = UF.score_all_entities(want, query, w1, b1, w2, b2)
        got_score = UF.layer_score_forward(csr, relation, x, boundary, w, b, gamma, beta, 1e-5, relu, True, query, w1, b1, w2, b2)
        assert got_score is not None and same(got_score, want_score)
        feature = torch.cat([want.transpose(0, 1), query.unsqueeze(1).expand(-1, n, -1)], dim=-1)
        ref_score = (torch.relu(feature @ w1.t() + b1) @ w2.t() + b2).squeeze(-1)
        assert torch.equal(ref_score.isnan(), want_score.isnan())


def test_second_layer_with_remapped_sources_equals_the_plain_layer(monkeypatch):
    """After the sparse first layer all but the LISTED rows of the layer's output hold one constant vector; the second layer then
    gathers, for every edge whose source is not listed, ONE fixed unlisted row (`ultra_second_layer_sources`): the same values, so
    the same bits as the layer over the plan's own sources -- with hubs, repeated and isolated boundary nodes."""
    from ultra_torchdrug_amd import RelCSR, functional as UF
    dev = _dev()
    monkeypatch.setattr(UF, "SPARSE_FIRST_LAYER_MIN_ROWS", 0)
    n, e, r = 30000, 150000, 12
    g = random_graph(seed=9, n_node=n, n_edge=e, n_rel=r)
    g["src"][:25] = 17                                        # node 17: 25 out-edges
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    csr = RelCSR(t(g["dst"]), t(g["src"]), t(g["rel"]), None, n, n, r, wide_ids=True, piece_len=512)
    gen = torch.Generator(device=dev).manual_seed(4)
    for nodes in ([17, 3], [17, 17, 29999, 0]):
        Q = len(nodes)
        relation = torch.randn(r, 64 * Q, device=dev, generator=gen)
        relation2 = torch.randn(r, 64 * Q, device=dev, generator=gen)
        boundary = (torch.tensor(nodes, dtype=torch.int32, device=dev), torch.randn(Q, 64, device=dev, generator=gen))
        w, b, gamma, beta = _layer_params(gen, dev)
        first = UF.first_layer_forward(csr, relation, boundary, w, b, gamma, beta, 1e-5, True, True, want_list=True)
        assert first is not None
        hidden1, row_list, list_count = first
        sources = UF.second_layer_sources(csr, row_list, list_count, Q)
        assert sources is not None and sources.shape == csr.fwd.node_a.shape
        E = csr.n_edges
        plain_src = csr.fwd.node_a[:E]
        moved = sources[:E] != plain_src
        assert 0.9 * E < int(moved.sum()) < E                          # almost every edge gathers the one constant row ...
        assert len(torch.unique(sources[:E][moved])) == 1              # ... the same one
        assert torch.equal(hidden1[sources[:E].long()], hidden1[plain_src.long()])      # and it holds what the edge's own source holds
        w2, b2, gamma2, beta2 = _layer_params(gen, dev)
        want = UF.layer_forward(csr, relation2, hidden1, boundary, w2, b2, gamma2, beta2, 1e-5, True, True)
        got = UF.layer_forward(csr, relation2, hidden1, boundary, w2, b2, gamma2, beta2, 1e-5, True, True, sources=sources)
        assert torch.equal(got, want)


def test_fused_layer_declines_what_it_does_not_cover():
    from ultra_torchdrug_amd import RelCSR, functional as UF
    dev = _dev()
    g = random_graph(seed=2, n_node=500, n_edge=4000, n_rel=9)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    csr = RelCSR(t(g["dst"]), t(g["src"]), t(g["rel"]), None, 500, 500, 9)          # a packed plan: the chunked kernels
    gen = torch.Generator(device=dev).manual_seed(1)
    w, b, gamma, beta = _layer_params(gen, dev)
    x = torch.randn(500, 2, 64, device=dev, generator=gen)
    relation = torch.randn(9, 128, device=dev, generator=gen)
    assert UF.layer_forward(csr, relation, x, None, w, b, gamma, beta) is None

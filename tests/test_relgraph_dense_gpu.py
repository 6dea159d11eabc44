"""GPU parity of the DENSE relation-graph form (csrc/relgraph_dense.hip; include/ultra_rspmm.h, ABI 7).

construct_relation_graph (/root/reference/ultra/rel_model.py:99-143) yields a dense graph over 2R nodes and 4 edge types with
unit weights; for such plans the library runs the sum aggregations as a product with a 0/1 matrix on the exact-f32 matrix
cores.  Bars: forward and d_input are IDENTICAL to the oracle in the REFERENCE order (``piece = 0``: strictly sequential per
row) for every row -- no pieces; d_relation is identical to the oracle's restatement of the kernel's documented order and within
rounding of the reference order; the knob that walks the edge list instead gives the plans' piece order as before.
"""
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


def _graph(seed, n, density):
    """Unit-weight graph over n nodes and 4 edge types; density 1.0 = the complete graph (the FB15k237-shaped relation graph)."""
    if density >= 1.0:
        dst, src, rel = np.meshgrid(np.arange(n), np.arange(n), np.arange(4), indexing="ij")
        perm = np.random.default_rng(seed).permutation(dst.size)             # any edge order, as the callers pass it
        return dict(dst=dst.ravel()[perm].astype(np.int64), src=src.ravel()[perm].astype(np.int64),
                    rel=rel.ravel()[perm].astype(np.int64), w=None)
    return random_graph(seed, n, int(density * n * n * 4 * 1.6), 4, unique=True)


def _relcsr(g, n):
    from ultra_torchdrug_amd import RelCSR
    return RelCSR(_t(g["dst"]), _t(g["src"]), _t(g["rel"]), None, n, n, 4)


CASES = {
    # name: (nodes, density, F)
    "complete_48": (48, 1.0, 64),
    "complete_ragged_rows": (90, 1.0, 128),            # 90 = 5 tiles + 10 rows; sources padded to 96
    "half_dense": (120, 0.5, 1024),
    "sparse_enough": (150, 0.15, 192),                  # F = 3 query blocks: the last workgroup's tiles end early
    "narrow_F": (64, 0.6, 16),
    "odd_tiles_F": (70, 0.4, 48),
}


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("mul", ["mul", "add"])
def test_dense_forward_is_the_reference_order_for_every_row(oracle, case, mul):
    from ultra_torchdrug_amd import functional as UF
    n, density, F = CASES[case]
    g = _graph(3, n, density)
    csr = _relcsr(g, n)
    assert csr.dense_form and csr.fwd.dense is not None and csr.kernel_order("add", mul, F) == (0, mul == "mul")
    rng = np.random.default_rng(11)
    relation, x = rng.standard_normal((4, F)).astype(np.float32), rng.standard_normal((n, F)).astype(np.float32)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], None, n, n, 4)
    want = oracle.rspmm_forward(csr_o, relation, x, "add", mul, piece=0)
    got = UF.rspmm_forward(csr, _t(relation), _t(x), "add", mul)
    assert np.array_equal(got.cpu().numpy(), want)
    # fused epilogues: dense rows and the sparse Bellman-Ford boundary
    add = rng.standard_normal((n, F)).astype(np.float32)
    got_add = UF.rspmm_forward(csr, _t(relation), _t(x), "add", mul, add_rows=_t(add))
    assert np.array_equal(got_add.cpu().numpy(), want + add)
    if F % 64 == 0:
        q = F // 64
        node = rng.integers(0, n, q).astype(np.int32)
        value = rng.standard_normal((q, 64)).astype(np.float32)
        dense_b = np.zeros((n, q, 64), dtype=np.float32)
        dense_b[node, np.arange(q)] = value
        got_b = UF.rspmm_forward(csr, _t(relation), _t(x), "add", mul, boundary=(_t(node), _t(value)))
        assert np.array_equal(got_b.cpu().numpy(), want + dense_b.reshape(n, F))
    # min / max keep walking the edge list (and do not depend on the order)
    for s in ("min", "max"):
        got_m = UF.rspmm_forward(csr, _t(relation), _t(x), s, mul)
        assert np.array_equal(got_m.cpu().numpy(), oracle.rspmm_forward(csr_o, relation, x, s, mul, piece=0))


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("mul", ["mul", "add"])
def test_dense_backward(oracle, case, mul):
    """d_input: the reference order, bit for bit (also when accumulated into the epilogue's gradient); d_relation (mul = mul): the
    documented dense order bit for bit and the reference order within rounding; mul = add: d_relation walks the edge list."""
    from ultra_torchdrug_amd import functional as UF
    n, density, F = CASES[case]
    g = _graph(5, n, density)
    csr = _relcsr(g, n)
    rng = np.random.default_rng(13)
    relation, x = rng.standard_normal((4, F)).astype(np.float32), rng.standard_normal((n, F)).astype(np.float32)
    grad = rng.standard_normal((n, F)).astype(np.float32)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], None, n, n, 4)
    out_o = oracle.rspmm_forward(csr_o, relation, x, "add", mul, piece=0)
    d_rel_seq, d_x_seq = oracle.rspmm_backward(csr_o, relation, x, out_o, grad, "add", mul, piece=0)
    d_x, d_rel = UF.rspmm_backward(csr, _t(relation), _t(x), None, _t(grad), "add", mul)
    assert np.array_equal(d_x.cpu().numpy(), d_x_seq)
    if mul == "mul":
        d_rel_k, _ = oracle.rspmm_backward(csr_o, relation, x, out_o, grad, "add", mul, piece=0, dense_relation=True)
        assert np.array_equal(d_rel.cpu().numpy(), d_rel_k), "d_relation differs from the documented dense order"
    else:
        d_rel_k, _ = oracle.rspmm_backward(csr_o, relation, x, out_o, grad, "add", mul, piece=csr.piece_len)
        assert np.array_equal(d_rel.cpu().numpy(), d_rel_k)
    # against the reference order: the same terms in another association -- within sqrt(n) * 2^-24 of the sum of |terms|
    terms = oracle.rspmm_backward(csr_o, np.abs(relation), np.abs(x), out_o, np.abs(grad), "add", mul, piece=0)[0]
    bound = 8 * np.sqrt(csr_o.n_edges / 4) * 2.0 ** -24 * terms + 1e-6
    assert (np.abs(d_rel.cpu().numpy() - d_rel_seq) <= bound).all()
    # autograd through the operator, and the accumulate form of d_input
    extra = rng.standard_normal((n, F)).astype(np.float32)
    acc = _t(extra.copy())
    d_x2, _ = UF.rspmm_backward(csr, _t(relation), _t(x), None, _t(grad), "add", mul, need_relation=False, d_input_add=acc)
    assert d_x2.data_ptr() == acc.data_ptr() and np.array_equal(d_x2.cpu().numpy(), d_x_seq + extra)
    rel_t, x_t = _t(relation).requires_grad_(), _t(x).requires_grad_()
    UF.generalized_rspmm(csr, rel_t, x_t, sum="add", mul=mul).backward(_t(grad))
    assert torch.equal(x_t.grad, d_x) and torch.equal(rel_t.grad, d_rel)


def test_the_knob_walks_the_edge_list_of_a_dense_plan(oracle):
    import ultra_torchdrug_amd as U
    from ultra_torchdrug_amd import functional as UF
    n, F = 100, 128
    g = _graph(9, n, 0.7)
    csr = _relcsr(g, n)
    rng = np.random.default_rng(1)
    relation, x = rng.standard_normal((4, F)).astype(np.float32), rng.standard_normal((n, F)).astype(np.float32)
    grad = rng.standard_normal((n, F)).astype(np.float32)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], None, n, n, 4)
    want = oracle.rspmm_forward(csr_o, relation, x, "add", "mul", piece=csr.piece_len)
    d_rel_p, d_x_p = oracle.rspmm_backward(csr_o, relation, x, want, grad, "add", "mul", piece=csr.piece_len)
    lib = U.require_library()
    lib.ultra_rspmm_force_general_path(64)
    try:
        got = UF.rspmm_forward(csr, _t(relation), _t(x), "add", "mul")
        d_x, d_rel = UF.rspmm_backward(csr, _t(relation), _t(x), None, _t(grad), "add", "mul")
    finally:
        lib.ultra_rspmm_force_general_path(0)
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(d_x.cpu().numpy(), d_x_p) and np.array_equal(d_rel.cpu().numpy(), d_rel_p)


def test_weights_and_other_vocabularies_keep_the_edge_list(oracle):
    """Per-edge weights (an edge removed from a training batch: weight 0) and graphs that are not 4-type / not dense enough never
    take the dense form."""
    from ultra_torchdrug_amd import RelCSR, functional as UF
    n, F = 80, 64
    g = _graph(2, n, 0.5)
    csr = _relcsr(g, n)
    assert csr.dense_form
    w = np.ones(len(g["dst"]), dtype=np.float32)
    w[::7] = 0.0
    reweighted = csr.with_edge_weights(_t(w))
    assert not reweighted.dense_form and reweighted.kernel_order("add", "mul", F) == (csr.piece_len, False)
    rng = np.random.default_rng(4)
    relation, x = rng.standard_normal((4, F)).astype(np.float32), rng.standard_normal((n, F)).astype(np.float32)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], w, n, n, 4)
    got = UF.rspmm_forward(reweighted, _t(relation), _t(x), "add", "mul")
    assert np.array_equal(got.cpu().numpy(), oracle.rspmm_forward(csr_o, relation, x, "add", "mul", piece=csr.piece_len))
    thin = random_graph(1, 400, 1500, 4, unique=True)
    assert not RelCSR(_t(thin["dst"]), _t(thin["src"]), _t(thin["rel"]), None, 400, 400, 4).dense_form
    five = random_graph(1, 50, 9000, 5, unique=True)
    assert not RelCSR(_t(five["dst"]), _t(five["src"]), _t(five["rel"]), None, 50, 50, 5).dense_form


def test_relation_graph_of_a_kg_takes_the_dense_form_and_matches_the_reference_order(oracle):
    """The real thing: construct_relation_graph on a seeded KG (rel_model.py:99-143), B = 16 queries x 64 d, the layer's own
    operands -- the relation stack's rspmm in the reference order for all 2R rows."""
    from ultra_torchdrug_amd import functional as UF
    from ultra_torchdrug_amd.data import synthetic_kg
    from ultra_torchdrug_amd.rel_model import construct_relation_graph
    rel_graph = construct_relation_graph(synthetic_kg("S-codexs", device=_dev()))
    csr = rel_graph.relcsr
    n = rel_graph.num_node
    assert csr.shape == (n, n, 4) and csr.dense_form
    F = 16 * 64
    rng = np.random.default_rng(8)
    relation = np.tile(rng.standard_normal((4, 64)).astype(np.float32), (1, 16))          # layer.py:125-126
    x = rng.standard_normal((n, F)).astype(np.float32)
    csr_o = oracle.coalesce_csr(csr.dst.cpu().numpy(), csr.src.cpu().numpy(), csr.rel_id.cpu().numpy(), None, n, n, 4)
    got = UF.rspmm_forward(csr, _t(relation), _t(x), "add", "mul")
    assert np.array_equal(got.cpu().numpy(), oracle.rspmm_forward(csr_o, relation, x, "add", "mul", piece=0))


def test_first_layer_frontier_on_a_dense_graph_sums_without_pieces(oracle):
    """The frontier kernel (first Bellman-Ford layer: only the boundary nodes' out-edges) reproduces the FULL kernels' association;
    on a dense-form graph that is the sequential order, also where a source's parallel edges straddle what would be a piece
    boundary of a long row (found by the FB15k237Inductive-v1-shaped end-to-end test: one ulp)."""
    from ultra_torchdrug_amd import functional as UF
    n, q = 150, 6
    g = _graph(12, n, 0.5)                    # ~300 in-edges per row: every row would be split into pieces of 128
    csr = _relcsr(g, n)
    assert csr.dense_form and int(torch.bincount(csr.dst).max()) > csr.piece_len
    rng = np.random.default_rng(3)
    relation = rng.standard_normal((4, q * 64)).astype(np.float32)
    node = rng.integers(0, n, q).astype(np.int32)
    value = rng.standard_normal((q, 64)).astype(np.float32)
    dense_b = np.zeros((n, q, 64), dtype=np.float32)
    dense_b[node, np.arange(q)] = value
    dense_b = dense_b.reshape(n, q * 64)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], None, n, n, 4)
    want = oracle.rspmm_forward(csr_o, relation, dense_b, "add", "mul", piece=0) + dense_b
    got = UF.rspmm_frontier(csr, _t(relation), (_t(node), _t(value)))
    assert np.array_equal(got.cpu().numpy(), want)
    full = UF.rspmm_forward(csr, _t(relation), _t(dense_b), "add", "mul", boundary=(_t(node), _t(value)))
    assert torch.equal(got, full)


@pytest.mark.parametrize("n,density,q", [(90, 1.0, 3), (120, 0.5, 16), (474, 1.0, 16), (70, 0.3, 1)])
@pytest.mark.parametrize("norm,relu,shortcut", [(True, True, True), (False, True, False), (True, False, True)])
def test_fused_dense_layer_equals_rspmm_plus_epilogue(n, density, q, norm, relu, shortcut):
    """ultra_dense_layer_forward_f32 (a whole relation-graph layer in one launch) == the dense rspmm with the sparse boundary
    followed by the fused epilogue kernel, bit for bit (which the other tests of this file and test_model_gpu hold to the oracle)."""
    from ultra_torchdrug_amd import functional as UF
    dev = _dev()
    g = _graph(21, n, density)
    csr = _relcsr(g, n)
    gen = torch.Generator(device=dev).manual_seed(n + q)
    hidden = torch.randn(n, q, 64, device=dev, generator=gen)
    relation = torch.randn(4, q * 64, device=dev, generator=gen)
    node = torch.randint(0, n, (q,), device=dev, generator=gen).to(torch.int32)
    value = torch.randn(q, 64, device=dev, generator=gen)
    weight, bias = torch.randn(64, 128, device=dev, generator=gen) * 0.2, torch.randn(64, device=dev, generator=gen)
    ln = (torch.rand(64, device=dev, generator=gen) + 0.5, torch.randn(64, device=dev, generator=gen)) if norm else (None, None)
    args = (weight, bias, ln[0], ln[1], 1e-5, relu, shortcut)
    update = UF.rspmm_forward(csr, relation, hidden.flatten(1), "add", "mul", boundary=(node, value))
    want = UF.combine_forward(hidden, update.view(n, q, 64), *args)
    got = UF.dense_layer_forward(csr, relation, hidden, (node, value), *args)
    assert got is not None and got.data_ptr() != hidden.data_ptr()
    assert torch.equal(got, want), "fused layer differs by %.3g" % (got - want).abs().max().item()

"""CPU: the oracle against (i) a hand-derived closed-form case, (ii) the reference's own materialised O(E)
definition (ultra/layer.py:232-296) restated in numpy fp64, (iii) algebraic properties, (iv) finite differences,
(v) the committed seeded golden vectors.  No GPU, no product code."""
import json
import os
import sys

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from graphs import random_graph

HERE = os.path.dirname(os.path.abspath(__file__))
SUMS = ["add", "min", "max"]
MULS = ["mul", "add"]


def _num(a):
    special = {"flt_max": np.finfo(np.float32).max, "-flt_max": np.finfo(np.float32).min}
    return np.array([[special[v] if isinstance(v, str) else float(v) for v in row] for row in a], dtype=np.float32)


def test_hand_computed_case(oracle):
    """x = [[1,2],[3,4],[5,6],[7,8]], rel0 = (2,-1), rel1 = (0.5,3); edges (dst,src,rel,w):
    (0,1,0,1) (0,2,1,2) (1,1,1,1) (2,0,0,1)+(2,0,0,0.5) -> merged 1.5, (2,3,1,1); node 3 has no in-edge.
    add/mul row 0: 1*(2*3, -1*4) + 2*(0.5*5, 3*6) = (6,-4)+(5,36) = (11,32);  row 2: 1.5*(2,-2)+(3.5,24) = (6.5,21).
    add/add row 0: (2+3,-1+4) + 2*(0.5+5, 3+6) = (5,3)+(11,18) = (16,21).  max/mul row 0: max((6,-4),(5,36)) = (6,36).
    d_x[1] = rel0*g0 + rel1*g1 = (2,-1)+(1,0) = (3,-1);  d_rel[1] = 2*x2*g0 + x1*g1 + x3*g2 = (10,12)+(6,0)+(0,-8)."""
    case = json.load(open(os.path.join(HERE, "golden", "handcomputed.json")))
    e = np.array(case["edges_dst_src_rel_w"], dtype=np.float64)
    dst, src, rel, w = e[:, 0].astype(np.int64), e[:, 1].astype(np.int64), e[:, 2].astype(np.int64), e[:, 3].astype(np.float32)
    n, r = case["n_node"], case["n_rel"]
    x, relation = _num(case["x"]), _num(case["relation"])
    csr = oracle.coalesce_csr(dst, src, rel, w, n, n, r)
    assert csr.n_edges == 5 and np.isclose(csr.w, [1, 2, 1, 1.5, 1]).all()
    for key, want in case["forward"].items():
        s, m = key.split("_")
        for piece in (0, 1, 2):
            got = oracle.rspmm_forward(csr, relation, x, s, m, piece=piece)
            assert np.array_equal(got, _num(want)), (key, piece)
        py = np.array(oracle.rspmm_python(csr.row, csr.col, csr.rel, csr.w, relation, x, n, s, m), dtype=np.float32)
        assert np.array_equal(py, _num(want)), key
    grad = _num(case["grad"])
    out = oracle.rspmm_forward(csr, relation, x, "add", "mul")
    d_rel, d_x = oracle.rspmm_backward(csr, relation, x, out, grad, "add", "mul")
    assert np.array_equal(d_x, _num(case["backward_add_mul"]["d_x"]))
    assert np.array_equal(d_rel, _num(case["backward_add_mul"]["d_relation"]))


@pytest.mark.parametrize("sum", SUMS)
@pytest.mark.parametrize("mul", MULS)
@pytest.mark.parametrize("weights", [False, True])
def test_c_oracle_equals_materialised_definition(oracle, sum, mul, weights):
    n, r, F = 120, 6, 24
    g = random_graph(seed=5, n_node=n, n_edge=2500, n_rel=r, skew=True, unique=True, weights=weights, hub_row=3,
                     hub_edges=300, isolated=5)
    rng = np.random.default_rng(0)
    relation = rng.standard_normal((r, F)).astype(np.float32)
    x = rng.standard_normal((n, F)).astype(np.float32)
    csr = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
    want = oracle.rspmm_materialised(g["dst"], g["src"], g["rel"], g["w"], relation, x, n, sum, mul, dtype=np.float64)
    for piece in (0, 16, 128):
        got = oracle.rspmm_forward(csr, relation, x, sum, mul, piece=piece)
        assert np.array_equal(np.isfinite(got), np.isfinite(want))
        fin = np.isfinite(want)
        np.testing.assert_allclose(got[fin], want[fin], rtol=2e-5, atol=2e-5)
    seq = oracle.rspmm_forward(csr, relation, x, sum, mul, piece=0)
    pc = oracle.rspmm_forward(csr, relation, x, sum, mul, piece=16)
    short = np.diff(csr.row_ptr) <= 16
    assert np.array_equal(seq[short], pc[short])          # rows that are not split: identical in both orders
    assert short.sum() < n                                # ... and the case does contain split rows


def test_duplicates_are_merged_by_weight_sum(oracle):
    dst = np.array([0, 0, 0, 1]); src = np.array([1, 1, 1, 0]); rel = np.array([0, 0, 0, 0])
    w = np.array([0.25, 0.5, 2.0, 1.0], dtype=np.float32)
    csr = oracle.coalesce_csr(dst, src, rel, w, 2, 2, 1)
    assert csr.n_edges == 2 and csr.w[0] == np.float32(2.75)
    x = np.array([[3.0], [4.0]], dtype=np.float32); relation = np.array([[2.0]], dtype=np.float32)
    assert oracle.rspmm_forward(csr, relation, x, "max", "mul")[0, 0] == np.float32(2.75 * 8)   # merged, then max


def test_dense_einsum_equivalence(oracle):
    """rspmm(A, rel, x) == sum_r A_r @ (x * rel_r) for a tiny dense adjacency (SURVEY.md 8c iii)."""
    rng = np.random.default_rng(3)
    n, r, F = 9, 3, 5
    A = (rng.random((n, n, r)) < 0.3) * rng.uniform(0.5, 2, (n, n, r))
    dst, src, rel = np.nonzero(A)
    w = A[dst, src, rel].astype(np.float32)
    x = rng.standard_normal((n, F)).astype(np.float32); relation = rng.standard_normal((r, F)).astype(np.float32)
    csr = oracle.coalesce_csr(dst, src, rel, w, n, n, r)
    want = np.einsum("vur,rf,uf->vf", A.astype(np.float64), relation.astype(np.float64), x.astype(np.float64))
    np.testing.assert_allclose(oracle.rspmm_forward(csr, relation, x), want, rtol=1e-5, atol=1e-5)


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(1, 40), e=st.integers(0, 300), r=st.integers(1, 5), F=st.integers(1, 9))
def test_properties(seed, n, e, r, F):
    from oracle import oracle
    g = random_graph(seed=seed, n_node=n, n_edge=e, n_rel=r, unique=True, weights=True)
    rng = np.random.default_rng(seed)
    relation = rng.standard_normal((r, F)).astype(np.float32); x = rng.standard_normal((n, F)).astype(np.float32)
    csr = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
    perm = rng.permutation(len(g["dst"]))
    csr_p = oracle.coalesce_csr(g["dst"][perm], g["src"][perm], g["rel"][perm], g["w"][perm], n, n, r)
    for s in SUMS:
        a = oracle.rspmm_forward(csr, relation, x, s, "mul")
        assert np.array_equal(a, oracle.rspmm_forward(csr_p, relation, x, s, "mul"))      # edge order is irrelevant
    # linear in the weights (exact: scaling by 2 commutes with fp32 rounding)
    csr2 = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"] * 2, n, n, r)
    assert np.array_equal(oracle.rspmm_forward(csr2, relation, x), 2 * oracle.rspmm_forward(csr, relation, x))
    # empty rows carry the identity
    empty = np.diff(csr.row_ptr) == 0
    assert (oracle.rspmm_forward(csr, relation, x, "max")[empty] == np.finfo(np.float32).min).all()
    assert (oracle.rspmm_forward(csr, relation, x, "min")[empty] == np.finfo(np.float32).max).all()
    assert (oracle.rspmm_forward(csr, relation, x, "add")[empty] == 0).all()


@pytest.mark.parametrize("mul", MULS)
def test_backward_against_finite_differences(oracle, mul):
    """sum=add is linear in each argument: central differences in fp64 on the materialised definition."""
    n, r, F = 20, 3, 4
    g = random_graph(seed=8, n_node=n, n_edge=150, n_rel=r, unique=True, weights=True)
    rng = np.random.default_rng(1)
    relation = rng.standard_normal((r, F)); x = rng.standard_normal((n, F)); grad = rng.standard_normal((n, F))
    csr = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
    out = oracle.rspmm_forward(csr, relation, x, "add", mul)
    d_rel, d_x, d_w = oracle.rspmm_backward(csr, relation, x, out, grad, "add", mul, need_weight_grad=True)

    def loss(rel_, x_, w_=None):
        w_ = g["w"] if w_ is None else w_
        return (oracle.rspmm_materialised(g["dst"], g["src"], g["rel"], w_, rel_, x_, n, "add", mul, np.float64) * grad).sum()

    eps = 1e-4
    for (i, f) in [(0, 0), (5, 3), (19, 1)]:
        d = np.zeros_like(x); d[i, f] = eps
        assert abs((loss(relation, x + d) - loss(relation, x - d)) / (2 * eps) - d_x[i, f]) < 1e-3
    for (i, f) in [(0, 0), (2, 3)]:
        d = np.zeros_like(relation); d[i, f] = eps
        assert abs((loss(relation + d, x) - loss(relation - d, x)) / (2 * eps) - d_rel[i, f]) < 1e-3
    key_o = (csr.row.astype(np.int64) * n + csr.col) * r + csr.rel
    key_in = (g["dst"] * n + g["src"]) * r + g["rel"]
    pos = np.searchsorted(key_o, key_in)
    for k in (0, 7, 100):
        d = np.zeros(len(g["w"])); d[k] = eps
        fd = (loss(relation, x, g["w"] + d) - loss(relation, x, g["w"] - d)) / (2 * eps)
        assert abs(fd - d_w[pos[k]]) < 1e-3


@pytest.mark.parametrize("sum", ["min", "max"])
def test_min_max_backward_routes_gradient_to_the_selected_edge(oracle, sum):
    n, r, F = 30, 3, 6
    g = random_graph(seed=4, n_node=n, n_edge=200, n_rel=r, unique=True)
    rng = np.random.default_rng(2)
    relation = rng.standard_normal((r, F)).astype(np.float32); x = rng.standard_normal((n, F)).astype(np.float32)
    grad = rng.standard_normal((n, F)).astype(np.float32)
    csr = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], None, n, n, r)
    out = oracle.rspmm_forward(csr, relation, x, sum, "mul")
    d_rel, d_x = oracle.rspmm_backward(csr, relation, x, out, grad, sum, "mul")
    # independent restatement: for every (row, f) the arg-extreme edge receives g * d(mul)
    want_dx = np.zeros((n, F)); want_drel = np.zeros((r, F))
    row = csr.row
    for f in range(F):
        y = relation[csr.rel, f] * x[csr.col, f]
        for k in range(csr.n_edges):
            if y[k] == out[row[k], f]:
                want_dx[csr.col[k], f] += grad[row[k], f] * relation[csr.rel[k], f]
                want_drel[csr.rel[k], f] += grad[row[k], f] * x[csr.col[k], f]
    np.testing.assert_allclose(d_x, want_dx, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(d_rel, want_drel, rtol=1e-5, atol=1e-5)


def test_seeded_golden_vectors(oracle):
    """The oracle must keep reproducing the committed vectors bit for bit (tests/golden/make_golden.py)."""
    z = np.load(os.path.join(HERE, "golden", "rspmm_seeded.npz"))
    n, r, piece = int(z["n_node"]), int(z["n_rel"]), int(z["piece"])
    csr = oracle.coalesce_csr(z["dst"], z["src"], z["rel"], z["w"], n, n, r)
    for s in SUMS:
        for m in MULS:
            fwd = oracle.rspmm_forward(csr, z["relation"], z["x"], s, m, piece=piece)
            assert np.array_equal(fwd, z["fwd_%s_%s" % (s, m)])
            d_rel, d_x = oracle.rspmm_backward(csr, z["relation"], z["x"], fwd, z["grad"], s, m, piece=piece)
            assert np.array_equal(d_rel, z["drel_%s_%s" % (s, m)]) and np.array_equal(d_x, z["dx_%s_%s" % (s, m)])


def test_filtered_rank(oracle):
    """ultra/task.py:307-315."""
    rng = np.random.default_rng(0)
    pred = rng.standard_normal((7, 50)).astype(np.float32)
    pred[0, 3] = pred[0, 9]                              # a tie counts against the positive (<=)
    mask = rng.random((7, 50)) < 0.8
    target = rng.integers(0, 50, 7)
    target[0] = 3
    mask[np.arange(7), target] = True
    pos = pred[np.arange(7), target][:, None]
    want = ((pos <= pred) & mask).sum(-1) + 1
    assert np.array_equal(oracle.filtered_rank(pred, mask, target), want)


def test_end_to_end_fixture_oracle_path(oracle):
    """tests/golden/e2e_codexs.json (tests/golden/make_e2e_golden.py): the oracle path must keep producing the
    committed int64 ranks and score checksums of the seeded S-codexs task -- in the reference's order and in the kernels' order."""
    import json
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_e2e_golden as G
    want = json.load(open(os.path.join(HERE, "golden", "e2e_codexs.json")))
    for name, piece in (("reference_order", 0), ("kernel_order", None)):
        pred, ranks = G.oracle_side(piece)
        got = G.summary(pred, ranks)
        assert list(pred.shape) == want["score_shape"]
        assert got["ranks"] == want[name]["ranks"], name
        assert got["scores_sha256"] == want[name]["scores_sha256"], name
    assert want["reference_order"]["ranks"] == want["kernel_order"]["ranks"]     # the orders differ in the last bits only

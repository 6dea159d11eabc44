"""GPU tests of the PyTorch-ROCm C++ extension (csrc/torch_ext.cpp): ``torch.ops.ultra_mi.*`` against the oracle.

``ultra_torchdrug_amd.functional`` routes the plan-based operator through ``torch.ops.ultra_mi.rspmm_plan_fwd/bwd`` by
default, so tests/test_rspmm_gpu.py already exercises those two through the dispatcher; here: the raw-CSR operators of
SURVEY.md 8b (``build_relcsr``, ``rspmm_fwd`` with its registered autograd, ``rspmm_bwd``) and the equality of the two
bindings (dispatcher vs ctypes) on the same plan.
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


@pytest.mark.parametrize("weights", [False, True])
def test_build_relcsr_and_raw_rspmm_ops_match_oracle(oracle, weights):
    from ultra_torchdrug_amd import _torch_ext
    ops = _torch_ext.load()
    n, r, F = 600, 11, 128
    g = random_graph(seed=3, n_node=n, n_edge=8000, n_rel=r, weights=weights, skew=True, isolated=30)
    # torchdrug edge_list rows are (node_in, node_out, relation); rspmm aggregates over node_out
    edge_list = _t(np.stack([g["src"], g["dst"], g["rel"]], axis=1))
    w = None if g["w"] is None else _t(g["w"])
    row_ptr, src, rel, wt, edge_of_input = ops.build_relcsr(edge_list, w, n, r)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
    assert row_ptr.dtype == torch.int32 and np.array_equal(row_ptr.cpu().numpy(), csr_o.row_ptr)
    assert np.array_equal(src.cpu().numpy(), csr_o.col) and np.array_equal(rel.cpu().numpy(), csr_o.rel)
    want_w = np.ones(csr_o.n_edges, dtype=np.float32) if csr_o.w is None else csr_o.w
    assert np.array_equal(wt.cpu().numpy(), want_w)
    assert edge_of_input.shape == (8000,) and int(edge_of_input.max()) == csr_o.n_edges - 1

    rng = np.random.default_rng(2)
    relation = rng.standard_normal((r, F)).astype(np.float32)
    x = rng.standard_normal((n, F)).astype(np.float32)
    grad = rng.standard_normal((n, F)).astype(np.float32)
    for s, s_name in enumerate(["add", "min", "max"]):
        for m, m_name in enumerate(["mul", "add"]):
            rel_t, x_t = _t(relation).requires_grad_(), _t(x).requires_grad_()
            out = ops.rspmm_fwd(row_ptr, src, rel, wt, rel_t, x_t, s, m)
            want = oracle.rspmm_forward(csr_o, relation, x, s_name, m_name, piece=0)     # sequential: the reference order
            assert np.array_equal(out.detach().cpu().numpy(), want), (s_name, m_name)
            out.backward(_t(grad))                                                          # registered Autograd kernel
            d_rel_o, d_x_o = oracle.rspmm_backward(csr_o, relation, x, want, grad, s_name, m_name, piece=256)
            assert np.array_equal(x_t.grad.cpu().numpy(), d_x_o), (s_name, m_name)
            assert np.array_equal(rel_t.grad.cpu().numpy(), d_rel_o), (s_name, m_name)
    with pytest.raises(RuntimeError, match="no CPU kernel"):
        ops.rspmm_fwd(row_ptr, src, rel, wt, _t(relation), torch.from_numpy(x), 0, 0)
    with pytest.raises(RuntimeError, match="unknown sum/mul"):
        ops.rspmm_fwd(row_ptr, src, rel, wt, _t(relation), _t(x), 3, 0)


def test_dispatcher_binding_equals_ctypes_binding(monkeypatch):
    """The same plan through torch.ops.ultra_mi.rspmm_plan_fwd / _bwd and through ctypes: identical tensors."""
    from ultra_torchdrug_amd import RelCSR, functional as UF
    n, r, F = 500, 30, 192
    g = random_graph(seed=9, n_node=n, n_edge=20000, n_rel=r, skew=True, hub_row=5, hub_edges=3000, weights=True)
    csr = RelCSR(_t(g["dst"]), _t(g["src"]), _t(g["rel"]), _t(g["w"]), n, n, r)
    gen = torch.Generator(device=_dev()).manual_seed(1)
    relation = torch.randn(r, F, device=_dev(), generator=gen)
    x = torch.randn(n, F, device=_dev(), generator=gen)
    grad = torch.randn(n, F, device=_dev(), generator=gen)
    node = torch.tensor([3, 5, 499], dtype=torch.int32, device=_dev())
    value = torch.randn(3, 64, device=_dev(), generator=gen)
    results = {}
    for binding in ("torch", "ctypes"):
        monkeypatch.setenv("ULTRA_BINDING", binding)
        outs = [UF.rspmm_forward(csr, relation, x, s, m) for s in ("add", "min", "max") for m in ("mul", "add")]
        outs.append(UF.rspmm_forward(csr, relation, x, "add", "mul", boundary=(node, value)))
        outs.append(UF.rspmm_forward(csr, relation, x, "max", "add", add_rows=grad))
        for s in ("add", "max"):
            out = UF.rspmm_forward(csr, relation, x, s, "mul")
            outs.extend(UF.rspmm_backward(csr, relation, x, out, grad, s, "mul"))
        outs.append(UF.rspmm_backward(csr, relation, x, None, grad, "add", "add", need_relation=False)[0])
        results[binding] = outs
    assert len(results["torch"]) == len(results["ctypes"]) == 13
    for a, b in zip(results["torch"], results["ctypes"]):
        assert torch.equal(a, b)


def test_plan_ops_capture_into_a_hip_graph():
    """The dispatcher ops allocate through the caching allocator and take the current stream: capturable."""
    from ultra_torchdrug_amd import RelCSR, functional as UF
    n, r, F = 400, 6, 128
    g = random_graph(seed=4, n_node=n, n_edge=5000, n_rel=r)
    csr = RelCSR(_t(g["dst"]), _t(g["src"]), _t(g["rel"]), None, n, n, r)
    relation = torch.randn(r, F, device=_dev())
    x = torch.randn(n, F, device=_dev())
    want = UF.rspmm_forward(csr, relation, x)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    static_x = x.clone()
    with torch.cuda.graph(graph):
        static_out = UF.rspmm_forward(csr, relation, static_x)
    static_x.copy_(2 * x)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, 2 * want)


def test_integration_snippet_replaces_torchdrug_generalized_rspmm(oracle):
    """INTEGRATION.md 1a verbatim: torchdrug's `generalized_rspmm(sparse, relation, input, sum=, mul=)` rebuilt from the
    extension's raw operators (build_relcsr + rspmm_fwd), given the un-coalesced `adjacency.transpose(0, 1)` the
    reference hands over (ultra/layer.py:127,328) -- forward and autograd, against the package's cached-plan operator."""
    from ultra_torchdrug_amd import _torch_ext, functional as UF
    _torch_ext.load()
    SUM = {"add": 0, "min": 1, "max": 2}; MUL = {"mul": 0, "add": 1}

    def generalized_rspmm(sparse, relation, input, sum="add", mul="mul"):
        edge_list = sparse._indices()[[1, 0, 2]].t()
        row_ptr, src, rel, w, _ = torch.ops.ultra_mi.build_relcsr(edge_list.contiguous(), sparse._values(),
                                                                  sparse.shape[0], sparse.shape[2])
        return torch.ops.ultra_mi.rspmm_fwd(row_ptr, src, rel, w, relation, input, SUM[sum], MUL[mul])

    n, r, F = 300, 6, 64
    g = random_graph(seed=12, n_node=n, n_edge=4000, n_rel=r, weights=True)            # duplicates: coalesce must merge
    adjacency = torch.sparse_coo_tensor(torch.stack([_t(g["src"]), _t(g["dst"]), _t(g["rel"])]), _t(g["w"]), (n, n, r))
    sparse = adjacency.transpose(0, 1)                                                   # (dst, src, rel), not coalesced
    gen = torch.Generator(device=_dev()).manual_seed(2)
    relation = torch.randn(r, F, device=_dev(), generator=gen)
    x = torch.randn(n, F, device=_dev(), generator=gen)
    grad = torch.randn(n, F, device=_dev(), generator=gen)
    for s in SUM:
        for m in MUL:
            a_rel, a_x = relation.clone().requires_grad_(), x.clone().requires_grad_()
            b_rel, b_x = relation.clone().requires_grad_(), x.clone().requires_grad_()
            got = generalized_rspmm(sparse, a_rel, a_x, sum=s, mul=m)
            want = UF.generalized_rspmm(sparse, b_rel, b_x, sum=s, mul=m)
            assert torch.equal(got, want), (s, m)                       # rows of <= piece_len edges: same order, same bits
            got.backward(grad); want.backward(grad)
            torch.testing.assert_close(a_x.grad, b_x.grad, rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(a_rel.grad, b_rel.grad, rtol=1e-5, atol=1e-4)

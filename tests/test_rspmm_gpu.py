"""GPU parity: libultra_rspmm.so (through the C ABI / ctypes) against the CPU oracle on the same seeded inputs.

Bars (fp32): with the oracle in the kernels' documented summation order (`piece = csr.piece_len`) every element must
be IDENTICAL; against the strictly sequential reference order (`piece=0`) rows that were not split are identical
and split rows (only the order of fp32 additions differs) agree to |diff| <= 1e-6 * S + 1e-6, where S is the
same reduction over absolute values (the sum of |terms| of that output element): both orders are within
O(sqrt(n) * 2^-24 * S) of the exact sum.
"""
import zlib

import numpy as np
import pytest
import torch

from graphs import kg_graph, random_graph

pytestmark = pytest.mark.gpu

SUMS = ["add", "min", "max"]
MULS = ["mul", "add"]
RTOL, ATOL = 1e-5, 1e-5


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _t(a):
    return torch.from_numpy(np.asarray(a)).to(_dev())


def _relcsr(g, n_dst, n_src, n_rel):
    from ultra_torchdrug_amd import RelCSR
    dev = _dev()
    t = lambda a: torch.from_numpy(a).to(dev)
    return RelCSR(t(g["dst"]), t(g["src"]), t(g["rel"]), None if g["w"] is None else t(g["w"]), n_dst, n_src, n_rel)


def _inputs(seed, n_src, n_rel, F):
    rng = np.random.default_rng(seed + 1000)
    return (rng.standard_normal((n_rel, F)).astype(np.float32), rng.standard_normal((n_src, F)).astype(np.float32))


def _same(a, b):
    """Bitwise-equal as numbers (+0 == -0; inf == inf)."""
    return np.array_equal(a, b)


CASES = {
    # name: (graph kwargs, n_node, n_rel, F)
    "small_uniform": (dict(n_edge=3000, skew=False), 200, 7, 64),
    "small_weights": (dict(n_edge=3000, skew=False, weights=True), 200, 7, 128),
    "skewed_hub": (dict(n_edge=20000, skew=True, hub_row=5, hub_edges=3000), 500, 30, 192),
    "isolated_and_ragged_F": (dict(n_edge=2000, isolated=150), 400, 5, 100),
    "narrow_F": (dict(n_edge=500, weights=True), 64, 3, 1),
    "many_relations_no_lds": (dict(n_edge=6000, skew=True), 300, 700, 64),
    "rel_graph_like": (dict(n_edge=40000, skew=False, unique=True), 120, 4, 1024),
}


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("sum", SUMS)
@pytest.mark.parametrize("mul", MULS)
def test_forward_matches_oracle(oracle, case, sum, mul):
    from ultra_torchdrug_amd import functional as UF
    kw, n, r, F = CASES[case]
    g = random_graph(seed=zlib.crc32(case.encode()) % 1000, n_node=n, n_rel=r, **kw)
    relation, x = _inputs(1, n, r, F)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
    csr = _relcsr(g, n, n, r)
    PIECE_LEN, _ = csr.kernel_order(sum, mul, F)      # 0: the plans carry their dense form (rel_graph_like: the reference order)
    want_kernel_order = oracle.rspmm_forward(csr_o, relation, x, sum, mul, piece=PIECE_LEN)
    want_sequential = oracle.rspmm_forward(csr_o, relation, x, sum, mul, piece=0)
    assert csr.n_edges == csr_o.n_edges
    dev = _dev()
    got = UF.generalized_rspmm(csr, torch.from_numpy(relation).to(dev), torch.from_numpy(x).to(dev), sum=sum, mul=mul)
    got = got.cpu().numpy()
    assert _same(got, want_kernel_order), "HIP result differs from the oracle in the kernels' summation order"
    finite = np.isfinite(want_sequential)
    assert np.array_equal(np.isfinite(got), finite)
    if sum == "add":
        w_abs = None if g["w"] is None else np.abs(g["w"])
        csr_abs = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], w_abs, n, n, r)
        if mul == "mul":
            scale = oracle.rspmm_forward(csr_abs, np.abs(relation), np.abs(x), "add", "mul")
        else:
            scale = oracle.rspmm_forward(csr_abs, np.abs(relation), np.abs(x), "add", "add")
        assert (np.abs(got - want_sequential) <= 1e-6 * scale + 1e-6).all()
    else:
        assert _same(got, want_sequential)      # min/max do not depend on the order
    # rows that were not split must be identical to the sequential reference order
    deg = np.diff(csr_o.row_ptr)
    short = (deg <= PIECE_LEN) if PIECE_LEN else np.ones(len(deg), dtype=bool)
    assert _same(got[short], want_sequential[short])


@pytest.mark.parametrize("case", ["small_uniform", "small_weights", "skewed_hub", "isolated_and_ragged_F",
                                  "many_relations_no_lds"])
@pytest.mark.parametrize("sum", SUMS)
@pytest.mark.parametrize("mul", MULS)
def test_backward_matches_oracle(oracle, case, sum, mul):
    from ultra_torchdrug_amd import functional as UF
    kw, n, r, F = CASES[case]
    kw = dict(kw, unique=True)   # min/max ties are only well defined on coalesced inputs; weights stay as given
    g = random_graph(seed=zlib.crc32(case.encode()) % 1000 + 7, n_node=n, n_rel=r, **kw)
    relation, x = _inputs(2, n, r, F)
    rng = np.random.default_rng(5)
    grad = rng.standard_normal((n, F)).astype(np.float32)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
    csr = _relcsr(g, n, n, r)
    PIECE_LEN = csr.piece_len
    out_o = oracle.rspmm_forward(csr_o, relation, x, sum, mul, piece=PIECE_LEN)
    d_rel_k, d_x_k = oracle.rspmm_backward(csr_o, relation, x, out_o, grad, sum, mul, piece=PIECE_LEN)
    d_rel_s, d_x_s = oracle.rspmm_backward(csr_o, relation, x, out_o, grad, sum, mul, piece=0)

    dev = _dev()
    rel_t = torch.from_numpy(relation).to(dev).requires_grad_()
    x_t = torch.from_numpy(x).to(dev).requires_grad_()
    out = UF.generalized_rspmm(csr, rel_t, x_t, sum=sum, mul=mul)
    assert _same(out.detach().cpu().numpy(), out_o)
    out.backward(torch.from_numpy(grad).to(dev))
    d_rel, d_x = rel_t.grad.cpu().numpy(), x_t.grad.cpu().numpy()
    assert _same(d_x, d_x_k), "d_input differs from the oracle in kernel order"
    assert _same(d_rel, d_rel_k), "d_relation differs from the oracle in kernel order"
    # sequential reference order: same terms, different fp32 addition order on split rows only
    for got_g, seq_g in ((d_x, d_x_s), (d_rel, d_rel_s)):
        bound = 2e-6 * np.abs(seq_g).max() * np.sqrt(max(csr_o.n_edges, 1)) + 1e-6
        assert np.abs(got_g - seq_g).max() <= bound


@pytest.mark.parametrize("sum", SUMS)
def test_weight_gradient(oracle, sum):
    """d(values) for a sparse tensor that requires grad (torchdrug returns it; unused by the shipped configs)."""
    from ultra_torchdrug_amd import functional as UF
    n, r, F = 150, 6, 96
    g = random_graph(seed=3, n_node=n, n_edge=2500, n_rel=r, unique=True, weights=True)
    relation, x = _inputs(3, n, r, F)
    grad = np.random.default_rng(9).standard_normal((n, F)).astype(np.float32)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
    out_o = oracle.rspmm_forward(csr_o, relation, x, sum, "mul", piece=0)
    _, _, d_w_o = oracle.rspmm_backward(csr_o, relation, x, out_o, grad, sum, "mul", need_weight_grad=True)
    dev = _dev()
    idx = torch.from_numpy(np.stack([g["dst"], g["src"], g["rel"]])).to(dev)
    val = torch.from_numpy(g["w"]).to(dev).requires_grad_()
    sparse = torch.sparse_coo_tensor(idx, val, (n, n, r))
    out = UF.generalized_rspmm(sparse, torch.from_numpy(relation).to(dev), torch.from_numpy(x).to(dev), sum=sum)
    out.backward(torch.from_numpy(grad).to(dev))
    # map oracle (coalesced order) back to input order
    key_o = (csr_o.row.astype(np.int64) * n + csr_o.col) * r + csr_o.rel
    key_in = (g["dst"] * n + g["src"]) * r + g["rel"]
    pos = np.searchsorted(key_o, key_in)
    np.testing.assert_allclose(val.grad.cpu().numpy(), d_w_o[pos], rtol=1e-4, atol=1e-4)


def test_fused_boundary_epilogue(oracle):
    """add_rows fuses `update + boundary` / `max(update, boundary)` (layer.py:156,162,358,364)."""
    from ultra_torchdrug_amd import functional as UF
    n, r, F = 300, 9, 128
    g = random_graph(seed=11, n_node=n, n_edge=9000, n_rel=r, skew=True, hub_row=2, hub_edges=1000, isolated=20)
    relation, x = _inputs(4, n, r, F)
    boundary = np.random.default_rng(2).standard_normal((n, F)).astype(np.float32)
    dev = _dev()
    csr = _relcsr(g, n, n, r)
    t = lambda a: torch.from_numpy(a).to(dev)
    for sum in SUMS:
        plain = UF.rspmm_forward(csr, t(relation), t(x), sum, "mul")
        fused = UF.rspmm_forward(csr, t(relation), t(x), sum, "mul", add_rows=t(boundary))
        b = t(boundary)
        want = plain + b if sum == "add" else (torch.maximum(plain, b) if sum == "max" else torch.minimum(plain, b))
        assert torch.equal(fused, want)


def test_wide_batch_forward_and_backward_match_oracle(oracle):
    """F = 4096 (the pre-training batch of 64 queries x 64 d, config 4): 64 column tiles, 8 per XCD."""
    from ultra_torchdrug_amd import functional as UF
    n, r, F = 700, 23, 4096
    g = random_graph(seed=21, n_node=n, n_edge=12000, n_rel=r, skew=True, hub_row=3, hub_edges=900, unique=True)
    relation, x = _inputs(9, n, r, F)
    grad = np.random.default_rng(4).standard_normal((n, F)).astype(np.float32)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
    csr = _relcsr(g, n, n, r)
    dev = _dev()
    t = lambda a: torch.from_numpy(a).to(dev)
    want = oracle.rspmm_forward(csr_o, relation, x, "add", "mul", piece=csr.piece_len)
    got = UF.rspmm_forward(csr, t(relation), t(x), "add", "mul")
    assert _same(got.cpu().numpy(), want)
    d_rel_o, d_x_o = oracle.rspmm_backward(csr_o, relation, x, want, grad, "add", "mul", piece=csr.piece_len)
    d_x, d_rel = UF.rspmm_backward(csr, t(relation), t(x), None, t(grad), "add", "mul")
    assert _same(d_x.cpu().numpy(), d_x_o) and _same(d_rel.cpu().numpy(), d_rel_o)


@pytest.mark.parametrize("flags", [0, 4, 1])
def test_sparse_boundary_equals_dense_boundary(flags):
    """ultra_rspmm_forward_boundary_f32: the Bellman-Ford boundary as (node per query, value per query) instead of the
    dense scatter_add_ result (model.py:106-107) -- identical output in every kernel (quad / packed / general),
    including split rows (node 2 is a hub: fix-up path), an isolated boundary node and a repeated node."""
    import ultra_torchdrug_amd as U
    from ultra_torchdrug_amd import functional as UF
    n, r, B, D = 300, 9, 5, 64
    g = random_graph(seed=11, n_node=n, n_edge=9000, n_rel=r, skew=True, hub_row=2, hub_edges=1000, isolated=20)
    relation, x = _inputs(4, n, r, B * D)
    dev = _dev()
    csr = _relcsr(g, n, n, r)
    t = lambda a: torch.from_numpy(a).to(dev)
    node = torch.tensor([2, 7, 299, 7, 150], dtype=torch.int64, device=dev)          # hub, twice 7, isolated 299
    value = torch.randn(B, D, generator=torch.Generator().manual_seed(3)).to(dev)
    dense = torch.zeros(n, B, D, device=dev)
    dense.scatter_add_(0, node.view(1, B, 1).expand(1, B, D), value.unsqueeze(0))
    lib = U.require_library()
    lib.ultra_rspmm_force_general_path(flags)
    try:
        for sum in SUMS:
            want = UF.rspmm_forward(csr, t(relation), t(x), sum, "mul", add_rows=dense.flatten(1))
            got = UF.rspmm_forward(csr, t(relation), t(x), sum, "mul", boundary=(node.to(torch.int32), value))
            assert torch.equal(got, want)
    finally:
        lib.ultra_rspmm_force_general_path(0)
    with pytest.raises(RuntimeError):
        UF.rspmm_forward(csr, t(relation), t(x), "add", "mul", boundary=(node, value))            # int64 nodes
    with pytest.raises(RuntimeError):
        UF.rspmm_forward(csr, t(relation), t(x), "add", "mul", boundary=(node.to(torch.int32), value[:, :32]))


def test_sparse_tensor_entry_and_errors():
    """Same call shape as the reference (layer.py:357): positional (adjacency, relation_input, input)."""
    from ultra_torchdrug_amd import functional as UF
    dev = _dev()
    n, r, F = 50, 3, 64
    g = random_graph(seed=1, n_node=n, n_edge=400, n_rel=r)
    idx = torch.from_numpy(np.stack([g["src"], g["dst"], g["rel"]])).to(dev)       # graph.adjacency: (in, out, rel)
    adjacency = torch.sparse_coo_tensor(idx, torch.ones(idx.shape[1], device=dev), (n, n, r)).transpose(0, 1)
    relation = torch.randn(r, F, device=dev)
    x = torch.randn(n, F, device=dev)
    out = UF.generalized_rspmm(adjacency, relation, x, sum="add", mul="mul")
    dense = torch.zeros(n, F, device=dev)
    msg = relation[idx[2]] * x[idx[0]]
    dense.index_add_(0, idx[1], msg)
    torch.testing.assert_close(out, dense, rtol=1e-4, atol=1e-4)
    with pytest.raises(ValueError):
        UF.generalized_rspmm(adjacency, relation, x, sum="mean")
    with pytest.raises(ValueError):
        UF.generalized_rspmm(adjacency, relation, x, mul="rotate")
    with pytest.raises(RuntimeError):
        UF.generalized_rspmm(adjacency, relation, x[:-1])
    # CPU tensors go to the CPU kernels of the same dispatcher operator (sequential order per row): every row of this
    # graph is unsplit, so the two devices agree bit for bit; mixing devices is an error
    assert torch.equal(UF.generalized_rspmm(adjacency.cpu(), relation.cpu(), x.cpu()), out.cpu())
    with pytest.raises(RuntimeError):
        UF.generalized_rspmm(adjacency, relation.cpu(), x.cpu())
    v = UF.generalized_rspmm(adjacency, relation[:, 0].contiguous(), x[:, 0].contiguous())   # 1-D input
    torch.testing.assert_close(v, dense[:, 0], rtol=1e-4, atol=1e-4)


def test_empty_graph_and_empty_rows():
    from ultra_torchdrug_amd import RelCSR, functional as UF
    dev = _dev()
    e = torch.zeros(0, dtype=torch.long, device=dev)
    csr = RelCSR(e, e, e, None, 130, 130, 4)
    relation = torch.randn(4, 64, device=dev)
    x = torch.randn(130, 64, device=dev)
    assert torch.equal(UF.generalized_rspmm(csr, relation, x, sum="add"), torch.zeros(130, 64, device=dev))
    fmax = torch.finfo(torch.float32).max       # torchdrug's NaryMin / NaryMax identities: numeric_limits max() / lowest()
    assert (UF.generalized_rspmm(csr, relation, x, sum="max") == -fmax).all()
    assert (UF.generalized_rspmm(csr, relation, x, sum="min") == fmax).all()


def test_fb15k237_shape_properties(oracle):
    """BASELINE-size graph (S-fb15k237, B=16): linearity in input and a CPU spot check on a row subset."""
    from ultra_torchdrug_amd import functional as UF
    n, base_r, triples, F = 14541, 237, 272115, 1024
    g = kg_graph(1024, n, triples, base_r)
    r = 2 * base_r
    dev = _dev()
    csr = _relcsr(g, n, n, r)
    gen = torch.Generator(device="cpu").manual_seed(1024)
    relation = torch.randn(r, F, generator=gen).to(dev)
    x1 = torch.randn(n, F, generator=gen).to(dev)
    x2 = torch.randn(n, F, generator=gen).to(dev)
    o1 = UF.generalized_rspmm(csr, relation, x1)
    o2 = UF.generalized_rspmm(csr, relation, x2)
    o12 = UF.generalized_rspmm(csr, relation, x1 + x2)
    scale = o12.abs().max().item()
    assert (o12 - (o1 + o2)).abs().max().item() <= 2e-5 * scale + 1e-3
    assert torch.equal(UF.generalized_rspmm(csr, relation, x1), o1)          # deterministic: run-to-run identical
    # oracle on the first 64 columns (one tile) of the whole graph
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], None, n, n, r)
    want = oracle.rspmm_forward(csr_o, relation[:, :64].cpu().numpy(), x1[:, :64].cpu().numpy(), piece=csr.piece_len)
    assert np.array_equal(o1[:, :64].cpu().numpy(), want)


@pytest.mark.parametrize("general", [False, True])
def test_committed_golden_vectors(general):
    """HIP library against tests/golden/rspmm_seeded.npz (bit for bit), on the packed fast path and with the
    general kernel forced (ultra_rspmm_force_general_path)."""
    import os
    import ultra_torchdrug_amd as U
    from ultra_torchdrug_amd import functional as UF
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rspmm_seeded.npz"))
    n, r, piece = int(z["n_node"]), int(z["n_rel"]), int(z["piece"])
    dev = _dev()
    t = lambda a: torch.from_numpy(a).to(dev)
    lib = U.require_library()
    lib.ultra_rspmm_force_general_path(1 if general else 0)
    try:
        csr = U.RelCSR(t(z["dst"]), t(z["src"]), t(z["rel"]), t(z["w"]), n, n, r, piece_len=piece)
        for s in SUMS:
            for m in MULS:
                rel_t, x_t = t(z["relation"]).requires_grad_(), t(z["x"]).requires_grad_()
                out = UF.generalized_rspmm(csr, rel_t, x_t, sum=s, mul=m)
                out.backward(t(z["grad"]))
                assert _same(out.detach().cpu().numpy(), z["fwd_%s_%s" % (s, m)]), (s, m)
                assert _same(x_t.grad.cpu().numpy(), z["dx_%s_%s" % (s, m)]), (s, m)
                assert _same(rel_t.grad.cpu().numpy(), z["drel_%s_%s" % (s, m)]), (s, m)
    finally:
        lib.ultra_rspmm_force_general_path(0)


@pytest.mark.parametrize("case", ["skewed_hub", "small_weights"])
def test_general_kernel_equals_packed_kernel(oracle, case):
    """All forward kernels implement one summation order: identical output."""
    import ultra_torchdrug_amd as U
    from ultra_torchdrug_amd import functional as UF
    kw, n, r, F = CASES[case]
    g = random_graph(seed=77, n_node=n, n_rel=r, **kw)
    relation, x = _inputs(6, n, r, F)
    dev = _dev()
    csr = _relcsr(g, n, n, r)
    assert csr.fwd.packed is not None
    lib = U.require_library()
    outs = []
    # bit 0: general kernel; bit 1: gathered matrix from L2 even when it fits LDS; bit 2: one chunk per wave
    # (packed_kernel) instead of four (quad_kernel).  0 = the default: quad (+ x staged in LDS when it fits)
    for general in (0, 1, 2, 4, 6):
        lib.ultra_rspmm_force_general_path(general)
        try:
            outs.append([UF.rspmm_forward(csr, torch.from_numpy(relation).to(dev), torch.from_numpy(x).to(dev), s, m)
                         for s in SUMS for m in MULS])
        finally:
            lib.ultra_rspmm_force_general_path(0)
    for first, *others in zip(*outs):
        for other in others:
            assert torch.equal(first, other)


@pytest.mark.parametrize("case", ["skewed_hub", "small_weights"])
def test_backward_kernels_agree(oracle, case):
    """d_input / d_relation of sum-aggregation: quad_kernel == packed_kernel == general kernel, bit for bit."""
    import ultra_torchdrug_amd as U
    from ultra_torchdrug_amd import functional as UF
    kw, n, r, F = CASES[case]
    g = random_graph(seed=78, n_node=n, n_rel=r, **kw)
    relation, x = _inputs(7, n, r, F)
    grad = np.random.default_rng(79).standard_normal((n, F)).astype(np.float32)
    dev = _dev()
    csr = _relcsr(g, n, n, r)
    lib = U.require_library()
    t = lambda a: torch.from_numpy(a).to(dev)
    outs = []
    for flags in (0, 4, 1):
        lib.ultra_rspmm_force_general_path(flags)
        try:
            outs.append([UF.rspmm_backward(csr, t(relation), t(x), None, t(grad), "add", m) for m in MULS])
        finally:
            lib.ultra_rspmm_force_general_path(0)
    for first, *others in zip(*outs):
        for other in others:
            for a, b in zip(first, other):
                assert (a is None) == (b is None)
                if a is not None:
                    assert torch.equal(a, b)


@pytest.mark.parametrize("case", ["skewed_hub", "small_weights", "isolated_and_ragged_F", "short_rows_wn18rr_like"])
def test_backward_accumulates_into_the_given_gradient(case):
    """``rspmm_backward(..., d_input_add=A)`` (ultra_rspmm_backward_accumulate_f32: the edge gradient lands IN ``A``, as
    the training layer node uses it) == ``A + rspmm_backward(...)`` bit for bit -- one fp32 addition per element either
    way -- with every kernel family: quad (fire-and-forget fp32 atomics in the row epilogue), packed, general, and the
    row-per-group kernel of wide-id plans.  Split hub rows (fixup_kernel adds), empty rows (A + 0) and rows ending every
    few edges included; ``d_relation`` must not change."""
    import ultra_torchdrug_amd as U
    from ultra_torchdrug_amd import RelCSR, functional as UF
    if case == "short_rows_wn18rr_like":
        n, r, F = 5000, 22, 256
        g = random_graph(seed=80, n_node=n, n_rel=r, n_edge=21000, skew=False)
    else:
        kw, n, r, F = CASES[case]
        g = random_graph(seed=78, n_node=n, n_rel=r, **kw)
    relation, x = _inputs(7, n, r, F)
    rng = np.random.default_rng(81)
    grad = rng.standard_normal((n, F)).astype(np.float32)
    base = rng.standard_normal((n, F)).astype(np.float32)
    base[::7] = 0.0                                             # rows of exact zeros (and -0.0) in the accumulator
    base[3::11] = -0.0
    dev = _dev()
    t = lambda a: torch.from_numpy(a).to(dev)
    lib = U.require_library()
    plans = [_relcsr(g, n, n, r)]
    if F % 4 == 0:
        plans.append(RelCSR(t(g["dst"]), t(g["src"]), t(g["rel"]), None if g["w"] is None else t(g["w"]), n, n, r,
                            wide_ids=True))
    for csr in plans:
        for m in MULS:
            for flags in (0, 4, 1):
                lib.ultra_rspmm_force_general_path(flags)
                try:
                    plain_x, plain_r = UF.rspmm_backward(csr, t(relation), t(x), None, t(grad), "add", m)
                    acc = t(base).clone()
                    got_x, got_r = UF.rspmm_backward(csr, t(relation), t(x), None, t(grad), "add", m, d_input_add=acc)
                finally:
                    lib.ultra_rspmm_force_general_path(0)
                assert got_x.data_ptr() == acc.data_ptr()                  # in place
                want = t(base) + plain_x
                assert torch.equal(got_x, want), (m, flags)
                assert got_x.view(torch.int32).eq(want.view(torch.int32)).all() or \
                    bool(((got_x == 0) & (want == 0))[got_x.view(torch.int32) != want.view(torch.int32)].all())
                assert torch.equal(got_r, plain_r), (m, flags)


@pytest.mark.parametrize("rows", [1, 31, 32, 33, 4096 + 17, 14541 * 16])
@pytest.mark.parametrize("ln,relu,shortcut", [(True, True, True), (False, True, False), (True, False, True)])
def test_fused_combine_matches_oracle_and_torch(oracle, rows, ln, relu, shortcut):
    """ultra_combine_forward_f32 vs (a) the oracle in the kernel's documented order and (b) the reference's own
    formulation in torch: relu(LayerNorm(Linear(cat[input, update]))) + input (layer.py:386-392, model.py:126-127)."""
    from ultra_torchdrug_amd import functional as UF
    dev = _dev()
    gen = torch.Generator(device="cpu").manual_seed(rows)
    x = torch.randn(rows, 64, generator=gen)
    u = torch.randn(rows, 64, generator=gen) * 3
    lin = torch.nn.Linear(128, 64)
    norm = torch.nn.LayerNorm(64)
    with torch.no_grad():
        norm.weight.copy_(torch.rand(64, generator=gen) + 0.5)
        norm.bias.copy_(torch.randn(64, generator=gen) * 0.1)
        got = UF.combine_forward(x.to(dev), u.to(dev), lin.weight.to(dev), lin.bias.to(dev),
                                 norm.weight.to(dev) if ln else None, norm.bias.to(dev) if ln else None, norm.eps,
                                 relu, shortcut).cpu()
        ref = lin(torch.cat([x, u], dim=-1))
        if ln:
            ref = norm(ref)
        if relu:
            ref = torch.relu(ref)
        if shortcut:
            ref = ref + x
    torch.testing.assert_close(got, ref, rtol=2e-5, atol=2e-5)
    sub = slice(0, min(rows, 5000))
    want = oracle.combine_forward(x[sub].numpy(), u[sub].numpy(), lin.weight.detach().numpy(), lin.bias.detach().numpy(),
                                  norm.weight.detach().numpy() if ln else None, norm.bias.detach().numpy() if ln else None,
                                  norm.eps, relu, shortcut)
    assert np.array_equal(got[sub].numpy(), want), "fused combine differs from the oracle's documented order"


@pytest.mark.parametrize("case", ["skewed_hub", "small_weights", "many_relations_no_lds", "isolated_and_ragged_F"])
def test_wide_id_variants_match_oracle(oracle, case):
    """Big-graph kernel variants (node ids outside the packed word; relation tile in LDS or, when it does not fit,
    read through L2), forced on small graphs with `wide_ids=True`: forward and sum-backward, bit for bit.
    (One chunk per wave: a four-chunks-per-wave form of these variants measured 5 % slower on the DRAM-bound
    S-stress graph, whose rows hold 10 edges.)"""
    from ultra_torchdrug_amd import RelCSR, functional as UF
    kw, n, r, F = CASES[case]
    g = random_graph(seed=zlib.crc32(case.encode()) % 1000 + 3, n_node=n, n_rel=r, **dict(kw, unique=True))
    relation, x = _inputs(8, n, r, F)
    grad = np.random.default_rng(3).standard_normal((n, F)).astype(np.float32)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
    dev = _dev()
    t = lambda a: torch.from_numpy(a).to(dev)
    csr = RelCSR(t(g["dst"]), t(g["src"]), t(g["rel"]), None if g["w"] is None else t(g["w"]), n, n, r, wide_ids=True)
    assert csr.fwd.packed_src_shift == 32 and csr.by_src.packed_src_shift == 32
    PIECE_LEN = csr.piece_len
    for s in SUMS:
        for m in MULS:
            want = oracle.rspmm_forward(csr_o, relation, x, s, m, piece=PIECE_LEN)
            got = UF.rspmm_forward(csr, t(relation), t(x), s, m)
            assert _same(got.cpu().numpy(), want), (s, m)
    for m in MULS:
        out = oracle.rspmm_forward(csr_o, relation, x, "add", m, piece=PIECE_LEN)
        d_rel_o, d_x_o = oracle.rspmm_backward(csr_o, relation, x, out, grad, "add", m, piece=PIECE_LEN)
        d_x, d_rel = UF.rspmm_backward(csr, t(relation), t(x), None, t(grad), "add", m)
        assert _same(d_x.cpu().numpy(), d_x_o) and _same(d_rel.cpu().numpy(), d_rel_o), m


@pytest.mark.parametrize("rows", [1, 33, 1000, 65536 + 7])
@pytest.mark.parametrize("ln,relu,shortcut", [(True, True, True), (False, True, False), (True, False, False)])
def test_fused_combine_backward_matches_autograd_of_the_reference_formulation(rows, ln, relu, shortcut):
    """functional.combine (fused forward + fused backward) against torch autograd on the reference's own chain
    cat -> Linear -> LayerNorm -> relu (+ input) (layer.py:386-392, model.py:126-127), evaluated in fp64 on the host.
    fp32 tolerance: 2e-5 relative to the largest entry of each gradient (reductions over up to 65k rows).
    ReLU is not differentiable at 0: rows holding a pre-activation within 1e-4 of 0 (where the fp32 and the fp64
    chain may pick different sides) get a zero output gradient on both sides, so they do not enter any gradient."""
    from ultra_torchdrug_amd import functional as UF
    dev = _dev()
    torch.manual_seed(rows + 17)        # nn.Linear's initialisation draws from the global generator
    gen = torch.Generator(device="cpu").manual_seed(rows + 17)
    x = torch.randn(rows, 64, generator=gen)
    u = torch.randn(rows, 64, generator=gen) * 2
    gout = torch.randn(rows, 64, generator=gen)
    lin = torch.nn.Linear(128, 64)
    norm = torch.nn.LayerNorm(64)
    with torch.no_grad():
        norm.weight.copy_(torch.rand(64, generator=gen) + 0.5)
        norm.bias.copy_(torch.randn(64, generator=gen) * 0.1)

    # reference chain in fp64
    lin64, norm64 = torch.nn.Linear(128, 64).double(), torch.nn.LayerNorm(64).double()
    lin64.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
    norm64.load_state_dict({k: v.double() for k, v in norm.state_dict().items()})
    x64, u64 = x.double().requires_grad_(), u.double().requires_grad_()
    ref = lin64(torch.cat([x64, u64], dim=-1))
    if ln:
        ref = norm64(ref)
    if relu:
        gout[(ref.detach().abs() < 1e-4).any(dim=1)] = 0.0
        ref = torch.relu(ref)
    if shortcut:
        ref = ref + x64
    ref.backward(gout.double())

    params = [lin.weight, lin.bias] + ([norm.weight, norm.bias] if ln else [])
    gpu = [t.detach().to(dev).requires_grad_() for t in [x, u] + params]
    out = UF.combine(gpu[0], gpu[1], gpu[2], gpu[3], gpu[4] if ln else None, gpu[5] if ln else None, norm.eps, relu, shortcut)
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=2e-5, atol=2e-5)
    out.backward(gout.to(dev))
    want = [x64.grad, u64.grad, lin64.weight.grad, lin64.bias.grad] + ([norm64.weight.grad, norm64.bias.grad] if ln else [])
    names = ["d_input", "d_update", "d_weight", "d_bias", "d_ln_weight", "d_ln_bias"]
    for name, g, w in zip(names, gpu, want):
        scale = w.abs().max().item() + 1e-12
        err = (g.grad.cpu().double() - w).abs().max().item()
        assert err <= 2e-5 * scale + 1e-6, "%s: err %.3g vs scale %.3g" % (name, err, scale)


@pytest.mark.parametrize("shape", [(64, 64, True), (64, 64, False), (128, 128, True), (128, 1, False)])
@pytest.mark.parametrize("rows", [1, 37, 7584, 232656])
def test_documented_order_linear_matches_oracle_and_torch(oracle, shape, rows):
    """ultra_linear_forward_f32 (relation projection 64->64->64, score head 128->128->1): identical bits to the
    oracle's fmaf chain, and within fp32 tolerance of nn.Linear (the reference's formulation)."""
    from ultra_torchdrug_amd import functional as UF
    k, n, relu = shape
    dev = _dev()
    gen = torch.Generator(device="cpu").manual_seed(rows + k)
    x = torch.randn(rows, k, generator=gen)
    lin = torch.nn.Linear(k, n)
    with torch.no_grad():
        got = UF.linear_forward(x.to(dev), lin.weight.to(dev), lin.bias.to(dev), relu=relu).cpu()
        ref = lin(x)
        if relu:
            ref = torch.relu(ref)
    torch.testing.assert_close(got, ref, rtol=2e-5, atol=2e-5)
    sub = slice(0, min(rows, 4000))
    want = oracle.linear_forward(x[sub].numpy(), lin.weight.detach().numpy(), lin.bias.detach().numpy(), relu)
    assert np.array_equal(got[sub].numpy(), want)


def test_hot_row_cache_variant_matches_plain_variant(oracle):
    """Plans with and without the LDS hot-row cache (the n_hot most gathered nodes) give identical forward and
    sum-backward results, equal to the oracle."""
    from ultra_torchdrug_amd import RelCSR, functional as UF
    n, r, F = 3000, 40, 128
    g = random_graph(seed=31, n_node=n, n_edge=120000, n_rel=r, skew=True, unique=True, weights=True)
    relation, x = _inputs(9, n, r, F)
    grad = np.random.default_rng(4).standard_normal((n, F)).astype(np.float32)
    dev = _dev()
    t = lambda a: torch.from_numpy(a).to(dev)
    hot = RelCSR(t(g["dst"]), t(g["src"]), t(g["rel"]), t(g["w"]), n, n, r, hot_cache=True)
    cold = RelCSR(t(g["dst"]), t(g["src"]), t(g["rel"]), t(g["w"]), n, n, r)
    assert hot.fwd.n_hot >= 16 and hot.by_src.n_hot >= 16 and cold.fwd.n_hot == 0
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
    for s in SUMS:
        for m in MULS:
            a = UF.rspmm_forward(hot, t(relation), t(x), s, m)
            b = UF.rspmm_forward(cold, t(relation), t(x), s, m)
            assert torch.equal(a, b), (s, m)
            assert _same(a.cpu().numpy(), oracle.rspmm_forward(csr_o, relation, x, s, m, piece=hot.piece_len))
    for m in MULS:
        da, ra = UF.rspmm_backward(hot, t(relation), t(x), None, t(grad), "add", m)
        db, rb = UF.rspmm_backward(cold, t(relation), t(x), None, t(grad), "add", m)
        assert torch.equal(da, db) and torch.equal(ra, rb)


@pytest.mark.parametrize("case", ["kg_unit_weights", "kg_zero_weight_edges", "skewed_weights_hub_split"])
@pytest.mark.parametrize("mul", ["mul", "add"])
def test_backward_with_activity_masks_equals_the_full_backward(case, mul):
    """Round 4: ``ultra_rspmm_backward_active_f32`` -- the backward of sum-aggregation told WHICH gradient rows (last layer of a
    training step: the candidates' rows, one bitmap per 64-column query block) or WHICH input row per block (first layer: the
    boundary node) can be non-zero.  Edges that can only add zero issue their gathers past the buffer descriptors (0.0, no
    memory traffic); d_input (alone and accumulated into a given gradient) and d_relation must EQUAL the unmasked backward bit
    for bit on operands that keep the promise."""
    from ultra_torchdrug_amd import RelCSR, functional as UF
    dev = _dev()
    if case == "skewed_weights_hub_split":
        n, r = 900, 14
        g = random_graph(seed=8, n_node=n, n_edge=30000, n_rel=r, skew=True, weights=True, hub_row=5, hub_edges=4000)
        opts = dict(chunk_edges=16, piece_len=64)
    else:
        n, r = 14541, 474
        g = kg_graph(1024, n, 272115, 237)
        opts = {}
        if case == "kg_zero_weight_edges":          # the training step's edge removal: per-edge weights, a few of them 0
            rng_w = np.random.default_rng(3)
            g["w"] = (rng_w.random(len(g["dst"])) > 0.01).astype(np.float32)
    csr = RelCSR(_t(g["dst"]), _t(g["src"]), _t(g["rel"]), None if g["w"] is None else _t(g["w"]), n, n, r, **opts)
    B, K = 16, 129
    F = B * 64
    gen = torch.Generator(device=dev).manual_seed(zlib.crc32((case + mul).encode()) % 1000)
    relation = torch.randn(r, F, device=dev, generator=gen)
    x = torch.randn(n, F, device=dev, generator=gen)
    already = torch.randn(n, F, device=dev, generator=gen)

    # ---- last layer: the gradient is non-zero at the candidates' (node, query) rows only
    t_index = torch.randint(0, n, (B, K), device=dev, generator=gen)
    t_index[:, 0] = torch.bincount(csr.dst, minlength=n).argmax()                      # a hub among the candidates
    bits = UF.candidate_rows(t_index, n)
    assert bits is not None and bits.shape == (B, (n + 31) // 32) and bits.dtype == torch.int32
    member = torch.zeros(B, n, dtype=torch.bool, device=dev)
    member[torch.arange(B, device=dev).unsqueeze(-1), t_index] = True
    words = bits.view(B, -1, 1).to(torch.int64) & 0xffffffff
    unpacked = ((words >> torch.arange(32, device=dev)) & 1).bool().flatten(1)[:, :n]
    assert torch.equal(unpacked, member), "node bitmap differs from the candidate sets"
    grad = torch.randn(n, B, 64, device=dev, generator=gen) * member.t().unsqueeze(-1)   # zero rows elsewhere
    grad = grad.flatten(1).contiguous()
    want_dx, want_drel = UF.rspmm_backward(csr, relation, x, None, grad, "add", mul)
    got_dx, got_drel = UF.rspmm_backward(csr, relation, x, None, grad, "add", mul, active_dst=bits)
    assert torch.equal(got_dx, want_dx) and torch.equal(got_drel, want_drel)
    want_acc, _ = UF.rspmm_backward(csr, relation, x, None, grad, "add", mul, need_relation=False, d_input_add=already.clone())
    got_acc, _ = UF.rspmm_backward(csr, relation, x, None, grad, "add", mul, need_relation=False, d_input_add=already.clone(),
                                   active_dst=bits)
    assert torch.equal(got_acc, want_acc)

    # ---- first layer: the input is the boundary -- zero outside row node[b] of block b (d_relation, mul = mul only)
    if mul == "mul":
        deg_out = torch.bincount(csr.src, minlength=n)
        node = torch.randint(0, n, (B,), device=dev, generator=gen).to(torch.int32)
        node[0], node[1] = int(deg_out.argmax()), int(deg_out.argmin())                # a hub head and a (nearly) isolated one
        boundary = torch.zeros(n, B, 64, device=dev)
        boundary[node.long(), torch.arange(B, device=dev)] = torch.randn(B, 64, device=dev, generator=gen)
        boundary = boundary.flatten(1).contiguous()
        dense_grad = torch.randn(n, F, device=dev, generator=gen)
        _, want = UF.rspmm_backward(csr, relation, boundary, None, dense_grad, "add", "mul", need_input=False)
        _, got = UF.rspmm_backward(csr, relation, boundary, None, dense_grad, "add", "mul", need_input=False, active_src=node)
        assert torch.equal(got, want)
        # ... and from those nodes' out-edges alone (ultra_rspmm_drelation_boundary_f32): no walk over the other edges at all
        items, src_ptr, src_relpos = csr.boundary_relation_index
        assert items.shape[0] == csr.by_rel.n_pieces + r - csr.by_rel.long_rows.shape[0]
        assert torch.equal(torch.sort(src_relpos.long()).values, torch.arange(csr.n_edges, device=dev))
        got = UF.rspmm_drelation_boundary(csr, boundary, dense_grad, node)
        assert torch.equal(got, want)
        same = node.clone()
        same[:] = node[0]                                                              # every query at the hub
        boundary = torch.zeros(n, B, 64, device=dev)
        boundary[same.long(), torch.arange(B, device=dev)] = torch.randn(B, 64, device=dev, generator=gen)
        boundary = boundary.flatten(1).contiguous()
        _, want = UF.rspmm_backward(csr, relation, boundary, None, dense_grad, "add", "mul", need_input=False)
        assert torch.equal(UF.rspmm_drelation_boundary(csr, boundary, dense_grad, same), want)
        with pytest.raises(RuntimeError):
            UF.rspmm_drelation_boundary(csr, boundary, dense_grad, same[:-1].contiguous())


@pytest.mark.parametrize("mul", ["mul", "add"])
def test_removed_edges_as_marked_words_equal_zero_weights(mul):
    """The training step's edge removal on a unit-weight graph (``RelCSR.with_removed_edges``) hands the sum / mul kernels copies of the
    packed words with bit 31 set at the removed edges (``ultra_edge_removal_marks``, quad.inc DEAD) beside the zero weights: forward,
    d_input, d_relation -- full and with the last layer's destination masks -- must EQUAL what the weighted kernels give (knob bit 7)
    and what a plainly reweighted graph gives; the TransE message keeps the weighted kernels."""
    from ultra_torchdrug_amd import RelCSR, _lib, functional as UF
    dev = _dev()
    n, r = 14541, 474
    g = kg_graph(1024, n, 272115, 237)
    distinct = np.unique(np.stack([g["dst"], g["src"], g["rel"]]), axis=1)      # a KG lists a triple once: every weight is 1
    csr = RelCSR(_t(distinct[0]), _t(distinct[1]), _t(distinct[2]), None, n, n, r)
    assert csr.unit_weight
    gen = torch.Generator(device=dev).manual_seed(11 + (mul == "add"))
    # triples to remove: 64 real edges (hub rows among them), listed with repeats, + 200 patterns that are no edges
    deg = torch.bincount(csr.dst, minlength=n)
    hub_edges = torch.nonzero(csr.dst == int(deg.argmax())).flatten()[:8]
    pick = torch.cat([torch.randint(0, csr.n_edges, (56,), device=dev, generator=gen), hub_edges])
    pick = pick[csr.rel_id[pick] < r // 2]                                  # base relations: the call removes the inverse itself
    h = torch.cat([csr.src[pick], csr.src[pick][:5], torch.randint(0, n, (200,), device=dev, generator=gen)])
    t = torch.cat([csr.dst[pick], csr.dst[pick][:5], torch.randint(0, n, (200,), device=dev, generator=gen)])
    rel = torch.cat([csr.rel_id[pick], csr.rel_id[pick][:5], torch.randint(0, r // 2, (200,), device=dev, generator=gen)])
    cut = csr.with_removed_edges(h, t, rel, r // 2)
    for plan, base in ((cut.fwd, csr.fwd), (cut.by_src, csr.by_src), (cut.by_rel, csr.by_rel)):
        assert plan.packed_dead is not None and plan.struct.packed_dead == plan.packed_dead.data_ptr()
        E = csr.n_edges
        marked = (plan.packed_dead[:E].long() & 0x80000000) != 0
        assert torch.equal(marked, plan.weight[:E] == 0) and int(marked.sum()) >= torch.unique(pick).numel()
        assert torch.equal(plan.packed_dead.long() & 0x7fffffff, base.packed.long() & 0xffffffff)
        assert torch.equal(plan.packed, base.packed)
    B = 16
    F = B * 64
    relation = torch.randn(r, F, device=dev, generator=gen)
    x = torch.randn(n, F, device=dev, generator=gen)
    grad = torch.randn(n, F, device=dev, generator=gen)
    t_index = torch.randint(0, n, (B, 129), device=dev, generator=gen)
    bits = UF.candidate_rows(t_index, n)
    member = torch.zeros(B, n, dtype=torch.bool, device=dev)
    member[torch.arange(B, device=dev).unsqueeze(-1), t_index] = True
    sparse_grad = (torch.randn(n, B, 64, device=dev, generator=gen) * member.t().unsqueeze(-1)).flatten(1).contiguous()
    lib = _lib.load()

    def run():
        out = UF.rspmm_forward(cut, relation, x, "add", mul)
        dx, drel = UF.rspmm_backward(cut, relation, x, None, grad, "add", mul)
        dx2, drel2 = UF.rspmm_backward(cut, relation, x, None, sparse_grad, "add", mul, active_dst=bits)
        return out, dx, drel, dx2, drel2

    got = run()
    lib.ultra_rspmm_force_general_path(128)              # bit 7: the weighted kernels
    try:
        want = run()
    finally:
        lib.ultra_rspmm_force_general_path(0)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert not torch.equal(got[0], UF.rspmm_forward(csr, relation, x, "add", mul))       # (the removal does change the result)


@pytest.mark.parametrize("shape", ["small", "hub_split", "wide_ids", "no_free_bit"])
def test_removed_edges_on_other_plan_shapes_equal_plain_reweighting(shape):
    """``with_removed_edges`` against a RelCSR that simply carries the zero weights (no marks: the weighted kernels), on plan shapes the
    KG-sized test does not reach: a graph of a few chunks, hub rows cut into pieces, node ids outside the packed word and a node range
    that leaves no free bit (both: no marks, weights only), every edge of one row removed."""
    from ultra_torchdrug_amd import RelCSR, functional as UF
    dev = _dev()
    opts = {}
    if shape == "small":
        n, r, g = 300, 6, random_graph(seed=3, n_node=300, n_edge=2500, n_rel=6)
    elif shape == "hub_split":
        n, r, g = 900, 14, random_graph(seed=8, n_node=900, n_edge=30000, n_rel=14, skew=True, hub_row=5, hub_edges=4000)
        opts = dict(chunk_edges=16, piece_len=64)
    elif shape == "wide_ids":
        n, r, g = 700, 10, random_graph(seed=5, n_node=700, n_edge=9000, n_rel=10)
        opts = dict(wide_ids=True)
    else:
        n, r, g = 20000, 474, random_graph(seed=6, n_node=20000, n_edge=40000, n_rel=474)     # 8 + 9 + 15 bits: the word is full
    distinct = np.unique(np.stack([g["dst"], g["src"], g["rel"]]), axis=1)
    keep = distinct[2] < r // 2                                              # base relations; the inverses are added below
    d, s_, rel_ = distinct[0][keep], distinct[1][keep], distinct[2][keep]
    dst = np.concatenate([d, s_]); src = np.concatenate([s_, d]); rel = np.concatenate([rel_, rel_ + r // 2])
    both = np.unique(np.stack([dst, src, rel]), axis=1)
    csr = RelCSR(_t(both[0]), _t(both[1]), _t(both[2]), None, n, n, r, **opts)
    assert csr.unit_weight
    gen = torch.Generator(device=dev).manual_seed(zlib.crc32(shape.encode()) % 1000)
    row = int(torch.bincount(csr.dst, minlength=n).argmax())
    whole_row = torch.nonzero((csr.dst == row) & (csr.rel_id < r // 2)).flatten()          # every base edge into the heaviest row
    pick = torch.cat([whole_row, torch.randint(0, csr.n_edges, (40,), device=dev, generator=gen)])
    pick = pick[csr.rel_id[pick] < r // 2]
    h, t, rr = csr.src[pick], csr.dst[pick], csr.rel_id[pick]
    cut = csr.with_removed_edges(h, t, rr, r // 2)
    marked = cut.fwd.packed_dead is not None
    assert marked == (shape in ("small", "hub_split"))
    # the same zero weights on a RelCSR of its own (plans rebuilt with weights: no marks anywhere)
    w = cut.weight.clone()
    ref = RelCSR(csr.dst, csr.src, csr.rel_id, w, n, n, r, **opts)
    assert ref.fwd.weight is not None and getattr(ref.fwd, "packed_dead", None) is None
    F = 256
    relation = torch.randn(r, F, device=dev, generator=gen)
    x = torch.randn(n, F, device=dev, generator=gen)
    grad = torch.randn(n, F, device=dev, generator=gen)
    for mul in ("mul", "add"):
        out_a = UF.rspmm_forward(cut, relation, x, "add", mul)
        out_b = UF.rspmm_forward(ref, relation, x, "add", mul)
        assert torch.equal(out_a, out_b), (shape, mul)
        da, ra = UF.rspmm_backward(cut, relation, x, None, grad, "add", mul)
        db, rb = UF.rspmm_backward(ref, relation, x, None, grad, "add", mul)
        assert torch.equal(da, db) and torch.equal(ra, rb), (shape, mul)

"""GPU: the native plan builder (`ultra_relcsr_coalesce` / `ultra_relcsr_plan`, csrc/relcsr_build.hip, rocPRIM) must emit
exactly the arrays of the torch-op construction in relcsr.py (the executable specification that the CPU tests cover):
coalesced triples, merged weights, edge_of_input, chunk schedule, long-row table, packed words.  Integer work: bit-exact.
"""
import numpy as np
import pytest
import torch

from graphs import kg_graph, random_graph

pytestmark = pytest.mark.gpu

CASES = {
    # name: (graph kwargs, n_node, n_rel, RelCSR options)
    "small_dups": (dict(seed=1, n_node=40, n_edge=600, n_rel=3), 40, 3, {}),
    "weights_dups": (dict(seed=2, n_node=90, n_edge=3000, n_rel=5, weights=True), 90, 5, {}),
    "skew_hubs": (dict(seed=3, n_node=500, n_edge=20000, n_rel=11, skew=True, weights=True), 500, 11, {}),
    "hub_row_split": (dict(seed=4, n_node=300, n_edge=9000, n_rel=7, unique=True, hub_row=5, hub_edges=4000), 300, 7,
                      dict(chunk_edges=32, piece_len=128)),
    "isolated_rows": (dict(seed=5, n_node=400, n_edge=1500, n_rel=4, isolated=150), 400, 4, {}),
    "wide_ids": (dict(seed=6, n_node=200, n_edge=5000, n_rel=9, skew=True), 200, 9, dict(wide_ids=True)),
    "no_balance": (dict(seed=7, n_node=200, n_edge=5000, n_rel=9, skew=True), 200, 9, dict(balance=False)),
    "one_relation": (dict(seed=8, n_node=64, n_edge=800, n_rel=1), 64, 1, {}),
    "single_edge": (dict(seed=9, n_node=3, n_edge=1, n_rel=2), 3, 2, {}),
    "big_chunks": (dict(seed=10, n_node=3000, n_edge=160000, n_rel=40, skew=True), 3000, 40, {}),
}


def _build(g, n_node, n_rel, builder, **opts):
    from ultra_torchdrug_amd import RelCSR
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    return RelCSR(t(g["dst"]), t(g["src"]), t(g["rel"]), None if g["w"] is None else t(g["w"]), n_node, n_node, n_rel,
                  builder=builder, **opts)


def _same(a, b, what):
    assert (a is None) == (b is None), what
    if a is not None:
        assert a.dtype == b.dtype and a.shape == b.shape, "%s: %s %s vs %s %s" % (what, a.dtype, a.shape, b.dtype, b.shape)
        assert torch.equal(a, b), what


def _compare(native, ref):
    assert native.n_edges == ref.n_edges and native.unit_weight == ref.unit_weight
    assert (native.chunk_edges, native.piece_len) == (ref.chunk_edges, ref.piece_len)
    for name in ("dst", "src", "rel_id", "weight", "edge_of_input"):
        _same(getattr(native, name), getattr(ref, name), name)
    for plan in ("fwd", "by_src", "by_rel"):
        a, b = getattr(native, plan), getattr(ref, plan)
        assert a.builder == "native" and b.builder == "torch"
        assert (a.n_rows, a.n_edges, a.n_pieces, a.packed_src_shift) == (b.n_rows, b.n_edges, b.n_pieces, b.packed_src_shift), plan
        for name in ("row", "node_a", "node_b", "rel", "weight", "chunks", "long_rows", "packed"):
            _same(getattr(a, name), getattr(b, name), "%s.%s" % (plan, name))


@pytest.mark.parametrize("case", sorted(CASES))
def test_native_builder_equals_torch_builder(case):
    kwargs, n_node, n_rel, opts = CASES[case]
    g = random_graph(**kwargs)
    _compare(_build(g, n_node, n_rel, "native", **opts), _build(g, n_node, n_rel, "torch", **opts))


def test_native_builder_kg_shape():
    g = kg_graph(11, 14541, 60000, 237)
    _compare(_build(g, 14541, 474, "native"), _build(g, 14541, 474, "torch"))


def test_native_builder_empty_graph():
    from ultra_torchdrug_amd import RelCSR
    dev = torch.device("cuda:0")
    e = torch.zeros(0, dtype=torch.long, device=dev)
    a = RelCSR(e, e, e, None, 10, 10, 2, builder="native")
    b = RelCSR(e, e, e, None, 10, 10, 2, builder="torch")
    _compare(a, b)


def test_native_builder_is_the_default_on_device():
    g = random_graph(seed=12, n_node=30, n_edge=200, n_rel=2)
    csr = _build(g, 30, 2, None)
    assert csr.fwd.builder == "native" and csr.by_src.builder == "native" and csr.by_rel.builder == "native"


def test_native_builder_rejects_short_workspace():
    from ultra_torchdrug_amd import _lib
    import ctypes
    lib = _lib.load()
    dev = torch.device("cuda:0")
    row = torch.zeros(8, dtype=torch.int32, device=dev)
    chunks = torch.empty(16, 4, dtype=torch.int32, device=dev)
    long_rows = torch.empty(4, 3, dtype=torch.int32, device=dev)
    temp = torch.empty(64, dtype=torch.uint8, device=dev)
    counts = (ctypes.c_int64 * 4)()
    status = lib.ultra_relcsr_plan(row.data_ptr(), row.data_ptr(), row.data_ptr(), 8, 4, 4, 2, 0, 0, 1, 32, 32, 128,
                                   chunks.data_ptr(), 16, long_rows.data_ptr(), 4, None, 16, counts, temp.data_ptr(), 64,
                                   None)
    assert status != 0 and b"workspace" in lib.ultra_rspmm_status_string(status).lower()


def test_native_relation_graph_equals_the_reference_products():
    """construct_relation_graph (ultra/rel_model.py:91-143) on the device -- ultra_relation_graph_marks over the entities'
    distinct relation lists -- gives the edge list of the four sparse incidence products (the ATen path, run here on a CPU copy
    of the graph): same edges, same order."""
    from ultra_torchdrug_amd.data import synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.rel_model import construct_relation_graph
    dev = torch.device("cuda:0")
    shapes = [((300, 2000, 6), 1.0), ((2034, 32888, 42), 1.0), ((14541, 272115, 237), 1.0), ((5000, 3000, 300), 0.0),
              ((50, 40, 70), 0.0)]
    for shape, alpha in shapes:
        tr, n, r = synthetic_triples(shape, 5, alpha=alpha)
        cpu = Graph(torch.from_numpy(tr), num_node=n + 3, num_relation=r + 2)          # isolated entities, unused relations
        want = construct_relation_graph(cpu)
        got = construct_relation_graph(cpu.to(dev))
        assert got.num_node == want.num_node == 2 * (r + 2) and got.num_relation == 4
        assert torch.equal(got.edge_list.cpu(), want.edge_list), shape
    empty = Graph(torch.zeros(0, 3, dtype=torch.long), num_node=5, num_relation=3)
    assert construct_relation_graph(empty.to(dev)).num_edge == 0


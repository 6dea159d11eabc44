"""The product's CPU path (SURVEY.md 8b "Native exports ... each with CPU and HIP kernels"; BASELINE config 1 runs
``--gpus null``, /root/reference/README.md:79,90): ``torch.ops.ultra_mi.{build_relcsr, rspmm_fwd, rspmm_bwd}`` on the
CPU dispatch key (``csrc/torch_ext.cpp``), reached through ``generalized_rspmm`` with CPU tensors.

Bars: bit-equal to the oracle's SEQUENTIAL order (``piece = 0``: the reference's order) for all six operator pairs,
forward and both gradients, with and without edge weights, ragged widths and empty rows included; ``build_relcsr``
equal to the oracle's coalesce; and config 1 end to end -- inductive zero-shot ``predict`` of the shipped architecture
on CPU tensors with no backend installed, against the same model with the oracle behind the operator.
"""
import os
import zlib

import numpy as np
import pytest
import torch

from graphs import kg_graph, random_graph
from oracle_ops import oracle_rspmm

CASES = {
    "uniform": (dict(n_edge=3000), 200, 7, 64),
    "weights_duplicates": (dict(n_edge=4000, weights=True, skew=True), 150, 5, 96),
    "hub_isolated_ragged": (dict(n_edge=6000, skew=True, hub_row=5, hub_edges=2500, isolated=60), 300, 11, 100),
    "narrow": (dict(n_edge=400, weights=True), 64, 3, 1),
    "wide_slabs": (dict(n_edge=1500), 90, 4, 600),
}


def _csr(g, n, r):
    from ultra_torchdrug_amd import RelCSR
    t = torch.from_numpy
    return RelCSR(t(g["dst"]), t(g["src"]), t(g["rel"]), None if g["w"] is None else t(g["w"]), n, n, r)


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("sum", ["add", "min", "max"])
@pytest.mark.parametrize("mul", ["mul", "add"])
def test_cpu_operator_forward_backward_equal_the_oracle_sequential_order(oracle, case, sum, mul):
    from ultra_torchdrug_amd import generalized_rspmm
    kw, n, r, F = CASES[case]
    g = random_graph(seed=zlib.crc32(case.encode()) % 1000, n_node=n, n_rel=r, **kw)
    rng = np.random.default_rng(5)
    relation = rng.standard_normal((r, F)).astype(np.float32)
    x = rng.standard_normal((n, F)).astype(np.float32)
    grad = rng.standard_normal((n, F)).astype(np.float32)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
    want = oracle.rspmm_forward(csr_o, relation, x, sum, mul, piece=0)
    want_d_rel, want_d_x = oracle.rspmm_backward(csr_o, relation, x, want, grad, sum, mul, piece=0)
    rel_t, x_t = torch.from_numpy(relation).requires_grad_(), torch.from_numpy(x).requires_grad_()
    out = generalized_rspmm(_csr(g, n, r), rel_t, x_t, sum=sum, mul=mul)
    assert np.array_equal(out.detach().numpy(), want)
    out.backward(torch.from_numpy(grad))
    assert np.array_equal(x_t.grad.numpy(), want_d_x)
    assert np.array_equal(rel_t.grad.numpy(), want_d_rel)


def test_cpu_operator_takes_the_reference_sparse_tensor_and_1d_input(oracle):
    """The reference's call form (ultra/layer.py:127,134: ``adjacency.transpose(0, 1)``, un-coalesced COO) on CPU."""
    from ultra_torchdrug_amd import generalized_rspmm
    n, r, F = 120, 6, 64
    g = random_graph(seed=9, n_node=n, n_edge=2500, n_rel=r, weights=True)
    sparse = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g["dst"], g["src"], g["rel"]])), torch.from_numpy(g["w"]), (n, n, r))
    rng = np.random.default_rng(2)
    relation, x = rng.standard_normal((r, F)).astype(np.float32), rng.standard_normal((n, F)).astype(np.float32)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
    got = generalized_rspmm(sparse, torch.from_numpy(relation), torch.from_numpy(x), sum="max", mul="add")
    assert np.array_equal(got.numpy(), oracle.rspmm_forward(csr_o, relation, x, "max", "add"))
    v = generalized_rspmm(sparse, torch.from_numpy(relation[:, 0].copy()), torch.from_numpy(x[:, 0].copy()))
    assert v.shape == (n,)
    assert np.array_equal(v.numpy(), oracle.rspmm_forward(csr_o, relation[:, :1].copy(), x[:, :1].copy())[:, 0])


def test_cpu_build_relcsr_equals_oracle_coalesce(oracle):
    """``ultra_mi::build_relcsr`` on the CPU key: torchdrug's (node_in, node_out, relation) rows -> CSR over destinations,
    duplicates merged by sequential weight sum in input order, ``edge_of_input`` = position of every input edge."""
    n, r = 80, 5
    g = random_graph(seed=4, n_node=n, n_edge=3000, n_rel=r, weights=True)       # many duplicate triples
    edge_list = torch.from_numpy(np.stack([g["src"], g["dst"], g["rel"]], axis=1))
    row_ptr, src, rel, w, edge_of_input = torch.ops.ultra_mi.build_relcsr(edge_list, torch.from_numpy(g["w"]), n, r)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], g["w"], n, n, r)
    assert csr_o.n_edges < 3000                                                  # duplicates did merge
    assert np.array_equal(row_ptr.numpy(), csr_o.row_ptr) and np.array_equal(src.numpy(), csr_o.col)
    assert np.array_equal(rel.numpy(), csr_o.rel) and np.array_equal(w.numpy(), csr_o.w)
    dst_of = np.repeat(np.arange(n), np.diff(csr_o.row_ptr))
    e = edge_of_input.numpy()
    assert np.array_equal(dst_of[e], g["dst"]) and np.array_equal(csr_o.col[e], g["src"]) and np.array_equal(csr_o.rel[e], g["rel"])
    unit = torch.ops.ultra_mi.build_relcsr(edge_list, None, n, r)
    assert np.array_equal(unit[0].numpy(), csr_o.row_ptr)
    with pytest.raises(RuntimeError, match="out of range"):
        torch.ops.ultra_mi.build_relcsr(edge_list, None, n - 40, r)
    empty = torch.ops.ultra_mi.build_relcsr(torch.zeros(0, 3, dtype=torch.long), None, 3, 2)
    assert empty[0].tolist() == [0, 0, 0, 0] and empty[1].numel() == 0


def test_cpu_operator_at_codexs_shape_is_the_sequential_oracle(oracle):
    """BASELINE config-2 graph, full model width of one side (F = 1 024): the CPU product kernels == oracle(piece = 0)."""
    from ultra_torchdrug_amd import generalized_rspmm
    n, triples, base_rel, F = 2034, 32888, 42, 1024
    g = kg_graph(1024, n, triples, base_rel)
    R = 2 * base_rel
    rng = np.random.default_rng(1)
    relation, x = rng.standard_normal((R, F)).astype(np.float32), rng.standard_normal((n, F)).astype(np.float32)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], None, n, n, R)
    got = generalized_rspmm(_csr(g, n, R), torch.from_numpy(relation), torch.from_numpy(x))
    assert np.array_equal(got.numpy(), oracle.rspmm_forward(csr_o, relation, x, "add", "mul", piece=0))


# FB15k237Inductive v1 (GraIL split) sizes, see tests/test_baseline_configs_gpu.py
FB_V1 = dict(n_rel=180, train=(1594, 4245), inference=(1093, 1993, 206, 205))


def _inductive_task(seed=1024):
    from ultra_torchdrug_amd.data import synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.task import build_ultra
    r = FB_V1["n_rel"]
    n_tr, t_tr = FB_V1["train"]
    n_inf, t_fact, t_valid, t_test = FB_V1["inference"]
    train, _, _ = synthetic_triples((n_tr, t_tr, r), seed)
    inf, _, _ = synthetic_triples((n_inf, t_fact + t_valid + t_test, r), seed + 1)
    torch.manual_seed(seed)
    task = build_ultra(r)
    g_train = Graph(torch.from_numpy(train), num_node=n_tr, num_relation=r)
    g_fact = Graph(torch.from_numpy(inf[:t_fact]), num_node=n_inf, num_relation=r)
    g_all = Graph(torch.from_numpy(inf), num_node=n_inf, num_relation=r)
    task.preprocess_inductive(g_train, g_train, g_fact, graph=g_train, inductive_graph=g_all)
    return task.eval().use("test"), torch.from_numpy(inf[t_fact + t_valid:])


def test_config1_inductive_zero_shot_inference_runs_on_cpu_and_matches_the_oracle_path():
    """Config 1 (`--gpus null`): ``task.predict`` + filtered ranking on CPU tensors with NO backend installed -- the
    product's own CPU kernels behind ``generalized_rspmm``, ATen for the dense layers as in the reference -- against the
    same model with the oracle (sequential order) behind the operator.  The operator is bit-equal (tests above); the
    dense layers run ATen on one side and the oracle's documented order on the other, hence the fp32 tolerance on
    scores (1e-4) and rank equality wherever the positive is not within that distance of a competitor."""
    from ultra_torchdrug_amd import backend, engine, functional
    task, test = _inductive_task()
    assert backend.get() is functional and task.device.type == "cpu"
    batch = test[:32]
    with torch.no_grad():
        pred = torch.cat([task.predict(batch[i:i + 16]) for i in (0, 16)])
        mask, target = task.target(batch)
        rank = task.get_ranking(pred, (mask, target))
    assert pred.shape == (32, 2, 1093) and torch.isfinite(pred).all()
    with torch.no_grad(), oracle_rspmm(0):
        pred_o = torch.cat([task.predict(batch[i:i + 16]) for i in (0, 16)])
        rank_o = task.get_ranking(pred_o, task.target(batch))
    diff = (pred - pred_o).abs().max().item()
    assert diff <= 1e-4, diff
    pos = pred_o.gather(-1, target.unsqueeze(-1))
    gap = torch.where(mask, (pred_o - pos).abs(), torch.full_like(pred_o, float("inf")))
    gap.scatter_(-1, target.unsqueeze(-1), float("inf"))
    safe = gap.min(dim=-1).values > 2 * diff + 1e-7
    assert safe.float().mean() > 0.8 and torch.equal(rank[safe], rank_o[safe])
    # the evaluation driver on CPU: same ranks as the loop above, metrics of task.py:317-351
    metric, ranking = engine.evaluate(task, batch, batch_size=16)
    assert torch.equal(ranking, rank) and 0 < float(metric["mrr"]) <= 1


def test_cpu_training_step_runs_and_matches_the_oracle_path():
    """A fine-tuning step on CPU tensors (edge removal by zero weights, strict negatives by masks, autograd through the
    CPU rspmm_bwd): loss and gradients against the oracle-backed operator."""
    from ultra_torchdrug_amd.data import synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.task import build_ultra
    triples, n, r = synthetic_triples((150, 900, 5), 3)
    torch.manual_seed(3)
    task = build_ultra(r, num_negative=8)
    task.preprocess(Graph(torch.from_numpy(triples), num_node=n, num_relation=r)).train()
    batch = torch.from_numpy(triples[:8])
    neg = task._strict_negative(*batch.t())
    task._static_negative = neg

    def step():
        task.zero_grad()
        loss, _ = task(batch)
        loss.backward()
        return loss.item(), {k: p.grad.clone() for k, p in task.named_parameters() if p.grad is not None}
    loss, grads = step()
    with oracle_rspmm(0):
        loss_o, grads_o = step()
    assert abs(loss - loss_o) <= 1e-6 * max(1.0, abs(loss_o))
    assert grads.keys() == grads_o.keys() and len(grads) > 80
    for k in grads:
        scale = grads_o[k].abs().max().item() + 1e-8
        assert (grads[k] - grads_o[k]).abs().max().item() <= 1e-5 * scale + 1e-7, k


def test_real_split_directory_and_checkpoint_produce_metrics(tmp_path):
    """The hook for real data ("if real TSV triples are present ... they are read"; VERDICT r3 missing 5): a split
    directory in the reference's layout (train / valid / test.txt of ``h<TAB>r<TAB>t`` lines, ONE vocabulary in file order,
    /root/reference/ultra/dataset.py:33-96) plus a checkpoint in the reference's layout (``{"model": state_dict, ...}``,
    /root/reference/ultra/util.py:233-276,319-323) go through ``data.task_from_split_dir`` and ``engine.evaluate`` and give
    MR / MRR / Hits@k (ultra/task.py:317-351) -- the same ranks as a task assembled by hand from the same ids and weights.
    ``bench.py --data DIR --ckpt PATH`` is this call on the GPU."""
    from ultra_torchdrug_amd import engine
    from ultra_torchdrug_amd.checkpoint import save_checkpoint
    from ultra_torchdrug_amd.data import load_split_dir, synthetic_triples, task_from_split_dir
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.task import build_ultra
    triples, n, r = synthetic_triples((80, 420, 5), 21)
    counts = [340, 40, 40]
    bounds = np.cumsum([0] + counts)
    for i, name in enumerate(("train.txt", "valid.txt", "test.txt")):
        rows = triples[bounds[i]:bounds[i + 1]]
        (tmp_path / name).write_text("".join("/m/e%d\t/rel/r%d\t/m/e%d\n" % (h, rel, t) for h, t, rel in rows))
    ids, got_counts, n_node, n_rel = load_split_dir(str(tmp_path))
    assert got_counts == counts and ids.shape == (420, 3) and n_node <= n and n_rel <= r
    # ids are handed out in order of first appearance over train, valid, test (dataset.py:69-96): a relabelling of `triples`
    ent, rel = {}, {}
    for h, t, q in triples:
        ent.setdefault(h, len(ent)); ent.setdefault(t, len(ent)); rel.setdefault(q, len(rel))
    assert ids.tolist() == [[ent[h], ent[t], rel[q]] for h, t, q in triples]

    torch.manual_seed(5)
    donor = build_ultra(n_rel)
    ckpt = tmp_path / "td_ultra_like.pth"
    save_checkpoint(donor, str(ckpt))
    task, splits = task_from_split_dir(str(tmp_path), checkpoint=str(ckpt))
    assert task.checkpoint_keys == ([], []) and not task.training
    assert [len(splits[k]) for k in ("train", "valid", "test")] == counts
    assert task.fact_graph.num_edge == counts[0] and task.graph.num_edge == 420
    for (k, a), (_, b) in zip(task.state_dict().items(), donor.state_dict().items()):
        assert torch.equal(a, b), k
    metric, ranking = engine.evaluate(task, splits["test"], batch_size=16)
    assert ranking.shape == (40, 2) and int(ranking.min()) >= 1 and int(ranking.max()) <= n_node
    assert set(task.metric) <= set(metric) and 0 < float(metric["mrr"]) <= 1

    by_hand = build_ultra(n_rel)
    by_hand.load_state_dict(donor.state_dict())
    mask = torch.zeros(420, dtype=torch.bool)
    mask[:counts[0]] = True
    by_hand.preprocess(Graph(torch.from_numpy(ids), num_node=n_node, num_relation=n_rel), mask).eval()
    _, want = engine.evaluate(by_hand, torch.from_numpy(ids[bounds[2]:]), batch_size=16)
    assert torch.equal(ranking, want)


def test_cpu_operators_refuse_a_malformed_csr():
    """the CPU kernels of ``torch.ops.ultra_mi.rspmm_fwd / rspmm_bwd`` trust ``row_ptr`` / ``src`` / ``rel``; a row
    pointer that runs backwards or past E, a source outside ``input`` or a relation outside ``relation`` must raise instead of
    reading (and, in the backward sweep, writing) out of bounds."""
    from ultra_torchdrug_amd import _torch_ext
    ops = _torch_ext.load()
    row_ptr = torch.tensor([0, 2, 3], dtype=torch.int32)
    src = torch.tensor([0, 1, 1], dtype=torch.int32)
    rel = torch.tensor([0, 1, 0], dtype=torch.int32)
    relation, x, g = torch.randn(2, 8), torch.randn(2, 8), torch.randn(2, 8)
    ok = ops.rspmm_fwd(row_ptr, src, rel, None, relation, x, 0, 0)
    assert ok.shape == (2, 8)
    bad = [(torch.tensor([0, 3, 2], dtype=torch.int32), src, rel, "row_ptr"),          # runs backwards, ends short of E
           (torch.tensor([0, 2, 4], dtype=torch.int32), src, rel, "row_ptr"),          # ends past E
           (torch.tensor([1, 2, 3], dtype=torch.int32), src, rel, "row_ptr"),          # does not start at 0
           (row_ptr, torch.tensor([0, 2, 1], dtype=torch.int32), rel, "src"),           # source row 2 of a 2-row input
           (row_ptr, torch.tensor([0, -1, 1], dtype=torch.int32), rel, "src"),
           (row_ptr, src, torch.tensor([0, 2, 0], dtype=torch.int32), "rel"),           # relation 2 of a 2-row table
           (row_ptr, src, torch.tensor([-3, 1, 0], dtype=torch.int32), "rel")]
    for rp, s, r, what in bad:
        with pytest.raises(RuntimeError, match=what):
            ops.rspmm_fwd(rp, s, r, None, relation, x, 0, 0)
        with pytest.raises(RuntimeError, match=what):
            ops.rspmm_bwd(rp, s, r, None, relation, x, ok, g, 0, 0)


def test_eager_forward_refuses_ids_outside_the_graph():
    """the fused training paths index hidden states with the candidate ids (LDS bitmap, gathered and scattered
    rows) without a device-side bounds check; with the reference's per-call index asserts on (eager steps) an id outside the
    active graph -- e.g. a negative drawn for another split's vocabulary -- raises, as ATen's index kernel does there."""
    from ultra_torchdrug_amd.data import synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.task import build_ultra
    triples, n, r = synthetic_triples((60, 300, 3), 9)
    torch.manual_seed(9)
    task = build_ultra(r, num_negative=4)
    task.preprocess(Graph(torch.from_numpy(triples), num_node=n, num_relation=r)).train()
    batch = torch.from_numpy(triples[:4])
    neg = task._strict_negative(*batch.t())
    for wrong in (n, n + 7, -1):
        task._static_negative = neg.clone()
        task._static_negative[0, 1] = wrong                    # a candidate tail
        with pytest.raises(IndexError, match="entity ids"):
            task(batch)
        task._static_negative = neg.clone()
        task._static_negative[3, 0] = wrong                    # a candidate head (second half of the batch)
        with pytest.raises(IndexError, match="entity ids"):
            task(batch)
    task._static_negative = neg
    loss, _ = task(batch)
    assert torch.isfinite(loss)


def test_cpu_path_equals_the_aten_definition_of_the_reference():
    """The product's CPU kernels against the reference's OWN O(E) definition of the operator (``message`` + ``aggregate``,
    /root/reference/ultra/layer.py:232-296) with every other operator of the stack in ATen too (tests/aten_definition.py) -- no
    C oracle on either side.  A whole evaluation batch and a whole training step; the definition in fp64 is the truth, and the
    product path must be as close to it as the definition in fp32 is."""
    from aten_definition import aten_definition
    from ultra_torchdrug_amd.data import synthetic_triples
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.task import build_ultra
    triples, n, r = synthetic_triples((150, 900, 5), 3)
    mask = np.zeros(len(triples), dtype=bool)
    mask[:800] = True
    torch.manual_seed(3)
    task = build_ultra(r, num_negative=8)
    task.preprocess(Graph(torch.from_numpy(triples), num_node=n, num_relation=r), torch.from_numpy(mask)).eval()
    tr = torch.from_numpy(triples)
    batch = tr[800:816]
    with torch.no_grad():
        pred = task.predict(batch)
        rank = task.rank_batch(batch, pred=pred)
        with aten_definition(task):
            pred_aten = task.predict(batch)
        with aten_definition(task, double=True):
            pred_true = task.predict(batch)
            rank_true = task.get_ranking(pred_true, task.target(batch))
    assert pred_true.dtype == torch.float64 and all(p.dtype == torch.float32 for p in task.parameters())
    scale = pred_true.abs().max().item()
    e, e_aten = (pred.double() - pred_true).abs().max().item(), (pred_aten.double() - pred_true).abs().max().item()
    assert e <= 4 * e_aten + 1e-5 * scale and (rank == rank_true).float().mean() > 0.9

    task.train()
    b = tr[:8]
    neg = task._strict_negative(*b.t())

    def step():
        task.zero_grad(set_to_none=True)
        task._static_negative = neg
        try:
            loss, _ = task(b)
            loss.backward()
        finally:
            task._static_negative = None
        return float(loss), {k: p.grad.double().clone() for k, p in task.named_parameters() if p.grad is not None}
    loss, g = step()
    with aten_definition(task):
        loss_aten, g_aten = step()
    with aten_definition(task, double=True):
        loss_true, g_true = step()
    assert g.keys() == g_aten.keys() == g_true.keys() and len(g_true) == 82
    assert abs(loss - loss_true) <= 4 * abs(loss_aten - loss_true) + 1e-5 * abs(loss_true)
    for k in g_true:
        s = g_true[k].abs().max().item() + 1e-12
        e, e_aten = (g[k] - g_true[k]).abs().max().item(), (g_aten[k] - g_true[k]).abs().max().item()
        assert e <= 4 * e_aten + 1e-4 * s, (k, e, e_aten, s)


def test_cpu_forward_kernel_gives_the_same_bits_on_every_isa_level():
    """The CPU forward kernel is compiled three times (baseline x86-64, AVX2, AVX-512: the library is built once and travels
    to other hosts) and picks a level at run time; ``ULTRA_CPU_ISA`` caps it.  Every level this host supports must give the
    SAME bytes for all six operator pairs (element-wise vector code in the row's edge order, no FMA, no reassociation) --
    checked in one child process per level against this process's result, which the other tests hold to the oracle."""
    import hashlib
    import subprocess
    import sys
    script = r"""
import hashlib, sys, numpy as np, torch
sys.path.insert(0, %r)
from ultra_torchdrug_amd import _torch_ext
ops = _torch_ext.load()
rng = np.random.default_rng(12)
n, r, e, F = 300, 7, 5000, 328            # 328 = 256 + 64 + 8: a full AVX-512 slab, a full AVX2 slab and a ragged tail
dst = np.sort(rng.integers(0, n, e)); src = rng.integers(0, n, e); rel = rng.integers(0, r, e)
row_ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=n))]).astype(np.int32))
src_t, rel_t = torch.from_numpy(src.astype(np.int32)), torch.from_numpy(rel.astype(np.int32))
w = torch.from_numpy(rng.uniform(0.25, 2.0, e).astype(np.float32))
relation = torch.from_numpy(rng.standard_normal((r, F)).astype(np.float32))
x = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32))
h = hashlib.sha256()
for s in range(3):
    for m in range(2):
        for weight in (None, w):
            h.update(ops.rspmm_fwd(row_ptr, src_t, rel_t, weight, relation, x, s, m).numpy().tobytes())
print("DIGEST", h.hexdigest())
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for level in ("0", "1", "2"):
        run = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300,
                             env=dict(os.environ, ULTRA_CPU_ISA=level, OMP_NUM_THREADS="2"))
        lines = [line for line in run.stdout.splitlines() if line.startswith("DIGEST")]
        assert run.returncode == 0 and lines, run.stderr[-1500:]
        digests[level] = lines[-1].split()[1]
    assert len(set(digests.values())) == 1, digests

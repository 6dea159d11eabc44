"""Host-side mirror of ``torchdrug.layers.functional.generalized_rspmm`` for the MI355X HIP library.

Reference call sites: ``/root/reference/ultra/layer.py:134-167`` and ``:336-369``, always
``functional.generalized_rspmm(adjacency, relation_input, input, sum=<"add"|"max"|"min">, mul=<"mul"|"add">)``.
Same name, argument order, keyword names and error behaviour (unknown operator names raise ``ValueError``;
shape/device problems raise ``RuntimeError``).  Extension: ``sparse`` may be a cached :class:`RelCSR` instead of
a 3-D sparse COO tensor, so the sort torchdrug repeats on every call happens once per graph.

Everything numerical happens in native code; this module only validates, allocates outputs with torch and passes raw
pointers plus the current HIP stream across the C ABI of ``libultra_rspmm.so``.  CPU tensors (the reference's
``--gpus null`` runs, ``/root/reference/README.md:79,90``) go to the CPU kernels of the same dispatcher operators
(``torch.ops.ultra_mi.rspmm_fwd / rspmm_bwd``, ``csrc/torch_ext.cpp``): the operator itself, in the reference's
sequential order; the fused / plan-based extras of this module (boundary epilogue, frontier, fused layers, device-side
ranking and sampling) are MI355X-only and raise on CPU tensors.  Without the native libraries every operator raises: there
is no Python or PyTorch fallback, and the CPU oracle under ``oracle/`` is test infrastructure that is never imported here.
"""
import collections

import torch

from . import _lib, _torch_ext
from .relcsr import RelCSR

__all__ = ["generalized_rspmm", "rspmm_forward", "first_layer_forward", "dense_layer_forward", "combine_forward", "combine", "linear_forward", "linear_supported", "score_all_entities", "relation_stack_inputs", "statistics", "bce_adversarial_loss", "candidate_tiles", "candidate_rows", "score_candidates", "RelCSR"]

# Plans built from raw sparse tensors, most recent last.  Every entry holds strong references to the index and
# value tensors it was built from, so a (data_ptr, version) key cannot be reused by another live tensor.
_sparse_cache = collections.OrderedDict()
_SPARSE_CACHE_ENTRIES = 4


def accepts(tensor):
    """Backend interface (``backend.py``): the HIP library computes on device tensors only."""
    return bool(tensor.is_cuda)


def _ops(sum, mul):
    if sum not in _lib.SUM_OPS:
        raise ValueError("Can't find a rspmm operator for sum=`%s` (expected add, min or max)" % sum)
    if mul not in _lib.MUL_OPS:
        raise ValueError("Can't find a rspmm operator for mul=`%s` (expected mul or add)" % mul)
    return _lib.SUM_OPS[sum], _lib.MUL_OPS[mul]


def _as_relcsr(sparse):
    if isinstance(sparse, RelCSR):
        return sparse, None
    if not isinstance(sparse, torch.Tensor) or not sparse.is_sparse:
        raise TypeError("sparse must be a 3-D torch sparse COO tensor or a RelCSR, got %s" % type(sparse).__name__)
    idx, val = sparse._indices(), sparse._values()
    key = (idx.data_ptr(), val.data_ptr(), idx._version, val._version, tuple(sparse.shape))
    hit = _sparse_cache.get(key)
    if hit is None:
        hit = (RelCSR.from_sparse(sparse), idx, val)
        _sparse_cache[key] = hit
        while len(_sparse_cache) > _SPARSE_CACHE_ENTRIES:
            _sparse_cache.popitem(last=False)
    else:
        _sparse_cache.move_to_end(key)
    return hit[0], (sparse if sparse.requires_grad else None)


def _check_dense(csr, relation, input, require_hip=True):
    n_dst, n_src, n_rel = csr.shape
    if input.dim() != 2 or relation.dim() != 2:
        raise RuntimeError("relation and input must be 2-D, got %s and %s" % (tuple(relation.shape), tuple(input.shape)))
    if input.shape[0] != n_src:
        raise RuntimeError("Expect input to have %d rows, but found %d" % (n_src, input.shape[0]))
    if relation.shape[0] != n_rel:
        raise RuntimeError("Expect relation to have %d rows, but found %d" % (n_rel, relation.shape[0]))
    if relation.shape[1] != input.shape[1]:
        raise RuntimeError("Expect relation and input to have the same width, but found %d and %d"
                           % (relation.shape[1], input.shape[1]))
    if input.dtype != torch.float32 or relation.dtype != torch.float32:
        raise RuntimeError("rspmm is fp32 only (TF32 is disabled in the reference, script/run_full.py:19-20)")
    if require_hip and (not input.is_cuda or not relation.is_cuda):
        raise RuntimeError("this operator of ultra_torchdrug_amd runs on an MI355X (HIP) device only; "
                           "got input on %s, relation on %s (only generalized_rspmm has CPU kernels)"
                           % (input.device, relation.device))
    if input.device != relation.device or input.device != csr.device:
        raise RuntimeError("sparse, relation and input must be on one device (%s, %s, %s)"
                           % (csr.device, relation.device, input.device))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _workspace(seg, F, device):
    n = seg.workspace_rows * F
    if n == 0:
        return None, 0
    ws = torch.empty(n, dtype=torch.float32, device=device)
    return ws, n * 4


def rspmm_forward(csr, relation, input, sum="add", mul="mul", add_rows=None, boundary=None):
    """Forward only, no autograd.  ``add_rows`` fuses the boundary epilogue of ``layer.py:156,162,358,364``;
    ``boundary = (node, value)`` is the same epilogue with the boundary in its sparse form: ``node`` int32 ``(B,)``,
    ``value`` fp32 ``(B, D)`` with ``B * D == F`` -- row ``node[b]`` of query block ``b`` holds ``value[b]``, all else 0
    (what ``scatter_add_`` builds in ``model.py:106-107``)."""
    sum_op, mul_op = _ops(sum, mul)
    _check_dense(csr, relation, input)
    relation, input = relation.contiguous(), input.contiguous()
    F = input.shape[1]
    out = torch.empty(csr.shape[0], F, dtype=torch.float32, device=input.device)
    if add_rows is not None:
        add_rows = add_rows.contiguous()
        if add_rows.shape != out.shape or add_rows.dtype != torch.float32 or add_rows.device != out.device:
            raise RuntimeError("add_rows must be fp32 %s on %s" % (tuple(out.shape), out.device))
    if boundary is not None:
        if add_rows is not None:
            raise RuntimeError("give the boundary either dense (add_rows) or sparse (boundary), not both")
        b_node, b_value = boundary
        b_value = b_value.contiguous()
        if (b_node.dtype != torch.int32 or b_value.dtype != torch.float32 or b_node.dim() != 1 or b_value.dim() != 2
                or b_value.shape[0] != b_node.shape[0] or b_value.numel() != F or b_node.device != out.device
                or b_value.device != out.device or not b_node.is_contiguous()):
            raise RuntimeError("boundary must be (int32 (B,), fp32 (B, D)) with B * D == %d on %s" % (F, out.device))
    if out.numel() == 0:
        return out
    seg = csr.fwd
    if _torch_ext.binding() == "torch":         # through the dispatcher: torch.ops.ultra_mi (csrc/torch_ext.cpp)
        b_node_, b_value_ = (boundary[0], b_value) if boundary is not None else (None, None)
        return _torch_ext.load().rspmm_plan_fwd(seg.plan_tensor, relation, input, add_rows, b_node_, b_value_,
                                                csr.shape[1], sum_op, mul_op)
    lib = _lib.load()
    ws, ws_bytes = _workspace(seg, F, input.device)
    if boundary is not None:
        with torch.cuda.device(input.device):
            _lib.check(lib.ultra_rspmm_forward_boundary_f32(
                seg.pointer, relation.data_ptr(), input.data_ptr(), b_node.data_ptr(), b_value.data_ptr(),
                b_value.shape[1], out.data_ptr(), ws.data_ptr() if ws is not None else None, ws_bytes, csr.shape[1],
                csr.shape[2], F, sum_op, mul_op, _stream()))
        return out
    with torch.cuda.device(input.device):
        _lib.check(lib.ultra_rspmm_forward_f32(
            seg.pointer, relation.data_ptr(), input.data_ptr(), add_rows.data_ptr() if add_rows is not None else None,
            out.data_ptr(), ws.data_ptr() if ws is not None else None, ws_bytes, csr.shape[1], csr.shape[2], F, sum_op,
            mul_op, _stream()))
    return out


def rspmm_forward_csr(row_ptr, src, rel, weight, relation, input, sum="add", mul="mul"):
    """The operator straight from a coalesced CSR (``ultra_rspmm_fwd_f32``, SURVEY.md 8b's raw entry): ``row_ptr`` int32
    ``(N + 1,)``, ``src`` / ``rel`` int32 ``(E,)`` sorted by (row, src, rel), ``weight`` fp32 ``(E,)`` or ``None``.
    No plan and no workspace; every row is reduced strictly sequentially (the reference order).  Forward only."""
    sum_op, mul_op = _ops(sum, mul)
    if input.dim() != 2 or relation.dim() != 2 or relation.shape[1] != input.shape[1]:
        raise RuntimeError("relation (R, F) and input (N_src, F) expected, got %s and %s"
                           % (tuple(relation.shape), tuple(input.shape)))
    tensors = [row_ptr, src, rel, relation, input] + ([weight] if weight is not None else [])
    if any(not t.is_cuda or t.device != input.device for t in tensors):
        raise RuntimeError("rspmm_forward_csr runs on an MI355X (HIP) device only. There is no CPU fallback.")
    if row_ptr.dtype != torch.int32 or src.dtype != torch.int32 or rel.dtype != torch.int32:
        raise RuntimeError("rspmm_forward_csr: row_ptr / src / rel must be int32")
    if input.dtype != torch.float32 or relation.dtype != torch.float32 or (weight is not None and weight.dtype != torch.float32):
        raise RuntimeError("rspmm is fp32 only (TF32 is disabled in the reference, script/run_full.py:19-20)")
    F = input.shape[1]
    if F % 4:
        raise RuntimeError("rspmm_forward_csr needs 16-byte rows (F % 4 == 0); use a RelCSR plan for other widths")
    n_rows, n_edges = row_ptr.numel() - 1, src.numel()
    relation, input = relation.contiguous(), input.contiguous()
    out = torch.empty(n_rows, F, dtype=torch.float32, device=input.device)
    if out.numel() == 0:
        return out
    lib = _lib.load()
    with torch.cuda.device(input.device):
        _lib.check(lib.ultra_rspmm_fwd_f32(
            row_ptr.contiguous().data_ptr(), src.contiguous().data_ptr(), rel.contiguous().data_ptr(),
            weight.contiguous().data_ptr() if weight is not None else None, relation.data_ptr(), input.data_ptr(),
            out.data_ptr(), n_rows, n_edges, relation.shape[0], F, sum_op, mul_op, _stream()))
    return out


def frontier_supported(sum, mul, F):
    """The first-layer shortcut holds where a zero source row contributes exactly +-0: summed DistMult messages with a
    FINITE relation table (``inf * 0 = NaN`` in the full kernels; the caller tests the table, ``layer._frontier_tables_finite``)."""
    return sum == "add" and mul == "mul" and F % 64 == 0


def rspmm_frontier(csr, relation, boundary):
    """``rspmm_forward(csr, relation, <dense boundary>, "add", "mul", boundary=boundary)`` for the FIRST Bellman-Ford
    layer, whose input is the boundary itself (``ultra/model.py:116-120``): zero outside row ``node[b]`` of query block
    ``b``.  Visits only the out-edges of the boundary nodes; same bits as the full kernels (see ``csrc/frontier.inc``).
    ``boundary = (node int32 (B,), value fp32 (B, 64))``; forward only.  Precondition: ``relation`` is finite -- an
    ``inf`` / ``NaN`` entry reaches only the destinations of the boundary nodes' out-edges here, every destination of an
    edge of that relation in the full kernel (``tests/test_frontier_sampler_gpu.py::test_frontier_non_finite_*``)."""
    b_node, b_value = boundary
    b_value = b_value.contiguous()
    n_dst, n_src, n_rel = csr.shape
    F = b_value.numel()
    if relation.dim() != 2 or relation.shape != (n_rel, F) or relation.dtype != torch.float32 or not relation.is_cuda:
        raise RuntimeError("rspmm_frontier: relation must be fp32 (%d, %d) on the HIP device" % (n_rel, F))
    if (b_node.dtype != torch.int32 or b_node.dim() != 1 or b_value.dtype != torch.float32 or b_value.dim() != 2
            or b_value.shape != (b_node.shape[0], 64) or not b_node.is_contiguous()
            or b_node.device != relation.device or b_value.device != relation.device or csr.device != relation.device):
        raise RuntimeError("rspmm_frontier: boundary must be (int32 (B,), fp32 (B, 64)) on %s" % relation.device)
    if n_dst != n_src:
        raise RuntimeError("rspmm_frontier: the boundary rows index the source nodes of a square adjacency")
    relation = relation.contiguous()
    out = torch.empty(n_dst, F, dtype=torch.float32, device=relation.device)
    if out.numel() == 0:
        return out
    src_ptr, fwd_rank = csr.frontier_index
    lib = _lib.load()
    with torch.cuda.device(relation.device):
        _lib.check(lib.ultra_rspmm_frontier_f32(
            csr.by_src.pointer, src_ptr.data_ptr(), fwd_rank.data_ptr(), relation.data_ptr(), b_node.data_ptr(),
            b_value.data_ptr(), 64, out.data_ptr(), n_dst, n_rel, F, _stream()))
    return out


# Sparse first layer in inference (ultra_first_layer_sparse_f32).  ULTRA_SPARSE_FIRST_LAYER=0: frontier kernel + dense epilogue.
SPARSE_FIRST_LAYER = __import__("os").environ.get("ULTRA_SPARSE_FIRST_LAYER", "1") != "0"
# ... from this many output rows (nodes x queries) on: on a small dense graph a hub's out-edges reach most nodes, the list is
# most of the rows and its indexed epilogue loses to the dense one (S-codexs, 65 k rows: 17 us against 13; S-fb15k237, 465 k rows:
# 24 us against 108 on the hub-heaviest batch)
SPARSE_FIRST_LAYER_MIN_ROWS = 1 << 17


# The row list (and, in training, the compacted backward workspace: 5 x 64 floats per slot) is sized for the WORST head:
# n_query * (max_runs + 1) slots.  The density gates look at the degree-weighted mean; one hub that reaches most of the nodes in an
# otherwise sparse graph would pass them and allocate several (N, B, 64)-sized scratch tensors per step: the sparse
# forms are taken only where the longest list is at most this fraction of the nodes (ULTRA_SPARSE_FIRST_LAYER_MAX_LIST).
SPARSE_FIRST_LAYER_MAX_LIST = float(__import__("os").environ.get("ULTRA_SPARSE_FIRST_LAYER_MAX_LIST", "1.0"))
# ... and, in training, where the backward's compacted workspace (5 rows of 64 floats per slot) stays within this many bytes
SPARSE_FIRST_LAYER_TRAIN_MAX_BYTES = int(__import__("os").environ.get("ULTRA_SPARSE_FIRST_LAYER_TRAIN_MAX_BYTES", str(2 << 30)))


def first_layer_forward(csr, relation, boundary, weight, bias, ln_weight=None, ln_bias=None, ln_eps=1e-5, relu=True,
                        shortcut=False, want_list=False):
    """The whole FIRST Bellman-Ford layer in inference -- ``rspmm_frontier`` followed by ``combine_forward(None, update, ...,
    input_boundary=boundary)``, bit for bit -- with the epilogue on the rows the frontier reaches only: every other row of the
    layer's output is one constant vector (``relu(LN(bias))``; ``ultra/model.py:116-127``), computed once and broadcast.
    ``boundary = (node int32 (Q,), value fp32 (Q, 64))``; returns ``(N, Q, 64)``, or ``None`` where the fused entry does not
    apply (vocabulary beyond LDS, more than 128 queries, outputs beyond 4 GiB: the caller runs the two calls)."""
    if not SPARSE_FIRST_LAYER:
        return None
    b_node, b_value = boundary
    n_dst, n_src, n_rel = csr.shape
    n_query = b_node.shape[0]
    lib = _lib.load()
    if (n_dst != n_src or n_dst * n_query < SPARSE_FIRST_LAYER_MIN_ROWS
            or not lib.ultra_first_layer_sparse_supported(n_dst, n_rel, n_query)
            or csr.frontier_runs[1] + 1 > SPARSE_FIRST_LAYER_MAX_LIST * n_dst):
        return None
    F = n_query * 64
    b_value = b_value.contiguous()
    tensors = [relation, b_value, weight, bias] + ([ln_weight, ln_bias] if ln_weight is not None else [])
    if (relation.dim() != 2 or tuple(relation.shape) != (n_rel, F) or tuple(b_value.shape) != (n_query, 64) or b_node.dtype != torch.int32
            or not b_node.is_contiguous() or tuple(weight.shape) != (64, 128)
            or any(t.dtype != torch.float32 or not t.is_cuda or t.device != relation.device for t in tensors)
            or b_node.device != relation.device or csr.device != relation.device):
        raise RuntimeError("first_layer_forward: relation fp32 (%d, %d), boundary (int32 (Q,), fp32 (Q, 64)), a (64, 128) weight, "
                           "all on one HIP device" % (n_rel, F))
    relation = relation.contiguous()
    dev = relation.device
    src_ptr, fwd_rank = csr.frontier_index
    run_prefix, max_runs = csr.frontier_runs
    out = torch.empty(n_dst, n_query, 64, dtype=torch.float32, device=dev)
    row_list = torch.empty(n_query * (max_runs + 1), dtype=torch.int32, device=dev)
    list_offset = torch.empty(n_query + 1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.ultra_first_layer_sparse_f32(
            csr.by_src.pointer, src_ptr.data_ptr(), fwd_rank.data_ptr(), run_prefix.data_ptr(), relation.data_ptr(),
            b_node.data_ptr(), b_value.data_ptr(), n_query, weight.contiguous().data_ptr(), bias.contiguous().data_ptr(),
            ln_weight.contiguous().data_ptr() if ln_weight is not None else None,
            ln_bias.contiguous().data_ptr() if ln_weight is not None else None, float(ln_eps), int(bool(relu)), int(bool(shortcut)),
            out.data_ptr(), row_list.data_ptr(), row_list.numel(), int(max_runs), list_offset.data_ptr(), n_dst, n_rel, _stream()))
    if want_list:           # (out, the rows that differ from the layer's constant row, their count on the device): second_layer_sources
        return out, row_list, list_offset[n_query:]
    return out


# First layer in TRAINING sparse end to end (csrc/first_layer_train.inc).  ULTRA_SPARSE_FIRST_LAYER_TRAIN=0: the dense epilogue forward
# and backward over all N x B rows (same gradients up to the association of three 64-float sums).
SPARSE_FIRST_LAYER_TRAIN = __import__("os").environ.get("ULTRA_SPARSE_FIRST_LAYER_TRAIN", "1") != "0"
# ... on graphs where a training batch's boundary nodes (heads of sampled edges: hubs are likely) reach at most this fraction of the
# nodes each (RelCSR.frontier_fraction): the sparse form's cost grows with the listed rows, the dense form's does not.
SPARSE_FIRST_LAYER_TRAIN_MAX_FRACTION = float(__import__("os").environ.get("ULTRA_SPARSE_FIRST_LAYER_TRAIN_MAX_FRACTION", "0.2"))


def first_layer_train_forward(csr, relation, boundary, weight, bias, ln_weight=None, ln_bias=None, ln_eps=1e-5, relu=True,
                              shortcut=False):
    """The first layer of an entity Bellman-Ford in training: ``(update, out, row_list, list_count)`` -- ``update`` ``(N, Q, 64)`` holds
    the frontier's raw sums + boundary at the LISTED rows (every other row is unwritten and never read), ``out`` = the layer's output
    (the epilogue on the listed rows, one constant row elsewhere; the bits of ``rspmm_frontier`` + ``combine_forward(input_boundary=
    ...)``), ``row_list`` int32 the listed rows (``node * Q + q``, -1 = empty slot) and ``list_count`` a one-element device tensor with
    the number of slots in use -- what the epilogue's backward (``ultra_first_layer_epilogue_backward_f32``) works from.  ``None``
    where the entry does not apply."""
    if not SPARSE_FIRST_LAYER_TRAIN:
        return None
    b_node, b_value = boundary
    n_dst, n_src, n_rel = csr.shape
    n_query = b_node.shape[0]
    lib = _lib.load()
    if (n_dst != n_src or n_dst * n_query < SPARSE_FIRST_LAYER_MIN_ROWS or n_dst * n_query > (1 << 24) - 64
            or not lib.ultra_first_layer_sparse_supported(n_dst, n_rel, n_query)
            or csr.frontier_fraction > SPARSE_FIRST_LAYER_TRAIN_MAX_FRACTION
            or csr.frontier_runs[1] + 1 > SPARSE_FIRST_LAYER_MAX_LIST * n_dst
            or n_query * (csr.frontier_runs[1] + 1) * 5 * 256 > SPARSE_FIRST_LAYER_TRAIN_MAX_BYTES):
        return None
    F = n_query * 64
    b_value = b_value.contiguous()
    if (tuple(relation.shape) != (n_rel, F) or tuple(b_value.shape) != (n_query, 64) or b_node.dtype != torch.int32
            or not b_node.is_contiguous() or tuple(weight.shape) != (64, 128)):
        return None
    relation = relation.contiguous()
    dev = relation.device
    src_ptr, fwd_rank = csr.frontier_index
    run_prefix, max_runs = csr.frontier_runs
    update = torch.empty(n_dst, n_query, 64, dtype=torch.float32, device=dev)
    out = torch.empty(n_dst, n_query, 64, dtype=torch.float32, device=dev)
    row_list = torch.empty(n_query * (max_runs + 1), dtype=torch.int32, device=dev)
    list_offset = torch.empty(n_query + 1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.ultra_first_layer_sparse_train_f32(
            csr.by_src.pointer, src_ptr.data_ptr(), fwd_rank.data_ptr(), run_prefix.data_ptr(), relation.data_ptr(),
            b_node.data_ptr(), b_value.data_ptr(), n_query, weight.contiguous().data_ptr(), bias.contiguous().data_ptr(),
            ln_weight.contiguous().data_ptr() if ln_weight is not None else None,
            ln_bias.contiguous().data_ptr() if ln_weight is not None else None, float(ln_eps), int(bool(relu)), int(bool(shortcut)),
            update.data_ptr(), out.data_ptr(), row_list.data_ptr(), row_list.numel(), int(max_runs), list_offset.data_ptr(),
            n_dst, n_rel, _stream()))
    return update, out, row_list, list_offset[n_query:]


# Entity layers in inference as ONE launch (rspmm + epilogue, csrc/layer_fused.hip) on plans that run one row per lane group
# (big graphs: row pointers, no split rows).  ULTRA_FUSED_LAYER=0: the two launches (same bits).
FUSED_LAYER = __import__("os").environ.get("ULTRA_FUSED_LAYER", "1") != "0"
# ... and the last layer with the score head inside the same launch (ULTRA_FUSED_SCORE=0: layer launch + score launch)
FUSED_SCORE = __import__("os").environ.get("ULTRA_FUSED_SCORE", "1") != "0"


# The SECOND layer's gathers when the first layer was sparse: all but the listed rows of its input hold one constant vector, so
# every edge whose source is not listed gathers ONE fixed unlisted row instead (cache hits; same bits).  ULTRA_SECOND_LAYER_SOURCES=0:
# the plan's own sources.  Taken where the list is at most 1 / 64 of the nodes (beyond that the remapped gathers are DRAM
# gathers again and the pass over the edges that builds them is pure cost).
SECOND_LAYER_SOURCES = __import__("os").environ.get("ULTRA_SECOND_LAYER_SOURCES", "1") != "0"


def second_layer_sources(csr, row_list, list_count, n_query):
    """int32 ``(E + slack,)``: for every edge of the forward plan the row the SECOND layer gathers -- the edge's own source when
    the sparse first layer listed it (``row_list`` / ``list_count`` of ``first_layer_forward(want_list=True)``), one fixed
    unlisted row otherwise (``ultra_second_layer_sources``).  ``None`` where it does not apply (plans the fused layer does not
    take, lists that are not small against the node count)."""
    if not SECOND_LAYER_SOURCES or not FUSED_LAYER:
        return None
    n_dst, n_src, n_rel = csr.shape
    plan = csr.fwd
    lib = _lib.load()
    if (n_dst != n_src or not lib.ultra_layer_forward_supported(plan.pointer, n_query, n_rel)
            or row_list.numel() * 64 > n_src or row_list.numel() + 1 >= n_src):
        return None
    dev = row_list.device
    E = plan.n_edges
    slack = plan.node_a.numel() - E
    sources = torch.empty(E + slack, dtype=torch.int32, device=dev)
    bitmap = torch.empty((n_src + 31) // 32, dtype=torch.int32, device=dev)
    c_node = torch.empty(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.ultra_second_layer_sources(plan.node_a.data_ptr(), E, slack, row_list.data_ptr(), list_count.data_ptr(),
                                                  row_list.numel(), n_query, n_src, bitmap.data_ptr(), c_node.data_ptr(),
                                                  sources.data_ptr(), _stream()))
    return sources


def layer_forward(csr, relation, input, boundary, weight, bias, ln_weight=None, ln_bias=None, ln_eps=1e-5, relu=True,
                  shortcut=False, sources=None):
    """``[input +] relu(LN(Linear(cat[input, rspmm(csr, relation, input, "add", "mul") + boundary])))`` -- one entity layer of a
    Bellman-Ford in inference (/root/reference/ultra/layer.py:298-392, ultra/model.py:126-127) -- in ONE launch where the forward
    plan runs one row per lane group: ``rspmm_forward(..., boundary=)`` + ``combine_forward`` bit for bit, without the ``update``
    tensor ever reaching memory.  ``input``: ``(N, Q, 64)``; ``boundary = (node int32 (Q,), value fp32 (Q, 64))`` or ``None``.
    Returns ``(N, Q, 64)`` (a new tensor), or ``None`` where the fused entry does not apply (the caller runs the two calls)."""
    if not FUSED_LAYER or input.dim() != 3 or input.shape[-1] != 64:
        return None
    n_dst, n_src, n_rel = csr.shape
    n_query = input.shape[1]
    lib = _lib.load()
    plan = csr.fwd
    if n_dst != n_src or input.shape[0] != n_src or not lib.ultra_layer_forward_supported(plan.pointer, n_query, n_rel):
        return None
    F = n_query * 64
    b_node, b_value = (None, None) if boundary is None else boundary
    tensors = [relation, input, weight, bias] + ([ln_weight, ln_bias] if ln_weight is not None else []) + ([b_value] if b_value is not None else [])
    if (ln_weight is not None and ln_bias is None) or any(t.numel() != 64 for t in tensors[3:4 + (2 if ln_weight is not None else 0)]):
        return None                                  # (bias, LayerNorm vectors: 64 entries each -- the kernel reads 64)
    if (tuple(relation.shape) != (n_rel, F) or tuple(weight.shape) != (64, 128)
            or any(t.dtype != torch.float32 or not t.is_cuda or t.device != input.device for t in tensors)
            or (b_node is not None and (b_node.dtype != torch.int32 or tuple(b_node.shape) != (n_query,) or tuple(b_value.shape) != (n_query, 64)))):
        return None
    relation, input = relation.contiguous(), input.contiguous()
    out = torch.empty_like(input)
    if sources is not None:
        if sources.dtype != torch.int32 or sources.numel() != plan.node_a.numel() or sources.device != input.device:
            raise RuntimeError("layer_forward: sources must be second_layer_sources() of this plan")
        with torch.cuda.device(input.device):
            _lib.check(lib.ultra_layer_forward_sources_f32(
                plan.pointer, sources.data_ptr(), relation.data_ptr(), input.data_ptr(),
                b_node.contiguous().data_ptr() if b_node is not None else None,
                b_value.contiguous().data_ptr() if b_value is not None else None, n_query, weight.contiguous().data_ptr(),
                bias.contiguous().data_ptr(), ln_weight.contiguous().data_ptr() if ln_weight is not None else None,
                ln_bias.contiguous().data_ptr() if ln_weight is not None else None, float(ln_eps), int(bool(relu)), int(bool(shortcut)),
                out.data_ptr(), n_rel, _stream()))
        return out
    with torch.cuda.device(input.device):
        _lib.check(lib.ultra_layer_forward_f32(
            plan.pointer, relation.data_ptr(), input.data_ptr(), b_node.contiguous().data_ptr() if b_node is not None else None,
            b_value.contiguous().data_ptr() if b_value is not None else None, n_query, weight.contiguous().data_ptr(),
            bias.contiguous().data_ptr(), ln_weight.contiguous().data_ptr() if ln_weight is not None else None,
            ln_bias.contiguous().data_ptr() if ln_weight is not None else None, float(ln_eps), int(bool(relu)), int(bool(shortcut)),
            out.data_ptr(), n_rel, _stream()))
    return out


def layer_score_forward(csr, relation, input, boundary, weight, bias, ln_weight, ln_bias, ln_eps, relu, shortcut, query, w1, b1, w2,
                        b2):
    """The LAST entity layer of full-batch evaluation AND the score head in one launch: ``score_all_entities(layer_forward(...),
    query, w1, b1, w2, b2)`` bit for bit -- scores ``(Q, N)`` -- with the layer's ``(N, Q, 64)`` output never reaching memory
    (/root/reference/ultra/model.py:120-138,177-193).  ``None`` where the entry does not apply (the caller runs the two calls)."""
    if not FUSED_LAYER or not FUSED_SCORE or input.dim() != 3 or input.shape[-1] != 64:
        return None
    n_dst, n_src, n_rel = csr.shape
    n_query = input.shape[1]
    lib = _lib.load()
    plan = csr.fwd
    if n_dst != n_src or input.shape[0] != n_src or not lib.ultra_layer_score_supported(plan.pointer, n_query, n_rel):
        return None
    F = n_query * 64
    b_node, b_value = (None, None) if boundary is None else boundary
    tensors = [relation, input, weight, bias, query, w1, b1, w2, b2] + ([ln_weight, ln_bias] if ln_weight is not None else []) \
        + ([b_value] if b_value is not None else [])
    if (ln_weight is not None and ln_bias is None) or bias.numel() != 64 or b1.numel() != 128 \
            or (ln_weight is not None and (ln_weight.numel() != 64 or ln_bias.numel() != 64)):
        return None
    if (tuple(relation.shape) != (n_rel, F) or tuple(weight.shape) != (64, 128) or tuple(w1.shape) != (128, 128) or w2.numel() != 128
            or b2.numel() != 1 or tuple(query.shape) != (n_query, 64)
            or any(t.dtype != torch.float32 or not t.is_cuda or t.device != input.device for t in tensors)
            or (b_node is not None and (b_node.dtype != torch.int32 or tuple(b_node.shape) != (n_query,) or tuple(b_value.shape) != (n_query, 64)))):
        return None
    relation, input = relation.contiguous(), input.contiguous()
    score = torch.empty(n_query, n_src, dtype=torch.float32, device=input.device)
    qbias = torch.empty(n_query, 128, dtype=torch.float32, device=input.device)
    with torch.cuda.device(input.device):
        _lib.check(lib.ultra_layer_score_forward_f32(
            plan.pointer, relation.data_ptr(), input.data_ptr(), b_node.contiguous().data_ptr() if b_node is not None else None,
            b_value.contiguous().data_ptr() if b_value is not None else None, n_query, weight.contiguous().data_ptr(),
            bias.contiguous().data_ptr(), ln_weight.contiguous().data_ptr() if ln_weight is not None else None,
            ln_bias.contiguous().data_ptr() if ln_weight is not None else None, float(ln_eps), int(bool(relu)), int(bool(shortcut)),
            query.contiguous().data_ptr(), w1.contiguous().data_ptr(), b1.contiguous().data_ptr(), w2.contiguous().data_ptr(),
            b2.contiguous().data_ptr(), qbias.data_ptr(), score.data_ptr(), n_rel, _stream()))
    return score


# Relation-graph layers in inference as ONE launch where the plan carries its dense form (ultra_dense_layer_forward_f32).
# ULTRA_DENSE_LAYER=0: the dense rspmm and the epilogue as two launches (same bits).
DENSE_LAYER = __import__("os").environ.get("ULTRA_DENSE_LAYER", "1") != "0"


def dense_layer_forward(csr, relation, input, boundary, weight, bias, ln_weight=None, ln_bias=None, ln_eps=1e-5, relu=True,
                        shortcut=False):
    """One whole layer of a relation-graph Bellman-Ford in inference -- ``rspmm_forward(csr, relation, input.flatten(1), "add",
    "mul", boundary=boundary)`` followed by ``combine_forward(input, update, ...)``, bit for bit -- in one launch, on a graph whose
    plans carry their dense form (``/root/reference/ultra/layer.py:111-190``, ``ultra/rel_model.py:371-372``).  ``input``
    ``(N, Q, 64)``; returns a new ``(N, Q, 64)`` tensor, or ``None`` where the entry does not apply (the caller runs the two calls)."""
    if not DENSE_LAYER or not getattr(csr, "dense_form", False) or input.dim() != 3 or input.shape[-1] != 64:
        return None
    n, n_query = input.shape[0], input.shape[1]
    F = n_query * 64
    seg = csr.fwd
    lib = _lib.load()
    if seg.dense is None or csr.shape[0] != n or csr.shape[1] != n or csr.shape[2] != 4 or \
            not lib.ultra_dense_layer_supported(seg.pointer, n_query):
        return None
    b_node, b_value = boundary
    b_value = b_value.contiguous()
    tensors = [relation, input, b_value, weight, bias] + ([ln_weight, ln_bias] if ln_weight is not None else [])
    if (tuple(relation.shape) != (4, F) or tuple(b_value.shape) != (n_query, 64) or b_node.dtype != torch.int32
            or not b_node.is_contiguous() or tuple(weight.shape) != (64, 128)
            or any(t.dtype != torch.float32 or not t.is_cuda or t.device != input.device for t in tensors)
            or b_node.device != input.device or csr.device != input.device):
        raise RuntimeError("dense_layer_forward: input fp32 (N, Q, 64), relation fp32 (4, Q * 64), boundary (int32 (Q,), fp32 "
                           "(Q, 64)), a (64, 128) weight, all on one HIP device")
    relation, input = relation.contiguous(), input.contiguous()
    out = torch.empty_like(input)
    with torch.cuda.device(input.device):
        _lib.check(lib.ultra_dense_layer_forward_f32(
            seg.pointer, relation.data_ptr(), input.data_ptr(), b_node.data_ptr(), b_value.data_ptr(), n_query,
            weight.contiguous().data_ptr(), bias.contiguous().data_ptr(),
            ln_weight.contiguous().data_ptr() if ln_weight is not None else None,
            ln_bias.contiguous().data_ptr() if ln_weight is not None else None, float(ln_eps), int(bool(relu)), int(bool(shortcut)),
            out.data_ptr(), _stream()))
    return out


# Training: the backward's gathers skipped where the caller knows the gradient (last layer) or the input (first layer) rows are
# zero -- see rspmm_backward(active_dst=, active_src=).  ULTRA_ACTIVE_ROW_BACKWARD=0: every edge gathers.
ACTIVE_ROW_BACKWARD = __import__("os").environ.get("ULTRA_ACTIVE_ROW_BACKWARD", "1") != "0"


# d_relation of the first layer from the boundary nodes' out-edges alone (ultra_rspmm_drelation_boundary_f32) instead of the masked
# walk over every edge of the graph; ULTRA_BOUNDARY_DRELATION=0: the masked walk.
BOUNDARY_DRELATION = __import__("os").environ.get("ULTRA_BOUNDARY_DRELATION", "1") != "0"


def rspmm_drelation_boundary(csr, input, output_grad, b_node):
    """``d_relation`` of ``rspmm(sum="add", mul="mul")`` for an ``input`` that is zero outside row ``b_node[q]`` of 64-column block
    ``q`` (the first Bellman-Ford layer: ``ultra/model.py:106-107,116-120``): only those nodes' out-edges are visited.  Same bits as
    ``rspmm_backward(..., active_src=b_node)[1]``.  ``input`` ``(N_src, F)``, ``output_grad`` ``(N_dst, F)``, ``b_node`` int32
    ``(F / 64,)``; returns ``(R, F)``."""
    input, output_grad = input.contiguous(), output_grad.contiguous()
    dev, F = input.device, input.shape[1]
    n_dst, n_src, n_rel = csr.shape
    if (input.dtype != torch.float32 or output_grad.dtype != torch.float32 or not input.is_cuda or output_grad.device != dev
            or tuple(input.shape) != (n_src, F) or tuple(output_grad.shape) != (n_dst, F) or F % 64 != 0 or F == 0):
        raise RuntimeError("rspmm_drelation_boundary: fp32 device tensors (N_src, F) and (N_dst, F) with F a multiple of 64")
    if b_node.dtype != torch.int32 or not b_node.is_contiguous() or b_node.device != dev or tuple(b_node.shape) != (F // 64,):
        raise RuntimeError("rspmm_drelation_boundary: b_node must be contiguous int32 (%d,) on %s" % (F // 64, dev))
    items, src_ptr, src_relpos = csr.boundary_relation_index
    plan = csr.by_rel
    d_relation = torch.empty(n_rel, F, dtype=torch.float32, device=dev)
    n_ws = int(plan.n_pieces) * F
    ws = torch.empty(n_ws, dtype=torch.float32, device=dev) if n_ws else None
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.ultra_rspmm_drelation_boundary_f32(
            plan.pointer, items.data_ptr(), items.shape[0], src_ptr.data_ptr(), src_relpos.data_ptr(), b_node.data_ptr(),
            input.data_ptr(), output_grad.data_ptr(), d_relation.data_ptr(), ws.data_ptr() if ws is not None else None, n_ws * 4,
            n_src, n_rel, F, _stream()))
    return d_relation


def candidate_rows(t_index, n_node):
    """Per query ``b`` the set ``{t_index[b, j]}`` as a bitmap over the nodes, int32 ``(B, ceil(N / 32))`` -- the rows of the LAST
    layer's output the score head reads (``ultra/model.py:177-183``), i.e. the only rows where that output's gradient is
    non-zero.  One launch, static shapes (capturable).  ``None`` beyond 512 Ki nodes or when the masked backward is switched off."""
    if not ACTIVE_ROW_BACKWARD or not t_index.is_cuda or t_index.dtype != torch.int64 or t_index.dim() != 2:
        return None
    n_words = (int(n_node) + 31) // 32
    if n_words > 16384 or n_words == 0:
        return None
    t_index = t_index.contiguous()
    bits = torch.empty(t_index.shape[0], n_words, dtype=torch.int32, device=t_index.device)
    lib = _lib.load()
    with torch.cuda.device(t_index.device):
        _lib.check(lib.ultra_node_bitmap(t_index.data_ptr(), t_index.shape[0], t_index.shape[1], int(n_node), bits.data_ptr(),
                                         _stream()))
    return bits


def rspmm_backward(csr, relation, input, output, output_grad, sum="add", mul="mul", need_input=True,
                   need_relation=True, d_input_add=None, active_dst=None, active_src=None):
    """``(d_input, d_relation)``.  ``d_input_add`` (sum aggregation only): a contiguous ``(N_src, F)`` gradient the
    same rows already hold (from the layer's dense epilogue); the edge gradient is accumulated INTO it inside the
    kernel's row epilogue and the same tensor is returned -- no separate add pass.
    ``active_dst`` (sum aggregation): int32 ``(F / 64, ceil(N_dst / 32))`` bitmaps (:func:`candidate_rows`) -- the caller's
    promise that ``output_grad[v, block b]`` is zero wherever bit ``v`` of row ``b`` is clear; ``active_src`` (sum, mul = mul):
    int32 ``(F / 64,)`` -- the promise that ``input[u, block b]`` is zero unless ``u == active_src[b]``.  Same results bit for bit;
    the kernels skip the gathers of edges that can only add zero (``ultra_rspmm_backward_active_f32``)."""
    sum_op, mul_op = _ops(sum, mul)
    relation, input, output_grad = relation.contiguous(), input.contiguous(), output_grad.contiguous()
    F = input.shape[1]
    dev = input.device
    if (ACTIVE_ROW_BACKWARD and sum == "add" and F % 64 == 0 and F > 0 and (need_input or need_relation)
            and (active_dst is not None or (active_src is not None and mul == "mul"))):
        n_tiles = F // 64
        if active_dst is not None and (active_dst.dtype != torch.int32 or not active_dst.is_contiguous() or active_dst.device != dev
                                       or tuple(active_dst.shape) != (n_tiles, (csr.shape[0] + 31) // 32)):
            raise RuntimeError("rspmm_backward: active_dst must be contiguous int32 (%d, %d) on %s"
                               % (n_tiles, (csr.shape[0] + 31) // 32, dev))
        if active_src is not None and (active_src.dtype != torch.int32 or not active_src.is_contiguous() or active_src.device != dev
                                       or tuple(active_src.shape) != (n_tiles,)):
            raise RuntimeError("rspmm_backward: active_src must be contiguous int32 (%d,) on %s" % (n_tiles, dev))
        if d_input_add is not None and (not need_input or not d_input_add.is_contiguous() or d_input_add.shape != input.shape
                                        or d_input_add.dtype != torch.float32):
            raise RuntimeError("d_input_add: contiguous fp32 tensor of input's shape, sum aggregation, with need_input")
        d_input = (d_input_add if d_input_add is not None else torch.empty_like(input)) if need_input else None
        d_relation = torch.empty_like(relation) if need_relation else None
        by_src = csr.by_src if need_input else None
        by_rel = csr.by_rel if need_relation else None
        lib = _lib.load()
        n_ws = max(by_src.workspace_rows if by_src is not None else 0, by_rel.workspace_rows if by_rel is not None else 0) * F
        ws = torch.empty(n_ws, dtype=torch.float32, device=dev) if n_ws else None
        with torch.cuda.device(dev):
            _lib.check(lib.ultra_rspmm_backward_active_f32(
                by_src.pointer if by_src is not None else None, by_rel.pointer if by_rel is not None else None,
                relation.data_ptr(), input.data_ptr(), output_grad.data_ptr(),
                d_input_add.data_ptr() if d_input_add is not None else None,
                d_input.data_ptr() if d_input is not None else None, d_relation.data_ptr() if d_relation is not None else None,
                ws.data_ptr() if ws is not None else None, n_ws * 4, csr.shape[1], csr.shape[0], csr.shape[2], F, mul_op,
                active_dst.data_ptr() if active_dst is not None else None, active_dst.shape[1] if active_dst is not None else 0,
                active_src.data_ptr() if (active_src is not None and mul == "mul") else None, _stream()))
        return d_input, d_relation
    if d_input_add is not None and (not need_input or sum != "add" or not d_input_add.is_contiguous()
                                    or d_input_add.shape != input.shape or d_input_add.dtype != torch.float32):
        raise RuntimeError("d_input_add: contiguous fp32 tensor of input's shape, sum aggregation, with need_input")
    d_input = (d_input_add if d_input_add is not None else torch.empty_like(input)) if need_input else None
    d_relation = torch.empty_like(relation) if need_relation else None
    if F == 0 or (not need_input and not need_relation):
        return d_input, d_relation
    by_src = csr.by_src if need_input else None
    by_rel = csr.by_rel if need_relation else None
    if _torch_ext.binding() == "torch":
        d_rel = _torch_ext.load().rspmm_plan_bwd(
            by_src.plan_tensor if by_src is not None else None, by_rel.plan_tensor if by_rel is not None else None,
            relation, input, output, output_grad, d_input, d_input_add is not None, csr.shape[1], csr.shape[0], sum_op, mul_op)
        return d_input, (d_rel if need_relation else None)
    lib = _lib.load()
    n_ws = max(by_src.workspace_rows if by_src is not None else 0, by_rel.workspace_rows if by_rel is not None else 0) * F
    ws = torch.empty(n_ws, dtype=torch.float32, device=dev) if n_ws else None
    with torch.cuda.device(dev):
        _lib.check(lib.ultra_rspmm_backward_accumulate_f32(
            by_src.pointer if by_src is not None else None, by_rel.pointer if by_rel is not None else None,
            relation.data_ptr(), input.data_ptr(), output.data_ptr() if output is not None else None,
            output_grad.data_ptr(), d_input_add.data_ptr() if d_input_add is not None else None,
            d_input.data_ptr() if d_input is not None else None,
            d_relation.data_ptr() if d_relation is not None else None, ws.data_ptr() if ws is not None else None,
            n_ws * 4, csr.shape[1], csr.shape[0], csr.shape[2], F, sum_op, mul_op, _stream()))
    return d_input, d_relation


def rspmm_backward_boundary_rows(csr, relation, output_grad, boundary_node, d_input, mul="mul"):
    """First layer of a Bellman-Ford in training: the rspmm's edge gradient of ``input`` at the rows autograd consumes -- row
    ``boundary_node[q]`` of query block ``q`` (the layer's input is the boundary, ``ultra/model.py:106-107,116-120``) -- added
    IN PLACE into ``d_input`` ``(N_src, Q * 64)`` (``ultra_rspmm_backward_boundary_rows_f32``: the out-edges of one node per
    query instead of the ``d_input`` pass over every edge).  Sum aggregation only."""
    _, mul_op = _ops("add", mul)
    n_query = boundary_node.shape[0]
    F = n_query * 64
    if (boundary_node.dtype != torch.int32 or not boundary_node.is_contiguous() or d_input.dtype != torch.float32
            or not d_input.is_contiguous() or tuple(d_input.shape) != (csr.shape[1], F) or not output_grad.is_contiguous()
            or tuple(output_grad.shape) != (csr.shape[0], F) or output_grad.dtype != torch.float32
            or tuple(relation.shape) != (csr.shape[2], F) or not relation.is_contiguous() or csr.shape[0] != csr.shape[1]):
        raise RuntimeError("rspmm_backward_boundary_rows: contiguous fp32 (N, Q * 64) gradients, int32 (Q,) nodes, a square adjacency")
    src_ptr, _ = csr.frontier_index
    lib = _lib.load()
    ws = torch.empty(max(int(lib.ultra_rspmm_backward_boundary_rows_workspace(n_query)) // 4, 1), dtype=torch.float32,
                     device=d_input.device)
    with torch.cuda.device(d_input.device):
        _lib.check(lib.ultra_rspmm_backward_boundary_rows_f32(
            csr.by_src.pointer, src_ptr.data_ptr(), relation.data_ptr(), output_grad.data_ptr(), boundary_node.data_ptr(),
            d_input.data_ptr(), ws.data_ptr(), ws.numel() * 4, csr.shape[1], n_query, F, mul_op, _stream()))
    return d_input


def rspmm_backward_weight(csr, relation, input, output, output_grad, sum="add", mul="mul"):
    """d(values) of the coalesced edges (forward-plan order)."""
    sum_op, mul_op = _ops(sum, mul)
    relation, input, output_grad = relation.contiguous(), input.contiguous(), output_grad.contiguous()
    d_w = torch.zeros(csr.n_edges, dtype=torch.float32, device=input.device)
    if csr.n_edges == 0 or input.shape[1] == 0:
        return d_w
    lib = _lib.load()
    with torch.cuda.device(input.device):
        _lib.check(lib.ultra_rspmm_backward_weight_f32(
            csr.fwd.pointer, relation.data_ptr(), input.data_ptr(), output.data_ptr() if output is not None else None,
            output_grad.data_ptr(), d_w.data_ptr(), csr.shape[2], input.shape[1], sum_op, mul_op, _stream()))
    return d_w


def combine_forward(input, update, weight, bias, ln_weight=None, ln_bias=None, ln_eps=1e-5, relu=True, shortcut=False,
                    reuse_update=False, input_boundary=None, z_out=None):
    """Fused ``combine`` (+ shortcut) of one layer, forward only: ``[input +] relu(LN(Linear(cat[input, update])))``
    (``ultra/layer.py:386-392``, ``ultra/model.py:126-127``).  ``input`` / ``update``: ``(..., 64)`` fp32 on the GPU.
    ``reuse_update``: the caller owns ``update`` and does not need it afterwards -- the result is written over it
    (every 32-row tile is read completely before it is written), which keeps a layer's working set at two
    ``(N, B, 64)`` tensors instead of three.  ``z_out`` (training): an ``input``-shaped fp32 tensor that receives the Linear's
    output before LayerNorm, for the fused backward (``ultra_combine_backward_fused_f32``) to load instead of recompute."""
    if input_boundary is not None:
        # first layer: `input` is the boundary (model.py:116-120), given as (node int32 (B,), value fp32 (B, 64)); the kernel
        # synthesises its rows -- row (v, q) = value[q] where v == node[q], +0 elsewhere -- instead of reading (N, B, 64) zeros
        if input is not None:
            raise RuntimeError("combine_forward: give the layer input either dense or as input_boundary, not both")
        b_node, b_value = input_boundary
        b_value = b_value.contiguous()
        n_query = b_node.shape[0]
        if (update.dim() != 3 or update.shape[1] != n_query or update.shape[-1] != 64 or tuple(weight.shape) != (64, 128)
                or b_node.dtype != torch.int32 or not b_node.is_contiguous() or b_value.shape != (n_query, 64)):
            raise RuntimeError("combine_forward(input_boundary=...): update (N, B, 64), node int32 (B,), value (B, 64)")
        tensors = [update, weight, bias, b_value] + ([ln_weight, ln_bias] if ln_weight is not None else [])
        if any(t.dtype != torch.float32 or not t.is_cuda or t.device != update.device for t in tensors) or b_node.device != update.device:
            raise RuntimeError("combine_forward needs fp32 tensors on one HIP device (no CPU fallback)")
        update = update.contiguous()
        out = update if reuse_update else torch.empty_like(update)
        lib = _lib.load()
        with torch.cuda.device(update.device):
            _lib.check(lib.ultra_combine_forward_boundary_f32(
                b_node.data_ptr(), b_value.data_ptr(), n_query, update.data_ptr(), weight.contiguous().data_ptr(),
                bias.contiguous().data_ptr(), ln_weight.contiguous().data_ptr() if ln_weight is not None else None,
                ln_bias.contiguous().data_ptr() if ln_weight is not None else None, float(ln_eps), int(bool(relu)),
                int(bool(shortcut)), out.data_ptr(), update.numel() // 64, 64, _stream()))
        return out
    if input.shape != update.shape or input.shape[-1] != 64 or tuple(weight.shape) != (64, 128):
        raise RuntimeError("combine_forward handles 64 -> 64 layers with a (64, 128) weight; got input %s, update %s, "
                           "weight %s" % (tuple(input.shape), tuple(update.shape), tuple(weight.shape)))
    tensors = [input, update, weight, bias] + ([ln_weight, ln_bias] if ln_weight is not None else [])
    if any(t.dtype != torch.float32 or not t.is_cuda or t.device != input.device for t in tensors):
        raise RuntimeError("combine_forward needs fp32 tensors on one HIP device (no CPU fallback)")
    input, update = input.contiguous(), update.contiguous()
    out = update if reuse_update else torch.empty_like(input)
    rows = input.numel() // 64
    if z_out is not None and (z_out.shape != input.shape or z_out.dtype != torch.float32 or not z_out.is_contiguous()
                              or z_out.device != input.device):
        raise RuntimeError("combine_forward: z_out must be a contiguous fp32 tensor of the input's shape on its device")
    lib = _lib.load()
    with torch.cuda.device(input.device):
        _lib.check(lib.ultra_combine_forward_f32(
            input.data_ptr(), update.data_ptr(), weight.contiguous().data_ptr(), bias.contiguous().data_ptr(),
            ln_weight.contiguous().data_ptr() if ln_weight is not None else None,
            ln_bias.contiguous().data_ptr() if ln_weight is not None else None,
            float(ln_eps), int(bool(relu)), int(bool(shortcut)), out.data_ptr(),
            z_out.data_ptr() if z_out is not None else None, rows, 64, _stream()))
    return out


def linear_supported(in_dim, out_dim):
    """Shapes ``libultra_rspmm`` computes in its documented order: the relation projection and the score head."""
    return (in_dim, out_dim) in ((64, 64), (128, 128)) or (out_dim == 1 and in_dim % 4 == 0)


def linear_forward(input, weight, bias, relu=False):
    """``relu?(F.linear(input, weight, bias))`` (forward only) in the library's documented summation order: the
    small dense layers of ``ultra/layer.py:228,318-319`` and ``ultra/model.py:53,193``."""
    out_dim, in_dim = weight.shape
    if not linear_supported(in_dim, out_dim) or input.shape[-1] != in_dim or bias is None:
        raise RuntimeError("linear_forward: unsupported shape %s x %s" % (tuple(input.shape), tuple(weight.shape)))
    if any(t.dtype != torch.float32 or not t.is_cuda or t.device != input.device for t in (input, weight, bias)):
        raise RuntimeError("linear_forward needs fp32 tensors on one HIP device (no CPU fallback)")
    x = input.contiguous()
    rows = x.numel() // in_dim
    out = torch.empty(tuple(input.shape[:-1]) + (out_dim,), dtype=torch.float32, device=input.device)
    lib = _lib.load()
    with torch.cuda.device(input.device):
        _lib.check(lib.ultra_linear_forward_f32(x.data_ptr(), weight.contiguous().data_ptr(), bias.contiguous().data_ptr(),
                                                out.data_ptr(), rows, in_dim, out_dim, int(bool(relu)), _stream()))
    return out


def score_all_entities(hidden, query, w1, b1, w2, b2):
    """Score head over ALL entities, forward only: ``hidden`` ``(N, B, 64)``, ``query`` ``(B, 64)`` ->  ``(B, N)``
    scores ``mlp(cat[hidden, query])`` (``ultra/model.py:134-138,177-193`` with every entity as candidate tail)."""
    n_node, batch, dim = hidden.shape
    if dim != 64 or tuple(query.shape) != (batch, 64) or tuple(w1.shape) != (128, 128) or w2.numel() != 128:
        raise RuntimeError("score_all_entities handles the 64-d model with a 128 -> 128 -> 1 head")
    tensors = (hidden, query, w1, b1, w2, b2)
    if any(t.dtype != torch.float32 or not t.is_cuda or t.device != hidden.device for t in tensors):
        raise RuntimeError("score_all_entities needs fp32 tensors on one HIP device (no CPU fallback)")
    out = torch.empty(batch, n_node, dtype=torch.float32, device=hidden.device)
    query_bias = torch.empty(batch, 128, dtype=torch.float32, device=hidden.device)     # the queries' share of layer 1
    lib = _lib.load()
    with torch.cuda.device(hidden.device):
        _lib.check(lib.ultra_score_forward_f32(hidden.contiguous().data_ptr(), query.contiguous().data_ptr(),
                                               w1.contiguous().data_ptr(), b1.contiguous().data_ptr(),
                                               w2.contiguous().data_ptr(), b2.contiguous().data_ptr(),
                                               query_bias.data_ptr(), out.data_ptr(), n_node, batch, _stream()))
    return out


def _rows_in_place(t):
    """``t`` ``(B, R, 64)`` as the kernels can read it without a copy: rows of 64 contiguous floats at 16-byte-aligned
    strides (the relation stack returns its ``(R, B, 64)`` output transposed, ``ultra/rel_model.py:378``) -- else a
    contiguous copy."""
    if t.stride(2) == 1 and t.stride(0) % 4 == 0 and t.stride(1) % 4 == 0 and t.data_ptr() % 16 == 0:
        return t
    return t.contiguous()


def statistics(values, repeated=None, repeat=0):
    """``(norm, mean, std)`` (``Tensor.norm() / .mean() / .std()``, unbiased) of all elements of ``values`` together with the
    elements of ``repeated`` taken ``repeat`` times, as one fp32 ``(3,)`` tensor -- two launches, double accumulation
    (``ultra_statistics_f32``).  The reference's training metrics ``query_*`` / ``output_*`` (``ultra/model.py:158-160,
    178-181``); ``cat[hidden, query]`` is never built: ``statistics(hidden, query, n_node)``."""
    a = values.detach().contiguous()
    b = None if repeated is None else repeated.detach().contiguous()
    if a.dtype != torch.float32 or not a.is_cuda or (b is not None and (b.dtype != torch.float32 or b.device != a.device)):
        raise RuntimeError("statistics needs fp32 tensors on one HIP device (no CPU fallback)")
    lib = _lib.load()
    out = torch.empty(3, dtype=torch.float32, device=a.device)
    partials = torch.empty(2 * lib.ultra_statistics_blocks(a.numel()), dtype=torch.float64, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(lib.ultra_statistics_f32(a.data_ptr(), a.numel(), b.data_ptr() if b is not None else None,
                                            b.numel() if b is not None else 0, int(repeat) if b is not None else 0,
                                            partials.data_ptr(), out.data_ptr(), _stream()))
    return out


class _BCEAdversarial(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, temperature):
        pred = pred.contiguous()
        rows, cols = pred.shape
        loss = torch.empty(rows, dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred)
        lib = _lib.load()
        with torch.cuda.device(pred.device):
            _lib.check(lib.ultra_bce_adversarial_f32(pred.data_ptr(), rows, cols, float(temperature), loss.data_ptr(),
                                                     dpred.data_ptr(), _stream()))
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, grad):
        (dpred,) = ctx.saved_tensors
        return grad.unsqueeze(-1) * dpred, None


def bce_adversarial_loss(pred, temperature):
    """Per-row training loss of ``ultra/task.py:169-180`` for logits ``pred`` ``(B, 1 + K)`` (positive in column 0): binary
    cross entropy with logits, negatives weighted by ``softmax(pred[:, 1:] / temperature)`` (no gradient through the
    weights; ``temperature <= 0``: ``1 / K`` each), weighted mean per row -- forward and gradient in one launch
    (``ultra_bce_adversarial_f32``).  Returns ``(B,)``."""
    if pred.dim() != 2 or pred.dtype != torch.float32 or not pred.is_cuda:
        raise RuntimeError("bce_adversarial_loss needs fp32 (B, 1 + K) logits on a HIP device (no CPU fallback)")
    return _BCEAdversarial.apply(pred, float(temperature))


SCORE_ROWS_MAX_CANDIDATES = 160        # ultra_score_rows_backward_f32 keeps a query's rows in LDS


class _ScoreRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden, query, t_index, w1, b1, w2, b2):
        hidden, query, t_index = hidden.contiguous(), query.contiguous(), t_index.contiguous()
        w1, b1, w2, b2 = w1.contiguous(), b1.contiguous(), w2.contiguous(), b2.contiguous()
        n_batch, per_row = t_index.shape
        dev = hidden.device
        h = torch.empty(n_batch * per_row, 128, dtype=torch.float32, device=dev)
        in_rows = torch.empty(n_batch * per_row, 128, dtype=torch.float32, device=dev)
        score = torch.empty(n_batch, per_row, dtype=torch.float32, device=dev)
        lib = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(lib.ultra_score_rows_forward_f32(hidden.data_ptr(), query.data_ptr(), t_index.data_ptr(), w1.data_ptr(),
                                                        b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), h.data_ptr(), in_rows.data_ptr(),
                                                        score.data_ptr(), n_batch, per_row, _stream()))
        # the backward never reads `hidden` (its rows are in `in_rows`): only the shape is kept, so the step's largest
        # activation is not held alive until the score head's backward
        ctx.hidden_shape = tuple(hidden.shape)
        ctx.save_for_backward(query, t_index, w1, w2, h, in_rows)
        return score

    @staticmethod
    def backward(ctx, grad):
        query, t_index, w1, w2, h, in_rows = ctx.saved_tensors
        n_batch, per_row = t_index.shape
        dev = query.device
        grad = grad.contiguous()
        d_pre = torch.empty_like(h)
        partial = torch.empty(16 * 129 * 129, dtype=torch.float32, device=dev)
        d_hidden = torch.empty(ctx.hidden_shape, dtype=torch.float32, device=dev)
        d_query = torch.empty_like(query)
        d_w1 = torch.empty(128, 128, dtype=torch.float32, device=dev)
        d_b1 = torch.empty(128, dtype=torch.float32, device=dev)
        d_w2 = torch.empty(128, dtype=torch.float32, device=dev)
        d_b2 = torch.empty(1, dtype=torch.float32, device=dev)
        lib = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(lib.ultra_score_rows_backward_f32(
                None, query.data_ptr(), t_index.data_ptr(), w1.data_ptr(), w2.data_ptr(), h.data_ptr(), in_rows.data_ptr(),
                grad.data_ptr(), d_pre.data_ptr(), partial.data_ptr(), d_hidden.data_ptr(), d_query.data_ptr(), d_w1.data_ptr(), d_b1.data_ptr(), d_w2.data_ptr(),
                d_b2.data_ptr(), ctx.hidden_shape[0], n_batch, per_row, _stream()))
        return d_hidden, d_query, None, d_w1, d_b1, d_w2.view(1, 128), d_b2


def score_candidates_supported(hidden, query, t_index, w1, w2):
    """Shapes :func:`score_candidates` covers: the 64-d model with the 128 -> 128 -> 1 head, at most 160 candidates per query."""
    return (hidden.is_cuda and hidden.dtype == torch.float32 and hidden.dim() == 3 and hidden.shape[-1] == 64
            and tuple(query.shape) == (hidden.shape[1], 64) and query.dtype == torch.float32 and t_index.dtype == torch.int64
            and t_index.dim() == 2 and t_index.shape[0] == hidden.shape[1] and 0 < t_index.shape[1] <= SCORE_ROWS_MAX_CANDIDATES
            and tuple(w1.shape) == (128, 128) and w2.numel() == 128)


def score_candidates(hidden, query, t_index, w1, b1, w2, b2):
    """Scores ``(B, K)`` of the candidate tails of a training step: ``mlp(cat[hidden[t_index, arange(B)], query])`` with the
    shipped 128 -> 128 (relu) -> 1 head (``ultra/model.py:177-183,193``), differentiable in ``hidden``, ``query`` and the four
    parameters -- one forward launch, three backward launches (``ultra_score_rows_*``) instead of an index, a cat, two BLAS
    products and the ~50 launches of their backward.  ``hidden`` ``(N, B, 64)``, ``query`` ``(B, 64)``, ``t_index`` int64 ``(B, K)``."""
    if not score_candidates_supported(hidden, query, t_index, w1, w2):
        raise RuntimeError("score_candidates: (N, B, 64) / (B, 64) fp32 on a HIP device, int64 (B, K <= 160) candidates, "
                           "a 128 -> 128 -> 1 head")
    return _ScoreRows.apply(hidden, query, t_index, w1, b1, w2, b2)


def relation_graph_blocks(edge_list, n_node, n_rel):
    """The four blocks of ``construct_relation_graph`` (/root/reference/ultra/rel_model.py:99-143) for a graph WITH inverse
    edges on the device: bool ``(4, n_rel, n_rel)``, ``[type, r1, r2]`` True iff some entity is the head / tail of an ``r1``
    edge and the head / tail of an ``r2`` edge (type 0 hh, 1 tt, 2 ht, 3 th) -- the index pattern of the reference's four sparse
    incidence products, from ``ultra_relation_graph_marks`` (one wave per entity) instead of four spgemm calls over 2R rows of
    ~E / 2R entries each (117 s on S-stress; this: two key sorts + one launch).  One-off preprocessing: allocates and reads
    the host once (the list lengths)."""
    if not edge_list.is_cuda or edge_list.dtype != torch.int64 or edge_list.dim() != 2 or edge_list.shape[1] != 3:
        raise RuntimeError("relation_graph_blocks: edge_list must be int64 (E, 3) on the HIP device")
    n_node, n_rel = int(n_node), int(n_rel)
    dev = edge_list.device
    lists = []
    nodes = torch.arange(n_node + 1, device=dev) * n_rel
    for col in (0, 1):
        key = torch.unique(edge_list[:, col] * n_rel + edge_list[:, 2])          # sorted DISTINCT (entity, relation) pairs
        if key.numel() >= 2 ** 31:
            raise RuntimeError("relation_graph_blocks: more than 2^31 (entity, relation) pairs")
        ptr = torch.searchsorted(key, nodes).to(torch.int32)
        rel = (key % n_rel).to(torch.int32)
        lists += [ptr, rel]
    marks = torch.empty(4, n_rel, n_rel, dtype=torch.uint8, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.ultra_relation_graph_marks(lists[0].data_ptr(), lists[1].data_ptr(), lists[2].data_ptr(), lists[3].data_ptr(),
                                                  n_node, n_rel, marks.data_ptr(), _stream()))
    return marks.bool()


def relation_stack_inputs(weights, h_index):
    """Inputs of the relation stack's first layer in one launch (``ultra_relation_stack_inputs``): ``weights``: the layers'
    relation embeddings, each fp32 ``(R4, 64)``; ``h_index`` int64 ``(Q,)``.  Returns ``(tables (L, R4, Q * 64), ones (Q, 64),
    node32 int32 (Q,))`` -- ``stack(weights).unsqueeze(2).expand(-1, -1, Q, -1).reshape(L, R4, Q * 64)``, ``torch.ones(Q, 64)``
    and ``h_index.to(int32)`` (``ultra/rel_model.py:351-378``, ``ultra/layer.py:143-151``)."""
    import ctypes
    n = len(weights)
    keep = [w.detach().contiguous() for w in weights]
    if (n == 0 or n > 8 or h_index.dtype != torch.int64 or h_index.dim() != 1 or not h_index.is_cuda
            or any(w.dtype != torch.float32 or w.dim() != 2 or w.shape != keep[0].shape or w.shape[1] != 64 or w.device != h_index.device
                   for w in keep)):
        raise RuntimeError("relation_stack_inputs: 1..8 fp32 (R, 64) tables and int64 (Q,) indices on one HIP device")
    n_rel, n_query, dev = keep[0].shape[0], h_index.shape[0], h_index.device
    tables = torch.empty(n, n_rel, n_query * 64, dtype=torch.float32, device=dev)
    ones = torch.empty(n_query, 64, dtype=torch.float32, device=dev)
    node32 = torch.empty(n_query, dtype=torch.int32, device=dev)
    if n_query:
        lib = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(lib.ultra_relation_stack_inputs((ctypes.c_void_p * n)(*[w.data_ptr() for w in keep]), n, n_rel, n_query,
                                                       h_index.data_ptr(), h_index.stride(0), tables.data_ptr(), ones.data_ptr(),
                                                       node32.data_ptr(), _stream()))
    return tables, ones, node32


def relation_project(relation, weights, repeat=1):
    """All layers' relation projections in one launch.  ``relation``: fp32 ``(B, R, 64)``; ``weights``: one
    ``(w1, b1, w2, b2)`` per layer (``nn.Linear`` weights ``(64, 64)`` / biases ``(64,)`` of the 2-layer
    ``relation_projection`` MLP, ``ultra/layer.py:228,318-319``).  Returns one ``(R, B * 64)`` table per layer
    -- ``relation_projection(relation).transpose(0, 1).flatten(1)`` (``layer.py:325-326``), bit for bit.
    ``repeat``: the tables of ``torch.cat([relation] * repeat)`` -- ``(R, repeat * B * 64)``, query block ``b + j B`` a copy of
    block ``b`` -- computed once (full-batch evaluation: tail and head queries share the relation representations)."""
    import ctypes
    if relation.dim() != 3 or relation.shape[-1] != 64 or relation.dtype != torch.float32 or not relation.is_cuda:
        raise RuntimeError("relation_project: fp32 (B, R, 64) on a HIP device, got %s" % (tuple(relation.shape),))
    relation = _rows_in_place(relation)
    batch, n_rel, _ = relation.shape
    n = len(weights)
    keep, cols = [], [[], [], [], []]
    for layer_weights in weights:
        for k, t in enumerate(layer_weights):
            t = t.detach().contiguous()
            if t.dtype != torch.float32 or t.device != relation.device or t.shape != ((64, 64) if k % 2 == 0 else (64,)):
                raise RuntimeError("relation_project: 64 -> 64 -> 64 fp32 projections only")
            keep.append(t)
            cols[k].append(t.data_ptr())
    repeat = int(repeat)
    outs = [torch.empty(n_rel, repeat * batch * 64, dtype=torch.float32, device=relation.device) for _ in range(n)]
    if n == 0 or relation.numel() == 0:
        return outs
    arr = lambda ptrs: (ctypes.c_void_p * n)(*ptrs)
    lib = _lib.load()
    with torch.cuda.device(relation.device):
        _lib.check(lib.ultra_relation_project_f32(
            relation.data_ptr(), relation.stride(0), relation.stride(1), arr(cols[0]), arr(cols[1]), arr(cols[2]), arr(cols[3]),
            arr([o.data_ptr() for o in outs]), n, batch, repeat, n_rel, 64, _stream()))
    return outs


class _ProjectFunction(torch.autograd.Function):
    """All layers' relation projections as ONE autograd node: forward = :func:`relation_project` (one launch, the
    documented order, the tables inference uses), backward = ``ultra_relation_project_backward_f32`` (one launch + a
    small reduction) instead of autograd through 2 L ``nn.Linear`` + relu + transposes (~100 launches of a few
    microseconds per fine-tuning step)."""

    @staticmethod
    def forward(ctx, relation, *flat_weights):
        weights = [tuple(flat_weights[4 * l:4 * l + 4]) for l in range(len(flat_weights) // 4)]
        ctx.save_for_backward(relation, *flat_weights)
        tables = relation_project(relation, weights)
        return tuple(tables)

    @staticmethod
    def backward(ctx, *grads):
        import ctypes
        relation, *flat_weights = ctx.saved_tensors
        relation = relation.contiguous()
        n = len(flat_weights) // 4
        batch, n_rel, _ = relation.shape
        dev = relation.device
        keep = [None if g is None else g.contiguous() for g in grads]
        w = [t.detach().contiguous() for t in flat_weights]
        d_w = [torch.empty_like(t) for t in w]
        d_layers = torch.empty(n, batch * n_rel, 64, dtype=torch.float32, device=dev)
        lib = _lib.load()
        blocks = ctypes.c_int64(0)
        _lib.check(lib.ultra_relation_project_backward_blocks(dev.index or 0, batch, n_rel, n, ctypes.byref(blocks)))
        ws = torch.empty(n * blocks.value * (2 * 64 * 64 + 128), dtype=torch.float32, device=dev)
        arr = lambda ptrs: (ctypes.c_void_p * n)(*ptrs)
        with torch.cuda.device(dev):
            _lib.check(lib.ultra_relation_project_backward_f32(
                relation.data_ptr(), arr([w[4 * l].data_ptr() for l in range(n)]), arr([w[4 * l + 1].data_ptr() for l in range(n)]),
                arr([w[4 * l + 2].data_ptr() for l in range(n)]), arr([None if g is None else g.data_ptr() for g in keep]),
                d_layers.data_ptr(), arr([d_w[4 * l].data_ptr() for l in range(n)]), arr([d_w[4 * l + 1].data_ptr() for l in range(n)]),
                arr([d_w[4 * l + 2].data_ptr() for l in range(n)]), arr([d_w[4 * l + 3].data_ptr() for l in range(n)]),
                ws.data_ptr(), ws.numel() * 4, n, batch, n_rel, 64, _stream()))
        d_relation = d_layers.sum(0).view(batch, n_rel, 64) if ctx.needs_input_grad[0] else None
        return (d_relation, *[g if need else None for g, need in zip(d_w, ctx.needs_input_grad[1:])])


def relation_project_train(relation, weights):
    """:func:`relation_project` with gradients (``relation`` and the four parameters of every layer): the training form
    of the grouped projections.  At most 8 layers per call (the kernels' parameter block)."""
    if len(weights) > 8:
        raise RuntimeError("relation_project_train: at most 8 layers per call")
    flat = [t for layer_weights in weights for t in layer_weights]
    return list(_ProjectFunction.apply(relation, *flat))


def prepare_queries(batch, rel_rep, n_base_rel):
    """The 2B tail-form queries of one full-batch evaluation step in ONE launch (``ultra_prepare_queries``): ``batch`` int64
    ``(B, 3)`` rows of (h, t, r); ``rel_rep`` fp32 ``(B, 2 R, 64)`` (the relation stack's output for the batch's relations).
    Returns ``(anchor int64 (2B,), anchor32 int32 (2B,), relation int64 (2B,), query fp32 (2B, 64))`` -- what
    ``cat([h, t])``, ``cat([r, r + R])`` and ``cat([rel_rep, rel_rep])[arange(2B), relation]`` give (task.py:249-259,
    model.py:76-83,101-105), nine index kernels of a few microseconds each otherwise."""
    batch = batch.contiguous()
    n_batch = batch.shape[0]
    if (batch.dtype != torch.int64 or batch.dim() != 2 or batch.shape[1] != 3 or not batch.is_cuda or rel_rep.dtype != torch.float32
            or rel_rep.dim() != 3 or rel_rep.shape[0] != n_batch or rel_rep.shape[2] != 64 or rel_rep.device != batch.device
            or 2 * int(n_base_rel) != rel_rep.shape[1]):
        raise RuntimeError("prepare_queries: batch int64 (B, 3) and rel_rep fp32 (B, 2 * n_base_rel, 64) on one HIP device")
    rel_rep = _rows_in_place(rel_rep)
    dev = batch.device
    anchor = torch.empty(2 * n_batch, dtype=torch.int64, device=dev)
    anchor32 = torch.empty(2 * n_batch, dtype=torch.int32, device=dev)
    relation = torch.empty(2 * n_batch, dtype=torch.int64, device=dev)
    query = torch.empty(2 * n_batch, 64, dtype=torch.float32, device=dev)
    if n_batch:
        lib = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(lib.ultra_prepare_queries(batch.data_ptr(), rel_rep.data_ptr(), rel_rep.stride(0), rel_rep.stride(1), n_batch,
                                                 rel_rep.shape[1], int(n_base_rel),
                                                 anchor.data_ptr(), anchor32.data_ptr(), relation.data_ptr(), query.data_ptr(),
                                                 _stream()))
    return anchor, anchor32, relation, query


def filtered_rank(pred, target, filt_ptr=None, filt_node=None):
    """``sum((pos_pred <= pred) & mask, -1) + 1`` (``ultra/task.py:307-315``) with the mask given as per-row lists of
    DISTINCT filtered candidates (``filt_ptr`` int32 ``(rows + 1,)``, ``filt_node`` int32) -- no dense ``(B, N)``
    boolean.  ``pred``: fp32 ``(rows, N)``; ``target``: int64 ``(rows,)``; returns int64 ``(rows,)``."""
    pred = pred.contiguous()
    rows, n_cand = pred.shape
    target = target.contiguous()
    if target.dtype != torch.int64 or target.shape != (rows,) or pred.dtype != torch.float32:
        raise RuntimeError("filtered_rank: pred fp32 (rows, N), target int64 (rows,)")
    if filt_ptr is not None and (filt_ptr.dtype != torch.int32 or filt_node.dtype != torch.int32
                                 or filt_ptr.shape != (rows + 1,) or not filt_ptr.is_contiguous()
                                 or not filt_node.is_contiguous()):
        raise RuntimeError("filtered_rank: filter lists must be int32 (rows + 1,) / int32 (n,)")
    rank = torch.empty(rows, dtype=torch.int64, device=pred.device)
    if rows == 0:
        return rank
    lib = _lib.load()
    with torch.cuda.device(pred.device):
        _lib.check(lib.ultra_filtered_rank(
            pred.data_ptr(), rows, n_cand, n_cand, target.data_ptr(),
            filt_ptr.data_ptr() if filt_ptr is not None else None,
            filt_node.data_ptr() if filt_ptr is not None else None, rank.data_ptr(), _stream()))
    return rank


def remove_triples(graph, h, t, r, n_base_rel):
    """``graph`` (with inverse edges) minus the edges ``(h, t, r)`` / ``(t, h, r + n_base_rel)``, as zero weights on its
    cached plans: ``remove_easy_edges`` (``ultra/model.py:57-74``) for summed messages, natively and capturable."""
    if not graph.edge_list.is_cuda:
        # host graphs: the same zero weights from sorted triple keys (no device plans to patch)
        n, rels = graph.num_node, graph.num_relation
        h, t, r = h.reshape(-1), t.reshape(-1), r.reshape(-1)
        gone = torch.cat([(h * n + t) * rels + r, (t * n + h) * rels + r + n_base_rel])
        e = graph.edge_list
        keep = ~torch.isin((e[:, 0] * n + e[:, 1]) * rels + e[:, 2], gone)
        return graph.reweighted(graph.edge_weight * keep)
    return graph.without_triples(h, t, r, n_base_rel)


def filtered_rank_keys(pred, target, keys, anchor, rel, n_rel, n_node=None):
    """Filtered ranks of one prediction side without filter lists or masks: ``pred`` fp32 ``(B, N)`` (a strided view of
    the ``(B, 2, N)`` scores is fine), ``target`` / ``anchor`` / ``rel`` int64 ``(B,)``; ``keys``: the graph's sorted
    distinct completion keys of that side (``Graph.completion_keys``) or ``None`` for the unfiltered rank.  Returns
    int64 ``(B,)``.  No host synchronisation (capturable).  ``n_node``: the node count the keys were built with; a
    mismatch with ``pred.shape[1]`` raises (the kernel decodes keys with the candidate count as stride)."""
    rows, n_cand = pred.shape
    if pred.dtype != torch.float32 or pred.stride(1) != 1 or not pred.is_cuda:
        raise RuntimeError("filtered_rank_keys: pred must be fp32 (B, N) with contiguous rows on the HIP device")
    if n_node is not None and int(n_node) != n_cand:
        # the keys were built as (anchor * n_rel + rel) * n_node + other: decoded with another stride the filter reads
        # other candidates' scores (or past the row) without any error
        raise RuntimeError("filtered_rank_keys: the scores list %d candidates but the completion keys were built over "
                           "%d nodes (filter graph and fact graph must share the entity set)" % (n_cand, int(n_node)))
    for name, t in (("target", target), ("anchor", anchor), ("rel", rel)):
        if t.dtype != torch.int64 or t.shape != (rows,) or t.device != pred.device:
            raise RuntimeError("filtered_rank_keys: %s must be int64 (%d,) on %s" % (name, rows, pred.device))
    if keys is not None and (keys.dtype != torch.int64 or keys.dim() != 1 or not keys.is_contiguous()
                             or keys.device != pred.device):
        raise RuntimeError("filtered_rank_keys: keys must be a contiguous int64 vector on %s" % pred.device)
    rank = torch.empty(rows, dtype=torch.int64, device=pred.device)
    if rows == 0:
        return rank
    anchor, rel = anchor.contiguous(), rel.contiguous()
    lib = _lib.load()
    with torch.cuda.device(pred.device):
        _lib.check(lib.ultra_filtered_rank_keys(
            pred.data_ptr(), rows, n_cand, pred.stride(0), target.data_ptr(), target.stride(0),
            keys.data_ptr() if keys is not None else None, keys.numel() if keys is not None else 0, anchor.data_ptr(),
            rel.data_ptr(), 1, int(n_rel), rank.data_ptr(), 1, _stream()))
    return rank


def strict_negatives(keys, anchor, rel, n_rel, n_node, rand):
    """Strict negative sampling (``ultra/task.py:102-118``) from the graph's sorted completion keys: for every row
    ``q`` and uniform number ``rand[q, s]`` the ``floor(rand * n_free)``-th entity that does NOT complete
    ``(anchor[q], rel[q], ?)`` -- the entity ``variadic_sample(mask.nonzero()[:, 1], mask.sum(-1), S)`` returns for the
    same numbers, with no ``(B, N)`` mask and no host synchronisation.  Returns int64 ``(B, S)``."""
    rows, n_sample = rand.shape
    if rand.dtype != torch.float32 or not rand.is_cuda or not rand.is_contiguous():
        raise RuntimeError("strict_negatives: rand must be contiguous fp32 (B, S) on the HIP device")
    anchor, rel = anchor.contiguous(), rel.contiguous()
    for name, t in (("anchor", anchor), ("rel", rel)):
        if t.dtype != torch.int64 or t.shape != (rows,) or t.device != rand.device:
            raise RuntimeError("strict_negatives: %s must be int64 (%d,) on %s" % (name, rows, rand.device))
    if keys.dtype != torch.int64 or keys.dim() != 1 or not keys.is_contiguous() or keys.device != rand.device:
        raise RuntimeError("strict_negatives: keys must be a contiguous int64 vector on %s" % rand.device)
    out = torch.empty(rows, n_sample, dtype=torch.int64, device=rand.device)
    if out.numel() == 0:
        return out
    lib = _lib.load()
    with torch.cuda.device(rand.device):
        _lib.check(lib.ultra_strict_negative(keys.data_ptr(), keys.numel(), anchor.data_ptr(), rel.data_ptr(), rows,
                                             int(n_rel), int(n_node), rand.data_ptr(), n_sample, out.data_ptr(),
                                             _stream()))
    return out


# Training, opt-in (ULTRA_KEEP_PRE_NORM=1): the fused epilogue's forward keeps z = Linear(cat[input, update]) (one more
# (N, B, 64) tensor per layer) and the fused backward loads it instead of recomputing it -- a third of that kernel's matrix
# work, identical gradients.  Measured on an MI355X it only moves the backward kernel 380 -> 365 us at 655 k rows (the kernel
# waits on memory 38 % of its cycles, PMC; its matrix cores are busy 44 %) and the forward pays 12 us for writing z: no gain
# per step, hence off by default.
KEEP_PRE_NORM = __import__("os").environ.get("ULTRA_KEEP_PRE_NORM", "0") == "1"


class _CombineFunction(torch.autograd.Function):
    """Fused epilogue with a fused backward (training).  Forward = ``combine_forward``; only the layer's inputs are
    saved.  Backward: ``libultra_rspmm`` recomputes z, applies the ReLU mask and LayerNorm-backward and reduces
    ``d_weight`` on the matrix cores; ``d_input`` / ``d_update`` are two 64x64 GEMMs."""

    @staticmethod
    def forward(ctx, input, update, weight, bias, ln_weight, ln_bias, ln_eps, relu, shortcut):
        z = torch.empty_like(input, memory_format=torch.contiguous_format) if KEEP_PRE_NORM else None
        out = combine_forward(input, update, weight, bias, ln_weight, ln_bias, ln_eps, relu, shortcut, z_out=z)
        ctx.save_for_backward(input, update, weight, bias, ln_weight, ln_bias, z)
        ctx.flags = (float(ln_eps), bool(relu), bool(shortcut))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        input, update, weight, bias, ln_weight, ln_bias, z = ctx.saved_tensors
        ln_eps, relu, shortcut = ctx.flags
        grad_out = grad_out.contiguous()
        input_c, update_c = input.contiguous(), update.contiguous()
        rows = input.numel() // 64
        lib = _lib.load()
        dev = input.device
        import ctypes
        import os
        if os.environ.get("ULTRA_COMBINE_BWD", "fused") == "fused":
            # one pass over the rows: recompute z, LayerNorm / ReLU backward, d_weight, d_input | d_update (3 row-sized
            # reads + 2 writes); parameter gradients come out finished.  ULTRA_COMBINE_BWD=split: the three-kernel form.
            has_ln = ln_weight is not None
            n_waves = ctypes.c_int(0)
            _lib.check(lib.ultra_combine_backward_fused_waves(dev.index or 0, rows, ctypes.byref(n_waves)))
            ws = torch.empty(n_waves.value * (64 * 128 + 192), dtype=torch.float32, device=dev)
            d_input, d_update = torch.empty_like(input_c), torch.empty_like(update_c)
            d_weight = torch.empty(64, 128, dtype=torch.float32, device=dev)
            d_bias = torch.empty(64, dtype=torch.float32, device=dev)
            d_g = torch.empty(64, dtype=torch.float32, device=dev) if has_ln else None
            d_b = torch.empty(64, dtype=torch.float32, device=dev) if has_ln else None
            with torch.cuda.device(dev):
                _lib.check(lib.ultra_combine_backward_fused_f32(
                    input_c.data_ptr(), update_c.data_ptr(), weight.contiguous().data_ptr(), bias.contiguous().data_ptr(),
                    ln_weight.contiguous().data_ptr() if has_ln else None, ln_bias.contiguous().data_ptr() if has_ln else None,
                    ln_eps, int(relu), int(shortcut), grad_out.data_ptr(), z.data_ptr() if z is not None else None,
                    d_input.data_ptr(), d_update.data_ptr(),
                    d_weight.data_ptr(), d_bias.data_ptr(), d_g.data_ptr() if has_ln else None,
                    d_b.data_ptr() if has_ln else None, ws.data_ptr(), ws.numel() * 4, None, 0, rows, 64, _stream()))
            needs = ctx.needs_input_grad
            return (d_input.view_as(input) if needs[0] else None, d_update.view_as(update) if needs[1] else None,
                    d_weight if needs[2] else None, d_bias if needs[3] else None, d_g if (has_ln and needs[4]) else None,
                    d_b if (has_ln and needs[5]) else None, None, None, None)
        n_ln, n_wg = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(lib.ultra_combine_backward_waves(dev.index or 0, rows, ctypes.byref(n_ln), ctypes.byref(n_wg)))
        d_z = torch.empty(rows, 64, dtype=torch.float32, device=dev)
        has_ln = ln_weight is not None
        dg_p = torch.empty(n_ln.value, 64, dtype=torch.float32, device=dev) if has_ln else None
        db_p = torch.empty(n_ln.value, 64, dtype=torch.float32, device=dev) if has_ln else None
        dw_p = torch.empty(n_wg.value, 64 * 128, dtype=torch.float32, device=dev)
        dbias_p = torch.empty(n_wg.value, 64, dtype=torch.float32, device=dev) if ctx.needs_input_grad[3] else None
        with torch.cuda.device(dev):
            _lib.check(lib.ultra_combine_backward_f32(
                input_c.data_ptr(), update_c.data_ptr(), weight.contiguous().data_ptr(), bias.contiguous().data_ptr(),
                ln_weight.contiguous().data_ptr() if has_ln else None, ln_bias.contiguous().data_ptr() if has_ln else None,
                ln_eps, int(relu), grad_out.data_ptr(), d_z.data_ptr(), dg_p.data_ptr() if has_ln else None,
                db_p.data_ptr() if has_ln else None, dw_p.data_ptr(),
                dbias_p.data_ptr() if dbias_p is not None else None, rows, 64, _stream()))
        needs = ctx.needs_input_grad
        d_input = d_update = d_weight = d_bias = d_g = d_b = None
        if needs[0] and needs[1]:       # both activation gradients: one pass over d_z instead of two GEMMs
            d_input, d_update = torch.empty_like(input_c), torch.empty_like(update_c)
            with torch.cuda.device(dev):
                _lib.check(lib.ultra_combine_dxdu_f32(
                    d_z.data_ptr(), weight.contiguous().data_ptr(), grad_out.data_ptr() if shortcut else None,
                    d_input.data_ptr(), d_update.data_ptr(), rows, 64, _stream()))
            d_input, d_update = d_input.view_as(input), d_update.view_as(update)
        elif needs[0]:
            if shortcut:        # d_input = grad_out + d_z . W[:, :64] in one GEMM call (beta = 1)
                d_input = torch.addmm(grad_out.view(rows, 64), d_z, weight[:, :64]).view_as(input)
            else:
                d_input = torch.mm(d_z, weight[:, :64]).view_as(input)
        if needs[1] and d_update is None:
            d_update = torch.mm(d_z, weight[:, 64:]).view_as(update)
        if needs[2]:
            d_weight = dw_p.sum(0).view(64, 128)
        if needs[3]:
            d_bias = dbias_p.sum(0)
        if has_ln and needs[4]:
            d_g = dg_p.sum(0)
        if has_ln and needs[5]:
            d_b = db_p.sum(0)
        return d_input, d_update, d_weight, d_bias, d_g, d_b, None, None, None


# The fused inference fast paths of model.py / rel_model.py (one query-preparation kernel, projection tables computed once
# for both sides, the first layer's boundary never materialised) run on this backend; the oracle-backed backend of the
# tests keeps the reference's op-by-op path, so the two are compared against each other.
FAST_INFERENCE = __import__("os").environ.get("ULTRA_FAST_INFERENCE", "1") != "0"


def combine(input, update, weight, bias, ln_weight=None, ln_bias=None, ln_eps=1e-5, relu=True, shortcut=False,
            reuse_update=False):
    """``combine`` + shortcut of one layer (``ultra/layer.py:386-392``, ``ultra/model.py:126-127``) as fused HIP
    kernels, differentiable: same forward as :func:`combine_forward`, fused backward.  ``reuse_update`` (inference
    only): see :func:`combine_forward`."""
    tensors = [t for t in (input, update, weight, bias, ln_weight, ln_bias) if t is not None]
    if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        return _CombineFunction.apply(input, update, weight, bias, ln_weight, ln_bias, ln_eps, relu, shortcut)
    return combine_forward(input, update, weight, bias, ln_weight, ln_bias, ln_eps, relu, shortcut,
                           reuse_update=reuse_update and update.is_contiguous())


class _RSPMMFunction(torch.autograd.Function):
    """Counterpart of torchdrug's ``RSPMM{Add,Min,Max}{Mul,Add}Function`` autograd classes."""

    @staticmethod
    def forward(ctx, sparse, relation, input, csr, sum, mul, add_rows=None, b_node=None, b_value=None):
        # add_rows (sum only): `update + boundary` of layer.py:156,358 inside the kernel; its gradient is grad_out.
        # (b_node, b_value): the same boundary in sparse form -- the gradient of b_value[b] is row b_node[b] of
        # query block b of grad_out (what scatter_add_'s backward would gather from the dense gradient)
        boundary = None if b_node is None else (b_node, b_value.detach())
        out = rspmm_forward(csr, relation, input, sum, mul, add_rows=add_rows, boundary=boundary)
        ctx.has_add_rows = add_rows is not None
        ctx.b_node = b_node
        ctx.csr, ctx.sum, ctx.mul = csr, sum, mul
        ctx.sparse_meta = None
        if sparse is not None and sparse.requires_grad:
            ctx.sparse_meta = (sparse._indices(), tuple(sparse.shape))
        ctx.save_for_backward(relation, input, out if (sum != "add") else None)
        return out

    @staticmethod
    def backward(ctx, output_grad):
        relation, input, out = ctx.saved_tensors
        need_rel, need_in = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        d_input, d_relation = rspmm_backward(ctx.csr, relation, input, out, output_grad, ctx.sum, ctx.mul,
                                             need_input=need_in, need_relation=need_rel)
        d_sparse = None
        if ctx.sparse_meta is not None and ctx.needs_input_grad[0]:
            d_w = rspmm_backward_weight(ctx.csr, relation, input, out, output_grad, ctx.sum, ctx.mul)
            indices, shape = ctx.sparse_meta
            # duplicates of one triple all receive its gradient
            d_sparse = torch.sparse_coo_tensor(indices, d_w[ctx.csr.edge_of_input], shape)
        d_add = output_grad if (ctx.has_add_rows and ctx.needs_input_grad[6]) else None
        d_value = None
        if ctx.b_node is not None and ctx.needs_input_grad[8]:
            n_query = ctx.b_node.shape[0]
            blocks = output_grad.view(output_grad.shape[0], n_query, -1)
            d_value = blocks[ctx.b_node.long(), torch.arange(n_query, device=output_grad.device)]
        return d_sparse, d_relation, d_input, None, None, None, d_add, None, d_value


# Training, first layer (the caller's `input_is_boundary` promise): the edge gradient of `input` only at the boundary rows
# While a hipGraph is being captured nothing can be read back, so the shortcuts whose precondition is a FINITE relation table
# (first-layer frontier, sparse first layer, dense relation-graph form) take it on trust.  engine.capture_semantics() tests the
# model's parameters once per capture and clears this for a model that holds a non-finite one: the capture then records the
# full kernels, whose NaN propagation is the reference's.
CAPTURE_ASSUMES_FINITE = True


def tables_finite(relation):
    """Eager: one reduction and a host read; under capture: :data:`CAPTURE_ASSUMES_FINITE`."""
    if relation.is_cuda and torch.cuda.is_current_stream_capturing():
        return CAPTURE_ASSUMES_FINITE
    return bool(torch.isfinite(relation).all())


# (see rspmm_backward_boundary_rows).  ULTRA_BOUNDARY_ROWS_BACKWARD=0: the full d_input pass.
BOUNDARY_ROWS_BACKWARD = __import__("os").environ.get("ULTRA_BOUNDARY_ROWS_BACKWARD", "1") != "0"


# Training, last layer: the epilogue's backward over the candidate rows' tiles only (see sum_layer).  ULTRA_SPARSE_LAST_LAYER=0: all tiles.
SPARSE_LAST_LAYER_BACKWARD = __import__("os").environ.get("ULTRA_SPARSE_LAST_LAYER", "1") != "0"


def candidate_tiles(t_index, n_query, n_node=None):
    """The 32-row tiles of an ``(N, B, 64)`` activation (rows ``(node, query)``, query fastest) that hold the rows
    ``(t_index[b, j], b)`` -- where the gradient of ``hidden[t_index, arange(B)]`` (``ultra/model.py:177-183`` with the
    candidates first, ``model.forward``) is non-zero.  int32 ``(B * K,)``, ascending, ``-1`` padding; static shapes, no host
    synchronisation (capturable).  ``None`` when the sparse backward is switched off."""
    if not SPARSE_LAST_LAYER_BACKWARD:
        return None
    if n_node is not None:
        n_rows = int(n_node) * int(n_query)
        # worth it only where the candidates' tiles are a small part of all tiles (the rest is zero-filled: two passes)
        if t_index.numel() * 8 > (n_rows + 31) // 32:
            return None
        if t_index.is_cuda and t_index.dtype == torch.int64 and t_index.dim() == 2 and n_rows <= (32 << 20):
            t_index = t_index.contiguous()
            out = torch.empty(t_index.numel(), dtype=torch.int32, device=t_index.device)
            lib = _lib.load()
            with torch.cuda.device(t_index.device):
                _lib.check(lib.ultra_candidate_tiles(t_index.data_ptr(), t_index.shape[0], t_index.shape[1], int(n_query), n_rows,
                                                     out.data_ptr(), _stream()))
            return out
    rows = t_index.long() * int(n_query) + torch.arange(t_index.shape[0], device=t_index.device).unsqueeze(-1)
    tiles = torch.div(rows.reshape(-1), 32, rounding_mode="floor")
    ordered, _ = torch.sort(tiles)
    first = torch.ones_like(ordered, dtype=torch.bool)
    first[1:] = ordered[1:] != ordered[:-1]
    # distinct tiles first (ascending), padding behind: a stable sort on the "duplicate" flag keeps the ascending order
    keyed = torch.where(first, ordered, torch.full_like(ordered, torch.iinfo(torch.int64).max))
    keyed, _ = torch.sort(keyed)
    return torch.where(keyed == torch.iinfo(torch.int64).max, torch.full_like(keyed, -1), keyed).to(torch.int32)


class _SumLayerFunction(torch.autograd.Function):
    """One Bellman-Ford layer with summed messages as ONE autograd node:
    ``out = combine(input, rspmm(adjacency, relation, input, sum="add") + boundary)``
    (``ultra/layer.py:298-392`` + the caller's shortcut, ``ultra/model.py:126-127``).  Forward: the rspmm kernel with the
    boundary epilogue, then the fused epilogue kernel.  Backward: the one-pass epilogue backward, then the rspmm
    backward, which adds its edge gradient INTO the epilogue's ``d_input`` inside its row epilogue -- as separate
    autograd nodes the two gradients of ``input`` meet in an extra add pass over an ``(N, B, 64)`` tensor per layer."""

    @staticmethod
    def forward(ctx, csr, relation, input, add_rows, b_node, b_value, mul, weight, bias, ln_weight, ln_bias, ln_eps,
                relu, shortcut, input_is_boundary=False, grad_tiles=None, grad_rows=None):
        shape = input.shape                                      # (N, B, 64)
        flat = input.flatten(1)
        boundary = None if b_node is None else (b_node, b_value.detach())
        # what BOTH first-layer shortcuts need (frontier forward, boundary-rows backward): the caller's promise, a square
        # adjacency, one 64-column block per query, a query count the backward's grid can take -- decided HERE, so that a
        # layer outside those limits takes the full kernels in both directions instead of failing in backward
        first_layer = bool(input_is_boundary and boundary is not None and BOUNDARY_ROWS_BACKWARD
                           and frontier_supported("add", mul, flat.shape[1]) and csr.shape[0] == csr.shape[1]
                           and flat.shape[1] == 64 * b_node.shape[0] and b_node.shape[0] <= 65535)
        sparse = None
        if first_layer and tables_finite(relation):
            # first layer: only the boundary nodes' out-edges carry a message (same bits as the full kernel, finite tables:
            # see rspmm_frontier) -- as in inference; and the epilogue runs on the rows they reach only (first_layer_train_forward)
            # (only where the backward takes d_relation from the boundary nodes' out-edges: every other d_relation kernel multiplies
            # ALL d_update rows, which the sparse backward writes at the listed rows only -- ADVICE r5)
            if (mul == "mul" and grad_tiles is None and not KEEP_PRE_NORM and BOUNDARY_DRELATION
                    and not csr.kernel_order("add", "mul", flat.shape[1])[1]):
                sparse = first_layer_train_forward(csr, relation.detach(), boundary, weight, bias, ln_weight, ln_bias, ln_eps, relu,
                                                   shortcut)
            update = rspmm_frontier(csr, relation.detach(), boundary) if sparse is None else sparse[0].flatten(1)
        else:
            update = rspmm_forward(csr, relation, flat, "add", mul, add_rows=None if add_rows is None else add_rows.flatten(1),
                                   boundary=boundary)
        update = update.view(shape)
        z = torch.empty_like(update) if KEEP_PRE_NORM else None
        if sparse is None:
            out = combine_forward(input, update, weight, bias, ln_weight, ln_bias, ln_eps, relu, shortcut, z_out=z)
        else:
            out = sparse[1].view(shape)
        ctx.first_rows = None if sparse is None else (sparse[2], sparse[3])
        ctx.csr, ctx.mul, ctx.b_node, ctx.has_add = csr, mul, b_node, add_rows is not None
        # first layer: `input` is the boundary, whose gradient is consumed at row (b_node[q], q) only
        ctx.boundary_rows_only = first_layer
        # last layer: the caller's word that the output's gradient is zero outside these 32-row tiles (see sum_layer)
        ctx.grad_tiles = grad_tiles if (grad_tiles is not None and SPARSE_LAST_LAYER_BACKWARD) else None
        # ... and outside these (node, query) rows (candidate_rows): the rspmm backward gathers d_update rows there only
        ctx.grad_rows = grad_rows
        ctx.flags = (float(ln_eps), bool(relu), bool(shortcut))
        ctx.save_for_backward(relation, input, update, weight, bias, ln_weight, ln_bias, z)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        import ctypes
        relation, input, update, weight, bias, ln_weight, ln_bias, z = ctx.saved_tensors
        ln_eps, relu, shortcut = ctx.flags
        needs = ctx.needs_input_grad
        tiles = ctx.grad_tiles
        shape = input.shape
        dev = input.device
        grad_out = grad_out.contiguous()
        input_c, update_c = input.contiguous(), update.contiguous()
        rows = input.numel() // 64
        has_ln = ln_weight is not None
        lib = _lib.load()
        n_waves = ctypes.c_int(0)
        _lib.check(lib.ultra_combine_backward_fused_waves(dev.index or 0, rows, ctypes.byref(n_waves)))
        ws = torch.empty(n_waves.value * (64 * 128 + 192), dtype=torch.float32, device=dev)
        first_rows = getattr(ctx, "first_rows", None)
        # first layer: d_relation from the boundary nodes' out-edges (reads d_update at the rows those edges reach only)?
        drel_boundary = bool(ctx.boundary_rows_only and needs[2] and needs[1] and BOUNDARY_DRELATION and ctx.mul == "mul"
                             and not ctx.csr.kernel_order("add", "mul", update_c[0].numel())[1])
        d_input = torch.empty_like(input_c)
        # The sparse first-layer backward writes d_update at the LISTED rows only.  Any other consumer than the boundary kernels
        # (a knob or switch flipped between forward and backward: the full / masked d_relation kernels multiply every d_update row
        # by the zero input row, and 0 * NaN of uninitialised memory is NaN) gets zeros elsewhere.
        d_update = torch.empty_like(update_c) if (first_rows is None or drel_boundary) else torch.zeros_like(update_c)
        d_weight = torch.empty(64, 128, dtype=torch.float32, device=dev)
        d_bias = torch.empty(64, dtype=torch.float32, device=dev)
        d_g = torch.empty(64, dtype=torch.float32, device=dev) if has_ln else None
        d_b = torch.empty(64, dtype=torch.float32, device=dev) if has_ln else None
        if first_rows is not None:
            # sparse first layer (csrc/first_layer_train.inc): the listed rows compacted and run through the one-pass backward, every
            # other row's share of d_bias / d_gamma / d_beta from the column sums of grad_out; d_update is written (and later read) at
            # the listed rows only, d_input -- which leaves this node -- is zero elsewhere
            row_list, list_count = first_rows
            ws1 = torch.empty(int(lib.ultra_first_layer_epilogue_backward_workspace(dev.index or 0, row_list.numel())) // 4,
                              dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.ultra_first_layer_epilogue_backward_f32(
                    input_c.data_ptr(), update_c.data_ptr(), grad_out.data_ptr(), row_list.data_ptr(), list_count.data_ptr(),
                    row_list.numel(), weight.contiguous().data_ptr(), bias.contiguous().data_ptr(),
                    ln_weight.contiguous().data_ptr() if has_ln else None, ln_bias.contiguous().data_ptr() if has_ln else None,
                    ln_eps, int(relu), int(shortcut), d_input.data_ptr(), d_update.data_ptr(), d_weight.data_ptr(), d_bias.data_ptr(),
                    d_g.data_ptr() if has_ln else None, d_b.data_ptr() if has_ln else None, ws1.data_ptr(), ws1.numel() * 4, rows,
                    _stream()))
        with torch.cuda.device(dev):
            _lib.check(0 if first_rows is not None else lib.ultra_combine_backward_fused_f32(
                input_c.data_ptr(), update_c.data_ptr(), weight.contiguous().data_ptr(), bias.contiguous().data_ptr(),
                ln_weight.contiguous().data_ptr() if has_ln else None, ln_bias.contiguous().data_ptr() if has_ln else None,
                ln_eps, int(relu), int(shortcut), grad_out.data_ptr(), z.data_ptr() if z is not None else None,
                d_input.data_ptr(), d_update.data_ptr(),
                d_weight.data_ptr(), d_bias.data_ptr(), d_g.data_ptr() if has_ln else None,
                d_b.data_ptr() if has_ln else None, ws.data_ptr(), ws.numel() * 4,
                tiles.data_ptr() if tiles is not None else None, tiles.numel() if tiles is not None else 0, rows, 64, _stream()))
        # the edge gradient accumulates into the epilogue's d_input (same buffer) inside the rspmm backward
        flat_du = d_update.flatten(1)
        if ctx.boundary_rows_only and needs[2]:
            # first layer: the input is zero outside row b_node[q] of block q -- d_relation needs that node's out-edges only
            if drel_boundary:
                d_relation = rspmm_drelation_boundary(ctx.csr, input_c.flatten(1), flat_du, ctx.b_node)
            else:
                _, d_relation = rspmm_backward(ctx.csr, relation, input_c.flatten(1), None, flat_du, "add", ctx.mul,
                                               need_input=False, need_relation=needs[1], active_dst=ctx.grad_rows,
                                               active_src=ctx.b_node)
            d_in = rspmm_backward_boundary_rows(ctx.csr, relation.contiguous(), flat_du, ctx.b_node, d_input.flatten(1), ctx.mul)
        else:
            d_in, d_relation = rspmm_backward(ctx.csr, relation, input_c.flatten(1), None, flat_du, "add", ctx.mul,
                                              need_input=needs[2], need_relation=needs[1],
                                              d_input_add=d_input.flatten(1) if needs[2] else None, active_dst=ctx.grad_rows)
        d_add = d_update if (ctx.has_add and needs[3]) else None
        d_value = None
        if ctx.b_node is not None and needs[5]:
            n_query = ctx.b_node.shape[0]
            d_value = torch.empty(n_query, 64, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.ultra_gather_boundary_rows_f32(d_update.data_ptr(), ctx.b_node.data_ptr(), n_query,
                                                              d_value.data_ptr(), _stream()))
        return (None, d_relation, d_in.view(shape) if d_in is not None else None, d_add, None, d_value, None,
                d_weight if needs[7] else None, d_bias if needs[8] else None, d_g if (has_ln and needs[9]) else None,
                d_b if (has_ln and needs[10]) else None, None, None, None, None, None, None)


def sum_layer(csr, relation, input, boundary_dense, boundary_sparse, mul, weight, bias, ln_weight=None, ln_bias=None,
              ln_eps=1e-5, relu=True, shortcut=False, input_is_boundary=False, grad_tiles=None, grad_rows=None):
    """A whole sum-aggregation layer for TRAINING as one autograd node (see :class:`_SumLayerFunction`):
    ``[input +] relu(LN(Linear(cat[input, rspmm(csr, relation, input) + boundary])))``.  ``input``: ``(N, B, 64)``;
    ``relation``: ``(R, B * 64)``; the boundary either dense ``(N, B, 64)`` or sparse ``(node int32 (B,), value (B, 64))``.
    ``input_is_boundary``: the caller's word that ``input`` is the boundary (first layer).  ``grad_tiles``: int32 tile ids
    (row ``// 32`` of the ``(N * B, 64)`` view, ascending, ``-1`` padding) -- the caller's word that the gradient arriving at
    this layer's OUTPUT is zero outside those tiles (the last layer, whose output is read at the candidate entities' rows only,
    ``ultra/model.py:177-183``; see :func:`candidate_tiles`): the epilogue's backward then computes those tiles alone.
    ``grad_rows``: the same promise per (node, query) row as bitmaps (:func:`candidate_rows`): the rspmm backward of the layer
    then gathers gradient rows at those destinations only."""
    _check_dense(csr, relation, input.flatten(1))
    b_node, b_value = (None, None) if boundary_sparse is None else boundary_sparse
    add_rows = boundary_dense if boundary_sparse is None else None
    return _SumLayerFunction.apply(csr, relation, input, add_rows, b_node, b_value, mul, weight, bias, ln_weight, ln_bias,
                                   ln_eps, relu, shortcut, input_is_boundary, grad_tiles, grad_rows)


def rspmm_sum_plus(sparse, relation, input, add_rows, mul="mul", boundary=None):
    """``generalized_rspmm(..., sum="add") + add_rows`` with the addition inside the kernel (differentiable).
    ``boundary = (node, value)`` instead of ``add_rows``: the boundary in sparse form (see :func:`rspmm_forward`),
    differentiable in ``value``."""
    csr, sparse_leaf = _as_relcsr(sparse)
    _check_dense(csr, relation, input)
    if boundary is not None:
        if add_rows is not None:
            raise RuntimeError("give the boundary either dense (add_rows) or sparse (boundary), not both")
        return _RSPMMFunction.apply(sparse_leaf, relation, input, csr, "add", mul, None, boundary[0], boundary[1])
    return _RSPMMFunction.apply(sparse_leaf, relation, input, csr, "add", mul, add_rows)


def _rspmm_host(csr, relation, input, sum_op, mul_op):
    """CPU tensors: the raw-CSR dispatcher operator (CPU key of ``torch.ops.ultra_mi.rspmm_fwd``; its Autograd key
    reaches ``rspmm_bwd``'s CPU kernel).  Every row strictly sequentially in (src, rel) order: the reference's order."""
    row_ptr, src, rel, w = csr.csr_arrays
    return _torch_ext.load().rspmm_fwd(row_ptr, src, rel, w, relation, input, sum_op, mul_op)


def generalized_rspmm(sparse, relation, input, sum="add", mul="mul"):
    r"""Generalized relational sparse-dense product (drop-in for torchdrug's function of the same name).

    .. math::  out_{v,:} = \bigoplus_{(v, u, r) \in sparse} w_{vur} \cdot (relation_{r,:} \otimes input_{u,:})

    with :math:`\oplus` = ``sum`` in {add, min, max} and :math:`\otimes` = ``mul`` in {mul, add}.

    Parameters: ``sparse`` -- 3-D sparse COO tensor ``(N_dst, N_src, R)`` (any order, duplicates are merged by
    summing their values, as ``coalesce()`` does) or a :class:`RelCSR`; ``relation`` -- ``(R, F)`` fp32;
    ``input`` -- ``(N_src, F)`` fp32 (a 1-D ``input`` is treated as ``(N_src, 1)``).  Returns ``(N_dst, F)``.
    """
    _ops(sum, mul)
    csr, sparse_leaf = _as_relcsr(sparse)
    squeeze = input.dim() == 1
    if squeeze:
        input = input.unsqueeze(-1)
        if relation.dim() == 1:
            relation = relation.unsqueeze(-1)
    _check_dense(csr, relation, input, require_hip=False)
    if not input.is_cuda:
        if sparse_leaf is not None:
            raise RuntimeError("generalized_rspmm on CPU tensors has no gradient for the sparse values "
                               "(the reference takes its scatter path then, ultra/layer.py:299)")
        out = _rspmm_host(csr, relation, input, *_ops(sum, mul))
    else:
        out = _RSPMMFunction.apply(sparse_leaf, relation, input, csr, sum, mul)
    return out.squeeze(-1) if squeeze else out

"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED (see rspmm_oracle.c).

numpy + ctypes front-end of the CPU parity oracle for the rspmm hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; nothing under
``ultra_torchdrug_amd/`` does.

Three independent restatements of the same operator, checked against each other in
``tests/test_oracle.py``:

* :func:`rspmm_forward` / :func:`rspmm_backward`  -- the C row loop (``rspmm_oracle.c``), i.e. the
  torchdrug CSR algorithm the reference reaches at ``ultra/layer.py:134-167,336-369``;
* :func:`rspmm_materialised` -- the reference's own O(E*F) definition, ``ultra/layer.py:232-296``:
  gather ``input[node_in]``, gather ``relation_input[relation]``, combine (``:252-255``), multiply by
  ``edge_weight`` and scatter over ``node_out`` with ``dim_size=num_node`` (``:275-285``);
* :func:`rspmm_python` -- plain Python loops, for hand-checkable graphs only.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librspmm_oracle.so")

SUM_OPS = {"add": 0, "min": 1, "max": 2}
MUL_OPS = {"mul": 0, "add": 1}

_lib = None


def build(force=False):
    """Compile ``librspmm_oracle.so`` with the committed Makefile (gcc); make handles staleness."""
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_abi_version.restype = ctypes.c_int
        for name in ("oracle_rspmm_forward", "oracle_rspmm_backward", "oracle_filtered_rank", "oracle_combine_forward",
                     "oracle_linear_forward", "oracle_linear_forward_grouped"):
            getattr(_lib, name).restype = ctypes.c_int
    return _lib


def native_forward_fn(threads=None):
    """For ``bench.py``'s ``cpu_baseline`` leg only: the SAME source rebuilt ``-O3 -march=native`` (still
    ``-ffp-contract=off``) on the machine that times it, in a temporary directory -- the committed Makefile's build must
    run on any x86-64 host the snapshot travels to, a ``-march=native`` object need not.  Returns a function with the
    signature of :func:`rspmm_forward`."""
    import tempfile
    out_dir = tempfile.mkdtemp(prefix="rspmm_oracle_native_")
    so = os.path.join(out_dir, "librspmm_oracle_native.so")
    subprocess.check_call([os.environ.get("CC", "gcc"), "-O3", "-march=native", "-fPIC", "-std=c11", "-ffp-contract=off",
                           "-fno-fast-math", "-fopenmp", "-shared", "-o", so, os.path.join(_HERE, "rspmm_oracle.c"), "-lm"])
    native = ctypes.CDLL(so)
    native.oracle_rspmm_forward.restype = ctypes.c_int
    native.oracle_set_threads.restype = ctypes.c_int

    def forward(csr, relation, x, sum="add", mul="mul", piece=0):
        F = x.shape[1]
        out = np.empty((csr.n_rows, F), dtype=np.float32)
        rc = native.oracle_rspmm_forward(_p(csr.row_ptr), _p(csr.col), _p(csr.rel), _p(csr.w), _p(relation), _p(x), _p(out),
                                         _i64(csr.n_rows), _i64(csr.n_edges), _i64(csr.n_rel), _i64(F), SUM_OPS[sum],
                                         MUL_OPS[mul], _i64(piece))
        if rc:
            raise RuntimeError("oracle_rspmm_forward (native build) failed: %d" % rc)
        return out
    forward.set_threads = lambda n: int(native.oracle_set_threads(int(n)))
    return forward


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p(0)


def _i64(v):
    return ctypes.c_int64(int(v))


class CSR:
    """Coalesced CSR of the (N_dst, N_src, R) adjacency (rows = destination)."""

    def __init__(self, row_ptr, col, rel, w, n_rows, n_cols, n_rel):
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.rel = np.ascontiguousarray(rel, dtype=np.int32)
        self.w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
        self.n_rows, self.n_cols, self.n_rel = int(n_rows), int(n_cols), int(n_rel)

    @property
    def n_edges(self):
        return int(self.col.shape[0])

    @property
    def row(self):
        return np.repeat(np.arange(self.n_rows, dtype=np.int32), np.diff(self.row_ptr))


def coalesce_csr(dst, src, rel, w, n_rows, n_cols, n_rel):
    """torch ``sparse.coalesce()`` + torchdrug ``coo2csr`` [from memory]: sort the COO triples by
    (dst, src, rel) and merge duplicate triples by summing their weights (in input order)."""
    dst = np.asarray(dst, dtype=np.int64)
    src = np.asarray(src, dtype=np.int64)
    rel = np.asarray(rel, dtype=np.int64)
    w = np.ones(dst.shape[0], dtype=np.float32) if w is None else np.asarray(w, dtype=np.float32)
    assert dst.shape == src.shape == rel.shape == w.shape
    if dst.size:
        assert 0 <= dst.min() and dst.max() < n_rows and 0 <= src.min() and src.max() < n_cols
        assert 0 <= rel.min() and rel.max() < n_rel
    order = np.lexsort((rel, src, dst))  # stable, last key is primary
    dst, src, rel, w = dst[order], src[order], rel[order], w[order]
    if dst.size:
        new = np.ones(dst.shape[0], dtype=bool)
        new[1:] = (dst[1:] != dst[:-1]) | (src[1:] != src[:-1]) | (rel[1:] != rel[:-1])
        starts = np.flatnonzero(new)
        # sequential fp32 sum of the duplicates of one triple, in input order
        wsum = np.empty(starts.shape[0], dtype=np.float32)
        ends = np.append(starts[1:], dst.shape[0])
        simple = (ends - starts) == 1
        wsum[simple] = w[starts[simple]]
        for i in np.flatnonzero(~simple):
            acc = np.float32(0.0)
            for k in range(starts[i], ends[i]):
                acc = np.float32(acc + w[k])
            wsum[i] = acc
        dst, src, rel, w = dst[starts], src[starts], rel[starts], wsum
    row_ptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(row_ptr, dst + 1, 1)
    row_ptr = np.cumsum(row_ptr)
    return CSR(row_ptr, src, rel, w, n_rows, n_cols, n_rel)


def rspmm_forward(csr, relation, x, sum="add", mul="mul", piece=0):
    relation = np.ascontiguousarray(relation, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    F = x.shape[1]
    assert relation.shape == (csr.n_rel, F) and x.shape[0] == csr.n_cols
    out = np.empty((csr.n_rows, F), dtype=np.float32)
    rc = lib().oracle_rspmm_forward(_p(csr.row_ptr), _p(csr.col), _p(csr.rel), _p(csr.w), _p(relation), _p(x),
                                    _p(out), _i64(csr.n_rows), _i64(csr.n_edges), _i64(csr.n_rel), _i64(F),
                                    SUM_OPS[sum], MUL_OPS[mul], _i64(piece))
    if rc:
        raise RuntimeError("oracle_rspmm_forward failed: %d" % rc)
    return out


def rspmm_backward(csr, relation, x, out, grad, sum="add", mul="mul", piece=0, need_weight_grad=False, dense_relation=False):
    """``dense_relation``: d_relation in the documented order of the HIP library's dense relation-graph form
    (``oracle_rspmm_drelation_dense``: 4 relation types, unit weights, sum of DistMult messages) instead of ``piece``'s."""
    relation = np.ascontiguousarray(relation, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.ascontiguousarray(out, dtype=np.float32)
    grad = np.ascontiguousarray(grad, dtype=np.float32)
    F = x.shape[1]
    d_rel = np.empty((csr.n_rel, F), dtype=np.float32)
    d_x = np.empty((csr.n_cols, F), dtype=np.float32)
    d_w = np.empty(csr.n_edges, dtype=np.float32) if need_weight_grad else None
    rc = lib().oracle_rspmm_backward(_p(csr.row_ptr), _p(csr.col), _p(csr.rel), _p(csr.w), _p(relation), _p(x),
                                     _p(out), _p(grad), _p(d_rel), _p(d_x), _p(d_w), _i64(csr.n_rows),
                                     _i64(csr.n_cols), _i64(csr.n_edges), _i64(csr.n_rel), _i64(F), SUM_OPS[sum],
                                     MUL_OPS[mul], _i64(piece))
    if rc:
        raise RuntimeError("oracle_rspmm_backward failed: %d" % rc)
    if dense_relation:
        assert csr.n_rel == 4 and sum == "add" and mul == "mul" and bool(np.all(csr.w == 1.0))
        rc = lib().oracle_rspmm_drelation_dense(_p(csr.row_ptr), _p(csr.col), _p(csr.rel), _p(x), _p(grad), _p(d_rel),
                                                _i64(csr.n_rows), _i64(F))
        if rc:
            raise RuntimeError("oracle_rspmm_drelation_dense failed: %d" % rc)
    return (d_rel, d_x, d_w) if need_weight_grad else (d_rel, d_x)


def rspmm_materialised(dst, src, rel, w, relation, x, n_rows, sum="add", mul="mul", dtype=np.float32):
    """``ultra/layer.py:232-296`` without the boundary rows: O(E*F) messages then a scatter."""
    relation = np.asarray(relation, dtype=dtype)
    x = np.asarray(x, dtype=dtype)
    dst = np.asarray(dst, dtype=np.int64)
    w = np.ones(dst.shape[0], dtype=dtype) if w is None else np.asarray(w, dtype=dtype)
    node_input = x[np.asarray(src, dtype=np.int64)]          # layer.py:249
    edge_input = relation[np.asarray(rel, dtype=np.int64)]   # layer.py:250
    message = edge_input + node_input if mul == "add" else edge_input * node_input  # :252-255
    message = message * w[:, None]                            # :275
    F = x.shape[1]
    if sum == "add":
        out = np.zeros((n_rows, F), dtype=dtype)
        np.add.at(out, dst, message)                          # scatter_add :276
    elif sum == "max":
        out = np.full((n_rows, F), np.finfo(np.float32).min, dtype=dtype)    # numeric_limits::lowest()
        np.maximum.at(out, dst, message)                      # scatter_max :280
    elif sum == "min":
        out = np.full((n_rows, F), np.finfo(np.float32).max, dtype=dtype)    # numeric_limits::max()
        np.minimum.at(out, dst, message)                      # scatter_min :285
    else:
        raise ValueError(sum)
    return out


def rspmm_python(dst, src, rel, w, relation, x, n_rows, sum="add", mul="mul"):
    """Plain loops in Python floats (fp64); tiny graphs only."""
    F = len(x[0])
    fmax = float(np.finfo(np.float32).max)
    ident = {"add": 0.0, "min": fmax, "max": -fmax}[sum]
    out = [[ident] * F for _ in range(n_rows)]
    for k in range(len(dst)):
        wk = 1.0 if w is None else float(w[k])
        for f in range(F):
            r, xv = float(relation[rel[k]][f]), float(x[src[k]][f])
            y = wk * (r * xv if mul == "mul" else r + xv)
            cur = out[dst[k]][f]
            out[dst[k]][f] = cur + y if sum == "add" else (min(cur, y) if sum == "min" else max(cur, y))
    return out


def filtered_rank(pred, mask, target):
    """``ultra/task.py:307-315``: ``sum((pos_pred <= pred) & mask, -1) + 1``."""
    pred = np.ascontiguousarray(pred, dtype=np.float32)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    target = np.ascontiguousarray(target, dtype=np.int64)
    nq, nc = pred.shape
    rank = np.empty(nq, dtype=np.int64)
    rc = lib().oracle_filtered_rank(_p(pred), _p(mask), _p(target), _p(rank), _i64(nq), _i64(nc))
    if rc:
        raise RuntimeError("oracle_filtered_rank failed")
    return rank


def combine_forward(input, update, weight, bias, gamma=None, beta=None, eps=1e-5, relu=True, shortcut=False):
    """``ultra/layer.py:386-392`` + the shortcut of ``ultra/model.py:126-127`` in the kernel's summation order."""
    input = np.ascontiguousarray(input, dtype=np.float32).reshape(-1, 64)
    update = np.ascontiguousarray(update, dtype=np.float32).reshape(-1, 64)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    bias = np.ascontiguousarray(bias, dtype=np.float32)
    assert weight.shape == (64, 128) and bias.shape == (64,) and input.shape == update.shape
    gamma = None if gamma is None else np.ascontiguousarray(gamma, dtype=np.float32)
    beta = None if beta is None else np.ascontiguousarray(beta, dtype=np.float32)
    out = np.empty_like(input)
    rc = lib().oracle_combine_forward(_p(input), _p(update), _p(weight), _p(bias), _p(gamma), _p(beta),
                                      ctypes.c_float(eps), int(bool(relu)), int(bool(shortcut)), _p(out),
                                      _i64(input.shape[0]))
    if rc:
        raise RuntimeError("oracle_combine_forward failed")
    return out


def linear_forward(input, weight, bias, relu=False):
    """The small dense layers (``ultra/layer.py:228,318-319``, ``ultra/model.py:53,193``) in the kernels' documented order."""
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    bias = np.ascontiguousarray(bias, dtype=np.float32)
    out_dim, in_dim = weight.shape
    x = np.ascontiguousarray(input, dtype=np.float32)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, in_dim)
    out = np.empty((x2.shape[0], out_dim), dtype=np.float32)
    rc = lib().oracle_linear_forward(_p(x2), _p(weight), _p(bias), _p(out), _i64(x2.shape[0]), _i64(in_dim),
                                     _i64(out_dim), int(bool(relu)))
    if rc:
        raise RuntimeError("oracle_linear_forward failed")
    return out.reshape(lead + (out_dim,))


def score_head_forward(hidden, query, w1, b1, w2, b2):
    """Score head of full-batch evaluation (``ultra/model.py:134-138,177-193``: ``mlp(cat[hidden, query])`` for every
    entity as candidate) in the kernels' documented order: ``hidden`` ``(N, B, 64)``, ``query`` ``(B, 64)`` -> ``(B, N)``.
    The query half of the 128-wide first layer is summed once per query (``c = b1 + W1[:, 64:] . query``, a linear layer
    over the B queries) and is the starting value of the hidden half's chain (``oracle_linear_forward_grouped``)."""
    hidden = np.ascontiguousarray(hidden, dtype=np.float32)
    query = np.ascontiguousarray(query, dtype=np.float32)
    w1 = np.ascontiguousarray(w1, dtype=np.float32)
    n_node, batch, dim = hidden.shape
    assert dim == 64 and query.shape == (batch, 64) and w1.shape == (128, 128)
    c = np.empty((batch, 128), dtype=np.float32)
    b1 = np.ascontiguousarray(b1, dtype=np.float32)
    w_query = w1[:, 64:]                                            # a view: rows 128 floats apart
    rc = lib().oracle_linear_forward_grouped(ctypes.c_void_p(query.ctypes.data), ctypes.c_void_p(w_query.ctypes.data),
                                             _i64(128), _p(b1), _i64(max(batch, 1)), _p(c), _i64(batch), _i64(64),
                                             _i64(128), 0)
    rows = np.ascontiguousarray(hidden.transpose(1, 0, 2)).reshape(batch * n_node, 64)          # (B, N, 64): query-major
    h = np.empty((batch * n_node, 128), dtype=np.float32)
    rc |= lib().oracle_linear_forward_grouped(_p(rows), _p(w1), _i64(128), _p(c), _i64(max(n_node, 1)), _p(h),
                                              _i64(batch * n_node), _i64(64), _i64(128), 1)
    if rc:
        raise RuntimeError("oracle_linear_forward_grouped failed")
    return linear_forward(h, np.asarray(w2, dtype=np.float32).reshape(1, 128), b2).reshape(batch, n_node)

"""RelCSR: the sorted, coalesced edge lists and chunk schedules the HIP kernels walk.

This is the host side of what torchdrug does inside every ``generalized_rspmm`` call -- ``sparse.coalesce()``
followed by ``coo2csr`` (call sites ``/root/reference/ultra/layer.py:127,328``) -- done
ONCE per graph and cached, plus the two extra orderings the atomic-free backward needs.

Three reduction plans over the same coalesced edge set (``include/ultra_rspmm.h``, ``ultra_segments``):

=============  ==================  ==========================  ===============================
plan           target row          sorted by                   used for
=============  ==================  ==========================  ===============================
``fwd``        destination node    (dst, src, rel)             forward  (rows of ``out``)
``by_src``     source node         (src, dst, rel)             ``d_input``
``by_rel``     relation            (rel, dst, src)             ``d_relation``
=============  ==================  ==========================  ===============================

A plan's *chunks* are what one wavefront processes: a run of whole rows whose first edges fall in the same
``chunk_edges``-sized block of the edge list (and in the same ``chunk_rows``-sized block of rows), or one piece
of ``piece_len`` consecutive edges of a row with more than ``piece_len`` edges.  Pieces are summed separately
and added in piece order, which is the summation order ``oracle/rspmm_oracle.c`` reproduces with ``piece > 0``.

torch is used here for device memory and sorting only (plumbing).
"""
import ctypes
import os

import torch

from . import _lib

CHUNK_ROWS = 64      # at most this many rows per chunk (bounds the work of runs of empty rows)


def auto_sizes(n_edges, wide=False, n_rows=None):
    """(chunk_edges, piece_len) for a graph with ``n_edges`` coalesced edges.  Bigger chunks mean fewer pieces, less
    fix-up traffic and longer runs of the branch-free batch path, but every 16-lane group of the quad kernel (four
    chunks per wave, 2 048 groups share a column tile on an MI355X) needs a few chunks to stay busy.  Measured best
    (tools/kbench.py --chunk --piece, forward / backward): S-codexs (66 k edges) and S-wn18rr (174 k) 32 / 128,
    S-fb15k237 (544 k) 128 / 512 (round 3, with a label's column tiles side by side: evaluation batch 2.40 -> 2.38 ms with
    128-edge chunks, -> 2.36 ms with 512-edge pieces; 1 024-edge pieces: 2.345 ms, but the fine-tuning step 6.88 -> 7.28 ms
    through the relation-major d_relation plan, whose rows are long; fine-tuning steps unchanged at 512).  Graphs of few,
    very long rows (relation graphs) keep 256: 512 there cost the evaluation batch 80 us.  ``wide``: node ids do not fit the packed word (one chunk
    per wave, kernel variants 2 / 3): 128 / 512.  ``ULTRA_CHUNK_EDGES`` overrides the chunk size (experiments)."""
    if wide:
        return 128, 512
    import os
    forced = os.environ.get("ULTRA_CHUNK_EDGES")            # experiments (tools/kbench.py has --chunk for single kernels)
    piece = os.environ.get("ULTRA_PIECE_LEN")
    if n_edges >= 300_000:
        # graphs of few, very long rows (a relation graph: 474 rows of 1 896 edges) keep short pieces -- they ARE the work
        # items; entity graphs split only their hubs and take long ones (ULTRA_BIG_PIECE, experiments)
        dense_rows = n_rows is not None and n_edges >= 512 * max(int(n_rows), 1)
        big = int(os.environ.get("ULTRA_BIG_PIECE", "512"))
        return (int(forced) if forced else 128), (int(piece) if piece else (256 if dense_rows else big))
    return (int(forced) if forced else 32), (int(piece) if piece else 128)


CHUNK_EDGES, PIECE_LEN = auto_sizes(0)    # the sizes small graphs get (kept as names for explicit callers / tests)
PACK_SLACK = 16      # readable words after the last edge of `packed` / `weight` (whole-batch loads)
# per-step edge removal on unit-weight graphs: marked word copies beside the zero weights (ultra_edge_removal_marks); 0: weights only
DEAD_EDGE_WORDS = os.environ.get("ULTRA_DEAD_EDGE_WORDS", "1") != "0"
LDS_TABLE_BYTES = 156 * 1024 - 16    # LDS the kernels give to the relation tile + hot-row cache (csrc: kMaxLdsBytes)
HOT_MAX = 512        # at most this many cached rows
HOT_MIN_COVERAGE = 0.2   # build a hot-row cache only if it serves at least this fraction of the gathers
_INT32_MAX = 2 ** 31 - 1

# Dense form of a plan (csrc/relgraph_dense.hip; include/ultra_rspmm.h, ABI 7): unit-weight adjacencies over exactly 4 relation
# types -- the relation graphs of /root/reference/ultra/rel_model.py:99-143 -- whose edges fill at least this fraction of the
# n_dst x n_src x 4 slots run their sum aggregations as a product with a 0/1 matrix on the exact-f32 matrix cores: the
# reference's sequential order for every row.  The product costs the same whatever the density, the edge walk is linear in the
# edges: the crossover is measured in DESIGN.md (tools/dense_crossover.py).  ULTRA_DENSE_RELGRAPH=0 switches the form off.
DENSE_RELGRAPH = os.environ.get("ULTRA_DENSE_RELGRAPH", "1") != "0"
DENSE_MIN_DENSITY = float(os.environ.get("ULTRA_DENSE_MIN_DENSITY", "0.08"))
DENSE_TYPES = 4


def dense_bytes(n_rows, n_cols, kind):
    """Size of a plan's dense matrix -- ``ultra_relcsr_dense_bytes`` restated (the decision must not depend on the device:
    CPU-side RelCSR objects tell the tests' oracle which summation order the kernels use).  0: shape not supported."""
    if n_rows <= 0 or n_cols <= 0 or n_rows > (1 << 20) or n_cols > (1 << 20):
        return 0
    n_vt = (n_rows + 15) // 16
    round_ = 16 if kind == 0 else 32
    size = n_vt * 64 * ((n_cols + round_ - 1) // round_ * round_) + (1024 if kind == 0 else 512)       # + slack the kernels may read
    return size if size < (1 << 31) else 0


class Segments:
    """One reduction plan; owns the device tensors and the ``ultra_segments`` struct pointing at them."""

    def __init__(self, row, node_a, node_b, rel, weight, n_rows, chunk_edges=CHUNK_EDGES, chunk_rows=CHUNK_ROWS,
                 piece_len=PIECE_LEN, balance=True, wide_ids=False, lds_rel_rows=0, n_gather_rows=0, hot_cache=False,
                 n_node_a=None, n_rel_table=None, builder=None):
        """``wide_ids`` forces the big-graph word layout (node ids outside the packed word) even when they would fit
        -- used by tests to exercise that kernel variant on small graphs.  ``n_node_a`` / ``n_rel_table``: declared
        ranges of the id fields (bit widths of the packed word).  ``builder``: "native" (libultra_rspmm, rocPRIM; the
        default for device tensors) or "torch" (the same construction in torch ops; CPU tensors, cross-check)."""
        n_edges = int(row.shape[0])
        if n_rows > _INT32_MAX or n_edges > _INT32_MAX:
            raise ValueError("graph too large for int32 indices: %d rows, %d edges" % (n_rows, n_edges))
        self.n_rows, self.n_edges, self.piece_len = int(n_rows), n_edges, int(piece_len)
        dev = row.device
        i32 = torch.int32
        self.row = row.to(i32).contiguous()
        self.node_a = node_a.to(i32).contiguous()
        self.node_b = None if node_b is None else node_b.to(i32).contiguous()
        self.rel = rel.to(i32).contiguous()
        self.weight = None if weight is None else torch.cat(
            [weight.to(torch.float32), torch.ones(PACK_SLACK, dtype=torch.float32, device=dev)]).contiguous()

        n_rel = int(n_rel_table) if n_rel_table is not None else (int(rel.max()) + 1 if n_edges else 1)
        n_a = int(n_node_a) if n_node_a is not None else (int(node_a.max()) + 1 if n_edges else 1)
        self.hot_nodes, self.n_hot = None, 0
        self.packed, self.packed_src_shift = None, 0
        self.row_ptr = None
        self.dense, self.dense_rows, self.dense_cols = None, 0, 0
        if builder is None:
            builder = os.environ.get("ULTRA_RELCSR_BUILDER") or ("native" if row.is_cuda else "torch")
        self.builder = "torch" if (hot_cache or not row.is_cuda) else builder
        if self.builder == "native":
            self._build_native(n_a, n_rel, chunk_edges, chunk_rows, piece_len, balance, wide_ids)
            self._build_row_ptr()
            self.struct = _lib.UltraSegments()
            self._refresh_struct()
            return

        deg = torch.bincount(row, minlength=n_rows) if n_edges else torch.zeros(n_rows, dtype=torch.long, device=dev)
        row_ptr = torch.zeros(n_rows + 1, dtype=torch.long, device=dev)
        torch.cumsum(deg, 0, out=row_ptr[1:])
        chunks, long_rows, n_pieces, row_begin = _schedule(row_ptr, deg, chunk_edges, chunk_rows, piece_len, balance)
        self.chunks = chunks.to(i32).contiguous()
        # packed edge words for the fast path: row delta | relation << 8 | node_a << src_shift   (ultra_rspmm.h)
        # d_relation plan (node_b given): the row IS the relation, no relation field; the word holds node_a only
        bits_rel = 0 if node_b is not None else max((n_rel - 1).bit_length(), 1)
        # hot-row cache (forward / d_input plans of KG-sized graphs): the most frequently gathered nodes, as many as fit
        # in LDS next to the relation tile -- unless the whole gathered matrix fits there anyway (relation graphs).
        # OFF by default: measured SLOWER on MI355X (S-fb15k237 forward 193 -> 250 us, S-wn18rr 122 -> 186 us): the
        # per-edge scalar branch breaks the 8-deep load batches and an LDS read is no cheaper than an L1-hit gather
        # in this instruction-bound loop; kept (tested, bit-identical) as a tuning switch.
        node_field = node_a
        room = (LDS_TABLE_BYTES - 256 * int(lds_rel_rows)) // 256
        whole_fits = 0 < int(n_gather_rows) <= room
        if hot_cache and n_edges and node_b is None and not wide_ids and not whole_fits and room >= 16:
            counts = torch.bincount(node_a, minlength=n_a)
            k = int(min(HOT_MAX, room, int((counts > 0).sum())))
            top = torch.topk(counts, k)
            if k >= 16 and float(top.values.sum()) >= HOT_MIN_COVERAGE * n_edges and \
                    8 + bits_rel + (n_a + k - 1).bit_length() <= 32:
                self.n_hot = k
                self.hot_nodes = top.indices.to(i32).contiguous()
                slot = torch.full((n_a,), -1, dtype=torch.long, device=dev)
                slot[top.indices] = torch.arange(k, device=dev)
                node_slot = slot[node_a]
                node_field = torch.where(node_slot >= 0, node_slot, node_a + k)
                n_a = n_a + k
        if n_edges and chunk_rows <= 256 and 8 + bits_rel + max((n_a - 1).bit_length(), 1) <= 32 and not (
                wide_ids and node_b is None):
            delta = row - row_begin[row]
            word = delta | ((rel << 8) if bits_rel else 0) | (node_field << (8 + bits_rel))
            word = torch.where(word >= 2 ** 31, word - 2 ** 32, word).to(i32)
            # PACK_SLACK zero words after the last edge: the kernel always loads whole batches of 8 words
            self.packed = torch.cat([word, torch.zeros(PACK_SLACK, dtype=i32, device=dev)]).contiguous()
            self.packed_src_shift = 8 + bits_rel
        elif n_edges and chunk_rows <= 256 and node_b is None and n_rel <= 2 ** 24:
            # big graphs: the word keeps row delta | relation << 8; the kernel reads the node id from node_a
            # (packed_src_shift = 32 says so)
            word = (row - row_begin[row]) | (rel << 8)
            word = torch.where(word >= 2 ** 31, word - 2 ** 32, word).to(i32)
            self.packed = torch.cat([word, torch.zeros(PACK_SLACK, dtype=i32, device=dev)]).contiguous()
            self.packed_src_shift = 32
            self.node_a = torch.cat([self.node_a, torch.zeros(PACK_SLACK, dtype=i32, device=dev)]).contiguous()
        self.long_rows = long_rows.to(i32).contiguous()
        self.n_pieces = int(n_pieces)
        self._build_row_ptr()

        self.struct = _lib.UltraSegments()
        self._refresh_struct()

    def _build_native(self, n_a, n_rel, chunk_edges, chunk_rows, piece_len, balance, wide_ids):
        """Chunk schedule + packed words through ``ultra_relcsr_plan`` (csrc/relcsr_build.hip)."""
        lib = _lib.load()
        dev, i32 = self.row.device, torch.int32
        E, R = self.n_edges, self.n_rows
        cap_chunks = R + E // piece_len + 2
        cap_long = E // piece_len + 1
        chunks = torch.empty(cap_chunks, 4, dtype=i32, device=dev)
        long_rows = torch.empty(cap_long, 3, dtype=i32, device=dev)
        packed = torch.empty(E + PACK_SLACK, dtype=i32, device=dev) if E else None
        temp = torch.empty(int(lib.ultra_relcsr_plan_temp_bytes(E, R, piece_len)), dtype=torch.uint8, device=dev)
        counts = (ctypes.c_int64 * 4)()
        with torch.cuda.device(dev):
            _lib.check(lib.ultra_relcsr_plan(
                self.row.data_ptr(), self.node_a.data_ptr(), self.rel.data_ptr(), E, R, n_a, n_rel,
                int(self.node_b is not None), int(bool(wide_ids)), int(bool(balance)), chunk_edges, chunk_rows, piece_len,
                chunks.data_ptr(), cap_chunks, long_rows.data_ptr(), cap_long,
                packed.data_ptr() if packed is not None else None, PACK_SLACK, counts, temp.data_ptr(), temp.numel(),
                torch.cuda.current_stream().cuda_stream))
        n_chunks, n_long, n_pieces, shift = (int(c) for c in counts)
        self.chunks = chunks[:n_chunks].clone()
        self.long_rows = long_rows[:n_long].clone()
        self.n_pieces = n_pieces
        if shift:
            self.packed, self.packed_src_shift = packed, shift
            if shift == 32:     # node ids are read from node_a in whole batches: same slack as the packed words
                self.node_a = torch.cat([self.node_a, torch.zeros(PACK_SLACK, dtype=i32, device=dev)]).contiguous()

    def _build_row_ptr(self):
        """Row pointers for the plans of big graphs (node ids outside the packed word): with them and no split rows the
        library reduces one row per 16-lane group (csrc/rowgroup.inc) instead of walking chunks."""
        if self.packed_src_shift == 32 and self.n_edges:
            rows = torch.arange(self.n_rows + 1, dtype=torch.int32, device=self.row.device)
            self.row_ptr = torch.searchsorted(self.row, rows).to(torch.int32).contiguous()

    def _refresh_struct(self):
        s = self.struct
        s.n_rows, s.n_edges = self.n_rows, self.n_edges
        s.row = self.row.data_ptr()
        s.node_a = self.node_a.data_ptr()
        s.node_b = self.node_b.data_ptr() if self.node_b is not None else None
        s.rel = self.rel.data_ptr()
        s.weight = self.weight.data_ptr() if self.weight is not None else None
        s.n_chunks = int(self.chunks.shape[0])
        s.chunks = self.chunks.data_ptr()
        s.n_long_rows = int(self.long_rows.shape[0])
        s.long_rows = self.long_rows.data_ptr()
        s.n_pieces = self.n_pieces
        s.piece_len = self.piece_len
        s.packed = self.packed.data_ptr() if self.packed is not None else None
        s.packed_src_shift = self.packed_src_shift
        s.n_hot = self.n_hot
        s.hot_nodes = self.hot_nodes.data_ptr() if self.hot_nodes is not None else None
        s.row_ptr = self.row_ptr.data_ptr() if self.row_ptr is not None else None
        s.dense = self.dense.data_ptr() if self.dense is not None else None
        s.dense_rows, s.dense_cols = self.dense_rows, self.dense_cols
        dead = getattr(self, "packed_dead", None)
        s.packed_dead = dead.data_ptr() if dead is not None else None

    def attach_dense(self, n_rows, n_cols, kind):
        """Build the plan's 0/1 matrix natively (``ultra_relcsr_dense``) and hang it on the struct.  ``kind`` 0: rows of the
        plan are nodes (forward / d_input); 1: rows are the 4 relation types (d_relation), ``n_rows`` = destination nodes."""
        lib = _lib.load()
        size = int(lib.ultra_relcsr_dense_bytes(n_rows, n_cols, kind))
        if size == 0 or self.weight is not None or not self.row.is_cuda:
            return False
        dense = torch.empty(size // 4, dtype=torch.int32, device=self.row.device)
        with torch.cuda.device(self.row.device):
            _lib.check(lib.ultra_relcsr_dense(self.pointer, n_rows, n_cols, kind, dense.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream))
        self.dense, self.dense_rows, self.dense_cols = dense, int(n_rows), int(n_cols)
        self._refresh_struct()
        return True

    @property
    def workspace_rows(self):
        """Rows of F floats a call over this plan needs as scratch (``ultra_rspmm_workspace_bytes``): the piece sums of the edge
        walk; for a d_relation plan in its dense form also one tile sum per 16 destination nodes and relation type."""
        rows = self.n_pieces
        if self.dense is not None and self.node_b is not None and self.n_rows == 4:
            rows = max(rows, 4 * ((self.dense_rows + 15) // 16))
        return rows

    @property
    def pointer(self):
        return ctypes.byref(self.struct)

    @property
    def plan_tensor(self):
        """The ``ultra_segments`` struct as a CPU uint8 tensor (shares the struct's memory): the form in which a plan
        crosses the PyTorch dispatcher to ``torch.ops.ultra_mi.rspmm_plan_*`` (csrc/torch_ext.cpp)."""
        if getattr(self, "_plan_tensor", None) is None or self._plan_tensor_of is not self.struct:
            self._plan_tensor = torch.frombuffer(self.struct, dtype=torch.uint8)
            self._plan_tensor_of = self.struct
        return self._plan_tensor

    def reweighted(self, weight):
        """Shallow copy that shares every index array / schedule of this plan and carries other edge weights
        (``weight``: fp32 [n_edges] in THIS plan's edge order)."""
        import copy
        other = copy.copy(self)
        other.packed_dead = None
        other.weight = torch.cat([weight.to(torch.float32),
                                  torch.ones(PACK_SLACK, dtype=torch.float32, device=weight.device)]).contiguous()
        other.struct = _lib.UltraSegments()
        other._refresh_struct()
        return other

    def with_weight_buffer(self, buffer, packed_dead=None):
        """As :meth:`reweighted`, for a weight array that already carries its ``PACK_SLACK`` trailing ones.  ``packed_dead``: a copy
        of ``packed`` with bit 31 set where ``buffer`` is 0, for weights that are otherwise all 1 (``ultra_edge_removal_marks``)."""
        import copy
        other = copy.copy(self)
        other.weight = buffer
        other.packed_dead = packed_dead
        other.struct = _lib.UltraSegments()
        other._refresh_struct()
        return other

    @property
    def device(self):
        return self.row.device


def _schedule(row_ptr, deg, chunk_edges, chunk_rows, piece_len, balance):
    """Vectorised chunk schedule.  Returns (chunks [n,4], long_rows [m,3], n_pieces, row_begin [n_rows]) where
    row_begin[r] is the first row of the chunk that owns row r (r itself for split rows)."""
    dev = row_ptr.device
    n_rows = deg.shape[0]
    if n_rows == 0:
        return (torch.zeros(0, 4, dtype=torch.long, device=dev), torch.zeros(0, 3, dtype=torch.long, device=dev), 0,
                torch.zeros(0, dtype=torch.long, device=dev))
    rows = torch.arange(n_rows, device=dev)
    is_long = deg > piece_len
    blk = torch.div(row_ptr[:-1], chunk_edges, rounding_mode="floor")
    rblk = torch.div(rows, chunk_rows, rounding_mode="floor")
    start = torch.ones(n_rows, dtype=torch.bool, device=dev)
    start[1:] = (blk[1:] != blk[:-1]) | (rblk[1:] != rblk[:-1]) | is_long[1:] | is_long[:-1]
    g_first = torch.nonzero(start).flatten()
    row_begin = g_first[torch.cumsum(start.long(), 0) - 1]
    g_last = torch.cat([g_first[1:], torch.tensor([n_rows], device=dev)])
    g_long = is_long[g_first]

    nf, nl = g_first[~g_long], g_last[~g_long]
    normal = torch.stack([row_ptr[nf], row_ptr[nl], nf, nl], dim=1)

    lrow = g_first[g_long]
    if lrow.numel():
        ldeg = deg[lrow]
        n_p = torch.div(ldeg + piece_len - 1, piece_len, rounding_mode="floor")
        first_slot = torch.cumsum(n_p, 0) - n_p
        n_pieces = int(n_p.sum().item())
        owner = torch.repeat_interleave(torch.arange(lrow.numel(), device=dev), n_p)
        k = torch.arange(n_pieces, device=dev) - first_slot[owner]
        e_begin = row_ptr[lrow][owner] + k * piece_len
        e_end = torch.minimum(e_begin + piece_len, row_ptr[lrow + 1][owner])
        slot = first_slot[owner] + k
        pieces = torch.stack([e_begin, e_end, lrow[owner], -(slot + 1)], dim=1)
        long_rows = torch.stack([lrow, first_slot, n_p], dim=1)
        chunks = torch.cat([pieces, normal], dim=0)
    else:
        n_pieces = 0
        long_rows = torch.zeros(0, 3, dtype=torch.long, device=dev)
        chunks = normal
    if balance and chunks.shape[0] > 1:
        # heaviest first: the kernel deals chunks round-robin to wavefronts, so every wavefront gets a similar mix
        cost = (chunks[:, 1] - chunks[:, 0]) * 4 + torch.where(chunks[:, 3] < 0, torch.ones_like(chunks[:, 3]),
                                                                chunks[:, 3] - chunks[:, 2])
        order = torch.sort(cost, descending=True, stable=True).indices
        chunks = chunks[order]
    return chunks, long_rows, n_pieces, row_begin


def _sum_duplicates(weight, first, count):
    """Sequential (input-order) fp32 sum of each run of duplicates; deterministic, unlike index_add_."""
    acc = weight[first].clone()
    max_count = int(count.max().item()) if count.numel() else 1
    for j in range(1, max_count):
        has = count > j
        idx = first[has] + j
        acc[has] = acc[has] + weight[idx]
    return acc


class RelCSR:
    """Coalesced relational adjacency of shape ``(n_dst, n_src, n_rel)`` plus its reduction plans."""

    def __init__(self, dst, src, rel, weight, n_dst, n_src, n_rel, chunk_edges=None, chunk_rows=CHUNK_ROWS,
                 piece_len=None, balance=True, wide_ids=False, hot_cache=False, builder=None):
        """``dst/src/rel``: int64 tensors [E] (any order, duplicates allowed); ``weight``: fp32 [E] or None (ones)."""
        dev = dst.device
        dst, src, rel = dst.long(), src.long(), rel.long()
        n_dst, n_src, n_rel = int(n_dst), int(n_src), int(n_rel)
        if dst.numel():
            lo = min(int(dst.min()), int(src.min()), int(rel.min()))
            if lo < 0 or int(dst.max()) >= n_dst or int(src.max()) >= n_src or int(rel.max()) >= n_rel:
                raise ValueError("edge index out of range for adjacency (%d, %d, %d)" % (n_dst, n_src, n_rel))
        if float(n_dst) * float(n_src) * float(max(n_rel, 1)) >= 2.0 ** 62:
            raise ValueError("adjacency too large for a 64-bit sort key")
        self.shape = (n_dst, n_src, n_rel)
        self._requested = (chunk_edges, piece_len)
        self._opts = dict(chunk_rows=chunk_rows, balance=balance, wide_ids=wide_ids, hot_cache=hot_cache,
                          builder=builder)
        if weight is None:
            weight = torch.ones(dst.shape[0], dtype=torch.float32, device=dev)
        weight = weight.to(torch.float32)

        # coalesce: sort by (dst, src, rel), merge duplicate triples by summing their weights
        if builder is None:
            builder = os.environ.get("ULTRA_RELCSR_BUILDER") or ("native" if dst.is_cuda else "torch")
        if builder == "native" and dst.is_cuda and dst.numel():
            dst, src, rel, w_merged, self.edge_of_input, unit = self._coalesce_native(dst, src, rel, weight)
        elif dst.numel():
            key = (dst * n_src + src) * max(n_rel, 1) + rel
            key, order = torch.sort(key, stable=True)
            uniq, inverse, count = torch.unique_consecutive(key, return_inverse=True, return_counts=True)
            first = torch.cumsum(count, 0) - count
            w_sorted = weight[order]
            w_merged = _sum_duplicates(w_sorted, first, count) if uniq.numel() != key.numel() else w_sorted
            sel = order[first]
            dst, src, rel = dst[sel], src[sel], rel[sel]
            # position of every ORIGINAL edge in the coalesced list (for d_weight of a sparse tensor that requires grad)
            self.edge_of_input = torch.empty_like(order)
            self.edge_of_input[order] = inverse
            unit = bool((w_merged == 1).all().item())
        else:
            w_merged, unit = weight, True
            self.edge_of_input = torch.zeros(0, dtype=torch.long, device=dev)
        self.dst, self.src, self.rel_id = dst, src, rel
        self.unit_weight = unit
        self.weight = w_merged
        self.n_edges = int(dst.shape[0])
        # chunk / piece sizes: explicit, or chosen from the coalesced edge count; ONE pair for all three plans, and
        # `piece_len` is the summation-order parameter the oracle needs (oracle `piece`)
        wide = wide_ids or 8 + max((n_rel - 1).bit_length(), 1) + max((max(n_dst, n_src) - 1).bit_length(), 1) > 32
        auto_chunk, auto_piece = auto_sizes(self.n_edges, wide, n_dst)
        self.chunk_edges = int(self._requested[0] or auto_chunk)
        self.piece_len = int(self._requested[1] or (auto_piece if self._requested[0] is None else 4 * self.chunk_edges))
        self._opts.update(chunk_edges=self.chunk_edges, piece_len=self.piece_len)
        self._fwd = self._by_src = self._by_rel = None
        # dense form (relation graphs): a property of the GRAPH alone -- same answer for a CPU copy of it
        self.dense_form = bool(
            DENSE_RELGRAPH and n_rel == DENSE_TYPES and unit and self.n_edges > 0 and not wide_ids and not hot_cache
            and self.n_edges >= DENSE_MIN_DENSITY * float(n_dst) * float(n_src) * DENSE_TYPES
            and dense_bytes(n_dst, n_src, 0) and dense_bytes(n_src, n_dst, 0) and dense_bytes(n_dst, n_src, 1))

    def kernel_order(self, sum="add", mul="mul", F=64):
        """``(piece, dense_d_relation)``: the summation order the library uses for this adjacency and call -- what the tests'
        oracle needs.  ``piece`` (forward, d_input): 0 = strictly sequential per row (the reference order; the dense form), else
        the plans' piece length.  ``dense_d_relation``: d_relation in the dense form's documented order (include/ultra_rspmm.h)."""
        dense = bool(getattr(self, "dense_form", False) and self.unit_weight and sum == "add" and F % 16 == 0 and F * 4 < (1 << 24))
        return (0 if dense else self.piece_len), (dense and mul == "mul")

    def _coalesce_native(self, dst, src, rel, weight, dims=None):
        """``ultra_relcsr_coalesce`` (csrc/relcsr_build.hip): radix sort of the 64-bit triple key + duplicate merge.
        ``dims``: ranges of the three key columns (default: this adjacency's ``(n_dst, n_src, n_rel)``)."""
        lib = _lib.load()
        dev, n = dst.device, int(dst.shape[0])
        n_dst, n_src, n_rel = self.shape if dims is None else dims
        dst, src, rel = dst.contiguous(), src.contiguous(), rel.contiguous()
        weight = weight.contiguous() if weight is not None else None
        out_idx = torch.empty(3, n, dtype=torch.int32, device=dev)
        out_w = torch.empty(n, dtype=torch.float32, device=dev)
        edge_of_input = torch.empty(n, dtype=torch.long, device=dev)
        temp = torch.empty(int(lib.ultra_relcsr_coalesce_temp_bytes(n)), dtype=torch.uint8, device=dev)
        n_unique, unit = ctypes.c_int64(0), ctypes.c_int(1)
        with torch.cuda.device(dev):
            _lib.check(lib.ultra_relcsr_coalesce(
                dst.data_ptr(), src.data_ptr(), rel.data_ptr(), weight.data_ptr() if weight is not None else None, n, n_dst,
                n_src, n_rel,
                out_idx[0].data_ptr(), out_idx[1].data_ptr(), out_idx[2].data_ptr(), out_w.data_ptr(),
                edge_of_input.data_ptr(), ctypes.byref(n_unique), ctypes.byref(unit), temp.data_ptr(), temp.numel(),
                torch.cuda.current_stream().cuda_stream))
        m = int(n_unique.value)
        ids = out_idx[:, :m].long()        # int64: the by_src / by_rel sort keys are products of these
        return ids[0], ids[1], ids[2], out_w[:m].clone(), edge_of_input, bool(unit.value)

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_sparse(cls, sparse, **opts):
        """From the 3-D sparse COO ``(N_dst, N_src, R)`` the reference passes (``layer.py:127,328``)."""
        if not sparse.is_sparse or sparse.dim() != 3:
            raise ValueError("expected a 3-D sparse COO tensor (N_dst, N_src, R), got %s" % (tuple(sparse.shape),))
        idx = sparse._indices()
        return cls(idx[0], idx[1], idx[2], sparse._values(), *sparse.shape, **opts)

    @classmethod
    def from_edge_list(cls, edge_list, edge_weight, num_node, num_relation, **opts):
        """From a torchdrug-style ``edge_list`` of (node_in, node_out, relation) rows: ``adjacency.transpose(0, 1)``
        as the layers use it (``layer.py:56,127``), i.e. destination = node_out, source = node_in."""
        return cls(edge_list[:, 1], edge_list[:, 0], edge_list[:, 2], edge_weight, num_node, num_node, num_relation,
                   **opts)

    # ------------------------------------------------------------------ plans
    @property
    def device(self):
        return self.dst.device

    def _w(self, order=None):
        if self.unit_weight:
            return None
        return self.weight if order is None else self.weight[order]

    @property
    def fwd(self):
        if self._fwd is None and getattr(self, "_base", None) is not None:
            self._fwd = self._base.fwd.reweighted(self.weight)
        if self._fwd is None:
            self._fwd = Segments(self.dst, self.src, None, self.rel_id, self._w(), self.shape[0],
                                 lds_rel_rows=self.shape[2], n_gather_rows=self.shape[1], n_node_a=self.shape[1],
                                 n_rel_table=self.shape[2], **self._opts)
            if self.dense_form:
                self._fwd.attach_dense(self.shape[0], self.shape[1], 0)
        return self._fwd

    def _reorder(self, first, second, third, dims):
        """The coalesced edges sorted by ``(first, second, third)``: returns the three sorted columns and ``order``
        (sorted position -> index in forward order).  On the device this is the library's radix sort over the
        significant key bits (the edges are already distinct, so nothing merges); torch.sort otherwise."""
        if self._opts.get("builder") != "torch" and first.is_cuda and first.numel() and \
                (self._opts.get("builder") or os.environ.get("ULTRA_RELCSR_BUILDER") or "native") == "native":
            a, b, c, _w, position, _unit = self._coalesce_native(first, second, third, None, dims)
            order = torch.empty_like(position)
            order[position] = torch.arange(position.numel(), device=position.device)
            return a, b, c, order
        key = (first * dims[1] + second) * max(dims[2], 1) + third
        order = torch.sort(key, stable=True).indices
        return first[order], second[order], third[order], order

    @property
    def by_src(self):
        if self._by_src is None and getattr(self, "_base", None) is not None:
            base = self._base.by_src
            self._by_src = base.reweighted(self.weight[self._base._by_src_order])
        if self._by_src is None:
            n_dst, n_src, n_rel = self.shape
            src, dst, rel, order = self._reorder(self.src, self.dst, self.rel_id, (n_src, n_dst, n_rel))
            self._by_src_order = order
            self._by_src = Segments(src, dst, None, rel, self._w(order), n_src,
                                    lds_rel_rows=n_rel, n_gather_rows=n_dst, n_node_a=n_dst, n_rel_table=n_rel,
                                    **self._opts)
            if self.dense_form:
                self._by_src.attach_dense(n_src, n_dst, 0)
        return self._by_src

    @property
    def by_rel(self):
        if self._by_rel is None and getattr(self, "_base", None) is not None:
            base = self._base.by_rel
            self._by_rel = base.reweighted(self.weight[self._base._by_rel_order])
        if self._by_rel is None:
            n_dst, n_src, n_rel = self.shape
            rel, dst, src, order = self._reorder(self.rel_id, self.dst, self.src, (n_rel, n_dst, n_src))
            self._by_rel_order = order
            self._by_rel = Segments(rel, src, dst, rel, self._w(order), n_rel, n_node_a=n_src, n_rel_table=n_rel,
                                    **self._opts)
            if self.dense_form:
                self._by_rel.attach_dense(n_dst, n_src, 1)
        return self._by_rel

    @property
    def frontier_index(self):
        """``(src_ptr, fwd_rank)`` for :func:`functional.rspmm_frontier` (first Bellman-Ford layer): the first out-edge
        of every source node in ``by_src`` order, and for each of those edges its position inside its destination row
        in FORWARD order (``rank // piece_len`` = the piece of a split row the edge is summed in).  int32, built once
        per graph; a reweighted RelCSR shares its base's."""
        base = getattr(self, "_base", None)
        if base is not None:
            return base._frontier_index_for(self.dense_form)
        return self._frontier_index_for(self.dense_form)

    def _frontier_index_for(self, dense_form):
        """Where the plans carry their dense form the full kernels never split a row (the reference order for every row), so
        the frontier kernel must not either: every edge gets rank 0 -- one piece per row, summed in (destination, relation)
        order.  (The first FB15k237Inductive-v1-shaped run caught the mix: a source's four parallel edges straddling a
        256-edge piece boundary of a row the dense kernels sum sequentially, scores off by one ulp.)"""
        if getattr(self, "_frontier_index", None) is None:
            _ = self.by_src                                        # builds the (src, dst, rel) order
            n_dst, n_src, _n_rel = self.shape
            dev = self.device
            fwd_ptr = torch.zeros(n_dst + 1, dtype=torch.long, device=dev)
            torch.cumsum(torch.bincount(self.dst, minlength=n_dst), 0, out=fwd_ptr[1:])
            rank_fwd = torch.arange(self.n_edges, device=dev) - fwd_ptr[self.dst]
            src_ptr = torch.zeros(n_src + 1, dtype=torch.long, device=dev)
            torch.cumsum(torch.bincount(self.src, minlength=n_src), 0, out=src_ptr[1:])
            self._frontier_index = (src_ptr.to(torch.int32).contiguous(),
                                    rank_fwd[self._by_src_order].to(torch.int32).contiguous())
        if not dense_form:
            return self._frontier_index
        if getattr(self, "_frontier_index_dense", None) is None:
            self._frontier_index_dense = (self._frontier_index[0], torch.zeros_like(self._frontier_index[1]))
        return self._frontier_index_dense

    @property
    def boundary_relation_index(self):
        """``(items, src_ptr, src_relpos)`` for :func:`functional.rspmm_drelation_boundary` (``d_relation`` of the first
        Bellman-Ford layer from the boundary nodes' out-edges): ``items`` int32 ``(n_pieces + n_unsplit_rows, 3)`` -- one
        ``{begin, end, target}`` per piece of a split row of the ``by_rel`` plan (``target = -(slot + 1)``) and per unsplit row
        (``target`` = the relation) --, the first out-edge of every source node, and per source node the ``by_rel`` positions of
        its out-edges, ascending.  Built once per graph (a reweighted RelCSR shares its base's: same plan, other weights)."""
        base = getattr(self, "_base", None)
        if base is not None:
            return base.boundary_relation_index
        if getattr(self, "_boundary_relation_index", None) is None:
            plan = self.by_rel
            n_dst, n_src, n_rel = self.shape
            dev, E = self.device, self.n_edges
            rel_ptr = torch.searchsorted(plan.row.long(), torch.arange(n_rel + 1, device=dev)) if E else \
                torch.zeros(n_rel + 1, dtype=torch.long, device=dev)
            long_rows = plan.long_rows.long()
            split = torch.zeros(n_rel, dtype=torch.bool, device=dev)
            if long_rows.shape[0]:
                lrow, first_slot, n_p = long_rows[:, 0], long_rows[:, 1], long_rows[:, 2]
                split[lrow] = True
                owner = torch.repeat_interleave(torch.arange(lrow.numel(), device=dev), n_p)
                k = torch.arange(int(n_p.sum()), device=dev) - (torch.cumsum(n_p, 0) - n_p)[owner]
                begin = rel_ptr[lrow[owner]] + k * plan.piece_len
                end = torch.minimum(begin + plan.piece_len, rel_ptr[lrow[owner] + 1])
                pieces = torch.stack([begin, end, -(first_slot[owner] + k + 1)], dim=1)
            else:
                pieces = torch.zeros(0, 3, dtype=torch.long, device=dev)
            rows = torch.nonzero(~split).flatten()
            whole = torch.stack([rel_ptr[rows], rel_ptr[rows + 1], rows], dim=1)
            items = torch.cat([pieces, whole], dim=0).to(torch.int32).contiguous()
            src_ptr = self.frontier_index[0]
            if E:
                key = plan.node_a.long() * E + torch.arange(E, device=dev)       # (source, plan position), positions distinct
                src_relpos = (torch.sort(key).values % E).to(torch.int32).contiguous()
            else:
                src_relpos = torch.zeros(1, dtype=torch.int32, device=dev)
            self._boundary_relation_index = (items, src_ptr, src_relpos)
        return self._boundary_relation_index

    @property
    def frontier_runs(self):
        """``(run_prefix, max_runs)`` for :func:`functional.first_layer_forward`: for every edge of the ``by_src`` order
        (sorted by (src, dst, rel)) the number of (source, destination) runs that start at or before it -- int32 ``(E,)`` -- and
        the largest number of distinct destinations any source node has.  Built once per graph; a reweighted RelCSR shares
        its base's."""
        base = getattr(self, "_base", None)
        if base is not None:
            return base.frontier_runs
        if getattr(self, "_frontier_runs", None) is None:
            _ = self.by_src
            order = self._by_src_order
            src, dst = self.src[order], self.dst[order]
            start = torch.ones(self.n_edges, dtype=torch.bool, device=self.device)
            if self.n_edges > 1:
                start[1:] = (src[1:] != src[:-1]) | (dst[1:] != dst[:-1])
            prefix = torch.cumsum(start.to(torch.int32), 0, dtype=torch.int32)
            runs = torch.bincount(src[start], minlength=self.shape[1]) if self.n_edges else torch.zeros(1, dtype=torch.long)
            self._frontier_runs = (prefix.contiguous(), int(runs.max()) if runs.numel() else 0)
        return self._frontier_runs

    @property
    def frontier_fraction(self):
        """Expected fraction of the nodes that the out-edges of ONE boundary node reach, for boundary nodes drawn like the heads of
        a training batch -- an edge picked at random, i.e. weighted by out-degree: ``sum_u runs(u) * deg(u) / (E * N)`` with
        ``runs(u)`` the distinct destinations of ``u``.  A property of the graph (cached; one host read), used to decide statically
        -- capturable -- whether the first layer of a training step runs sparse (``functional.first_layer_train_forward``)."""
        base = getattr(self, "_base", None)
        if base is not None:
            return base.frontier_fraction
        if getattr(self, "_frontier_fraction", None) is None:
            n_dst, n_src, _ = self.shape
            if not self.n_edges or not n_dst:
                self._frontier_fraction = 0.0
            else:
                runs = torch.bincount(torch.unique(self.src * n_dst + self.dst) // n_dst, minlength=n_src).double()
                deg = torch.bincount(self.src, minlength=n_src).double()
                self._frontier_fraction = float((runs * deg).sum() / (deg.sum() * n_dst))
        return self._frontier_fraction

    def with_edge_weights(self, edge_weight):
        """RelCSR over the same edge set with other weights, given per ORIGINAL (un-coalesced) edge; duplicates of
        one triple add up, as ``coalesce()`` would.  Shares the sorted index arrays and chunk schedules."""
        other = RelCSR.__new__(RelCSR)
        other.shape, other._opts = self.shape, self._opts
        other.chunk_edges, other.piece_len = self.chunk_edges, self.piece_len
        other.dst, other.src, other.rel_id = self.dst, self.src, self.rel_id
        other.edge_of_input, other.n_edges = self.edge_of_input, self.n_edges
        w = torch.zeros(self.n_edges, dtype=torch.float32, device=self.device)
        w.index_add_(0, self.edge_of_input, edge_weight.to(torch.float32))
        other.weight, other.unit_weight = w, False
        other.dense_form = False                    # per-edge weights: the edge list is walked
        # the object that owns the sorted plans and their permutations: reweighting a reweighted RelCSR goes back to it
        other._base = getattr(self, "_base", None) or self
        other._fwd = other._by_src = other._by_rel = None
        return other

    def with_removed_edges(self, h, t, r, n_base_rel):
        """RelCSR over the same edge set in which the edges ``(h[i] -> t[i], r[i])`` and their inverses
        ``(t[i] -> h[i], r[i] + n_base_rel)`` carry weight 0 (``remove_easy_edges``, ``ultra/model.py:57-74``, on the
        graph with inverse edges; triples that are not edges are ignored).  One native call fills the weight arrays of
        all three plans by binary search in their sorted index arrays: no ``match``, no host synchronisation, static
        shapes (capturable).  Shares every index array and schedule with this object."""
        base = getattr(self, "_base", None) or self
        lib = _lib.load()
        dev = self.device
        h, t, r = (x.reshape(-1).contiguous() for x in (h, t, r))
        E = base.n_edges
        plans = (self.fwd, self.by_src, self.by_rel)         # their weights are the starting point
        w = [torch.empty(E + PACK_SLACK, dtype=torch.float32, device=dev) for _ in range(3)]
        # a graph of unit weights whose packed words leave bit 31 free: the removed edges ALSO as marks in copies of the words, on
        # which the sum / mul kernels run their unit-weight form (csrc/quad.inc DEAD) instead of loading a weight per edge
        n_node = max(self.shape[0], self.shape[1])
        marks = [None, None, None]
        if (DEAD_EDGE_WORDS and self.unit_weight and self.shape[0] == self.shape[1] and E and
                all(p.weight is None and p.packed is not None and p.packed_src_shift < 31
                    and (n_node - 1) >> (31 - p.packed_src_shift) == 0 for p in plans)):
            marks = [torch.empty(E + PACK_SLACK, dtype=torch.int32, device=dev) for _ in range(3)]
        with torch.cuda.device(dev):
            if marks[0] is not None:
                _lib.check(lib.ultra_edge_removal_marks(
                    plans[0].pointer, plans[1].pointer, plans[2].pointer, h.data_ptr(), t.data_ptr(), r.data_ptr(),
                    h.numel(), int(n_base_rel), w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(), PACK_SLACK,
                    marks[0].data_ptr(), marks[1].data_ptr(), marks[2].data_ptr(), n_node, torch.cuda.current_stream().cuda_stream))
            else:
                _lib.check(lib.ultra_edge_removal_weights(
                    plans[0].pointer, plans[1].pointer, plans[2].pointer, h.data_ptr(), t.data_ptr(), r.data_ptr(),
                    h.numel(), int(n_base_rel), w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(), PACK_SLACK,
                    torch.cuda.current_stream().cuda_stream))
        other = RelCSR.__new__(RelCSR)
        other.shape, other._opts = self.shape, self._opts
        other.chunk_edges, other.piece_len = self.chunk_edges, self.piece_len
        other.dst, other.src, other.rel_id = self.dst, self.src, self.rel_id
        other.edge_of_input, other.n_edges = self.edge_of_input, self.n_edges
        other.weight, other.unit_weight = w[0][:E], False
        other.dense_form = False
        other._base = base
        other._fwd = plans[0].with_weight_buffer(w[0], marks[0])
        other._by_src = plans[1].with_weight_buffer(w[1], marks[1])
        other._by_rel = plans[2].with_weight_buffer(w[2], marks[2])
        return other

    @property
    def csr_arrays(self):
        """The coalesced adjacency as the raw-CSR operators of ``torch.ops.ultra_mi`` take it: ``(row_ptr int32 (n_dst +
        1,), src int32 (E,), rel int32 (E,), w fp32 (E,) | None)``, rows in (src, rel) order.  Cached; a re-weighted
        RelCSR shares the index arrays of its base."""
        base = getattr(self, "_base", None) or self
        if getattr(base, "_csr_index", None) is None:
            n_dst = base.shape[0]
            row_ptr = torch.zeros(n_dst + 1, dtype=torch.long, device=base.device)
            if base.n_edges:
                torch.cumsum(torch.bincount(base.dst, minlength=n_dst), 0, out=row_ptr[1:])
            base._csr_index = (row_ptr.to(torch.int32).contiguous(), base.src.to(torch.int32).contiguous(),
                               base.rel_id.to(torch.int32).contiguous())
        return base._csr_index + (None if self.unit_weight else self.weight.contiguous(),)

    def degree_in(self):
        """Weighted in-degree per destination row (``graph.degree_out`` of the transposed adjacency)."""
        out = torch.zeros(self.shape[0], dtype=torch.float32, device=self.device)
        return out.index_add_(0, self.dst, self.weight)

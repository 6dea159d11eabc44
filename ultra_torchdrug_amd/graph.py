"""Minimal relational graph container: only what the hot path and its callers touch.

Stands in for ``torchdrug.data.Graph`` (un-vendored) at the call sites the reference uses:
``edge_list`` / ``edge_weight`` / ``num_node`` / ``num_relation`` / ``num_edge`` (``ultra/layer.py:56,82-86``),
``degree_out`` (``:86,121``), ``adjacency`` (``:127,328``), ``undirected(add_inverse=True)``
(``ultra/model.py:166``, ``ultra/rel_model.py:92``), ``edge_mask`` and ``match``
(``ultra/model.py:72-74``, ``ultra/task.py:48,70``), free-form per-call attributes ``query`` / ``boundary``
(``ultra/model.py:111-114``).  Added: ``relcsr`` -- the cached sorted/coalesced plans the HIP kernels walk, built
once per graph instead of once per rspmm call.
"""
import torch

from .relcsr import RelCSR


class Graph:
    """``edge_list``: int64 ``(E, 3)`` rows of (node_in, node_out, relation) or ``(E, 2)`` without relations."""

    def __init__(self, edge_list=None, edge_weight=None, num_node=None, num_relation=None, **attributes):
        if edge_list is None:
            edge_list = torch.zeros(0, 3 if num_relation else 2, dtype=torch.long)
        edge_list = torch.as_tensor(edge_list, dtype=torch.long)
        if edge_list.dim() != 2 or edge_list.shape[1] not in (2, 3):
            raise ValueError("edge_list must be (E, 2) or (E, 3), got %s" % (tuple(edge_list.shape),))
        if num_node is None:
            num_node = int(edge_list[:, :2].max()) + 1 if edge_list.numel() else 0
        if edge_list.shape[1] == 3 and num_relation is None:
            num_relation = int(edge_list[:, 2].max()) + 1 if edge_list.numel() else 0
        if edge_list.numel() and int(edge_list[:, :2].max()) >= num_node:
            raise ValueError("`num_node` is %d, but found node %d in `edge_list`" % (num_node, int(edge_list[:, :2].max())))
        if edge_weight is None:
            edge_weight = torch.ones(len(edge_list), device=edge_list.device)
        self.edge_list = edge_list
        self.edge_weight = torch.as_tensor(edge_weight, dtype=torch.float, device=edge_list.device)
        self.num_node = int(num_node)
        self.num_relation = int(num_relation) if num_relation is not None else None
        self.requires_grad = False
        self._relcsr = None
        self._adjacency = None
        self._match_index = None
        self._completion = None
        for k, v in attributes.items():
            setattr(self, k, v)

    # -------------------------------------------------------------- basic properties
    @property
    def num_edge(self):
        return self.edge_list.shape[0]

    @property
    def device(self):
        return self.edge_list.device

    @property
    def degree_out(self):
        """Weighted number of edges that have each node as ``node_out`` (torchdrug's naming)."""
        out = torch.zeros(self.num_node, device=self.device, dtype=self.edge_weight.dtype)
        return out.index_add_(0, self.edge_list[:, 1], self.edge_weight)

    @property
    def degree_in(self):
        out = torch.zeros(self.num_node, device=self.device, dtype=self.edge_weight.dtype)
        return out.index_add_(0, self.edge_list[:, 0], self.edge_weight)

    @property
    def adjacency(self):
        """Sparse COO ``(num_node, num_node[, num_relation])`` indexed (node_in, node_out[, relation])."""
        if self._adjacency is None:
            shape = (self.num_node, self.num_node) + ((self.num_relation,) if self.edge_list.shape[1] == 3 else ())
            self._adjacency = torch.sparse_coo_tensor(self.edge_list.t(), self.edge_weight, shape)
        return self._adjacency

    @property
    def relcsr(self):
        """Plans over ``adjacency.transpose(0, 1)`` (destination = node_out), cached for the life of the graph."""
        if self._relcsr is None:
            if self.edge_list.shape[1] != 3:
                raise ValueError("relcsr needs a relational graph (edge_list with 3 columns)")
            self._relcsr = RelCSR.from_edge_list(self.edge_list, self.edge_weight, self.num_node, self.num_relation)
        return self._relcsr

    # -------------------------------------------------------------- device movement
    def to(self, device):
        g = Graph(self.edge_list.to(device), self.edge_weight.to(device), self.num_node, self.num_relation)
        for k, v in self.__dict__.items():
            if k.startswith("_") or k in ("edge_list", "edge_weight", "num_node", "num_relation"):
                continue
            setattr(g, k, v.to(device) if isinstance(v, torch.Tensor) else v)
        return g

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def cpu(self):
        return self.to(torch.device("cpu"))

    def clone(self):
        return self.to(self.device)

    # -------------------------------------------------------------- transformations
    def undirected(self, add_inverse=False):
        """Every edge followed by its flipped copy (relation ``r + num_relation`` when ``add_inverse``)."""
        flipped = self.edge_list.clone()
        flipped[:, :2] = flipped[:, :2].flip(1)
        num_relation = self.num_relation
        if num_relation and add_inverse:
            flipped[:, 2] += num_relation
            num_relation = num_relation * 2
        edge_list = torch.stack([self.edge_list, flipped], dim=1).flatten(0, 1)
        edge_weight = self.edge_weight.repeat_interleave(2)
        return Graph(edge_list, edge_weight, self.num_node, num_relation)

    def reweighted(self, edge_weight):
        """Same nodes, edges and edge order, other weights; the sorted plans (``relcsr``) of this graph are SHARED,
        only their weight arrays are new.  Used to drop edges for one training step by zeroing them
        (``ultra/model.py:57-74`` builds a new graph -- and torchdrug re-sorts it -- every step)."""
        edge_weight = torch.as_tensor(edge_weight, dtype=torch.float, device=self.device)
        if edge_weight.shape != self.edge_weight.shape:
            raise ValueError("edge_weight must have one entry per edge")
        g = Graph.__new__(Graph)            # same (already validated) edge list: no range check, no host sync
        g.edge_list, g.edge_weight = self.edge_list, edge_weight
        g.num_node, g.num_relation = self.num_node, self.num_relation
        g.requires_grad, g._relcsr, g._adjacency = False, None, None
        g._match_index = self._match_index
        if self.edge_list.shape[1] == 3:
            g._relcsr = self.relcsr.with_edge_weights(edge_weight)
        return g

    def without_triples(self, h, t, r, n_base_rel):
        """This graph (one WITH inverse edges, ``undirected(add_inverse=True)``) minus the edges ``(h, t, r)`` and
        their inverses ``(t, h, r + n_base_rel)``, as zero weights on the shared plans (``RelCSR.with_removed_edges``):
        what ``remove_easy_edges`` + ``undirected`` give (``ultra/model.py:57-74,166``) for summed messages, without a
        new edge list, a re-sort or a host synchronisation.  Only ``relcsr`` of the result reflects the removal."""
        g = Graph.__new__(Graph)
        g.edge_list, g.edge_weight = self.edge_list, self.edge_weight
        g.num_node, g.num_relation = self.num_node, self.num_relation
        g.requires_grad, g._adjacency, g._match_index = False, None, self._match_index
        g._completion = getattr(self, "_completion", None)
        g._relcsr = self.relcsr.with_removed_edges(h, t, r, n_base_rel)
        return g

    def completion_keys(self, anchor_col):
        """Sorted DISTINCT int64 keys ``(anchor * num_relation + relation) * num_node + other`` of the triples, with
        ``anchor`` = column ``anchor_col`` of ``edge_list`` (0: head, answers "which tails complete (h, r, ?)"; 1: tail)
        and ``other`` the opposite column.  Built once per graph; the device-side filter / negative-sampling kernels
        binary-search it (``csrc/sampler.inc``) where the reference builds ``(B, N)`` masks (``ultra/task.py:65-100``)."""
        cache = getattr(self, "_completion", None)
        if cache is None:
            cache = self._completion = {}
        if anchor_col not in cache:
            if self.edge_list.shape[1] != 3:
                raise ValueError("completion_keys needs a relational graph")
            n_rel = max(self.num_relation, 1)
            if float(self.num_node) ** 2 * n_rel >= 2.0 ** 63:
                raise ValueError("graph too large for 64-bit completion keys")
            anchor, other = self.edge_list[:, anchor_col], self.edge_list[:, 1 - anchor_col]
            key = (anchor * n_rel + self.edge_list[:, 2]) * self.num_node + other
            cache[anchor_col] = torch.unique(key)          # sorted, distinct
        return cache[anchor_col]

    def edge_mask(self, index):
        """Keep the edges selected by a bool mask or an index tensor; nodes are kept."""
        return Graph(self.edge_list[index], self.edge_weight[index], self.num_node, self.num_relation)

    def match(self, pattern):
        """Edges matching each pattern row; ``-1`` is a wildcard.  Returns ``(edge_index, num_match)`` with the
        matches of pattern 0 first, then pattern 1, ... (as torchdrug's ``Graph.match``)."""
        pattern = torch.as_tensor(pattern, dtype=torch.long, device=self.device)
        if pattern.dim() == 1:
            pattern = pattern.unsqueeze(0)
        n_col = self.edge_list.shape[1]
        if pattern.shape[1] != n_col:
            raise ValueError("pattern has %d columns, edge_list has %d" % (pattern.shape[1], n_col))
        if pattern.shape[0] == 0 or self.num_edge == 0:
            return (torch.zeros(0, dtype=torch.long, device=self.device),
                    torch.zeros(pattern.shape[0], dtype=torch.long, device=self.device))
        # group patterns by which columns are wildcards; each group is a sorted-key range lookup
        wild = pattern < 0
        code = (wild.long() * (2 ** torch.arange(n_col, device=self.device))).sum(1)
        sizes = [self.num_node, self.num_node] + ([max(self.num_relation, 1)] if n_col == 3 else [])
        num_match = torch.zeros(pattern.shape[0], dtype=torch.long, device=self.device)
        starts = torch.zeros(pattern.shape[0], dtype=torch.long, device=self.device)
        orders = {}
        for c in code.unique().tolist():
            cols = [i for i in range(n_col) if not (c >> i) & 1]
            sel = (code == c).nonzero().flatten()
            if not cols:   # all wildcards: every edge matches
                num_match[sel] = self.num_edge
                orders[c] = torch.arange(self.num_edge, device=self.device)
                continue
            key_p = torch.zeros(sel.numel(), dtype=torch.long, device=self.device)
            for i in cols:
                key_p = key_p * sizes[i] + pattern[sel, i]
            if self._match_index is None:
                self._match_index = {}
            if c not in self._match_index:      # sorted edge keys per wildcard layout, built once per graph
                key_e = torch.zeros(self.num_edge, dtype=torch.long, device=self.device)
                for i in cols:
                    key_e = key_e * sizes[i] + self.edge_list[:, i]
                self._match_index[c] = torch.sort(key_e, stable=True)
            key_sorted, order = self._match_index[c]
            lo = torch.searchsorted(key_sorted, key_p, right=False)
            hi = torch.searchsorted(key_sorted, key_p, right=True)
            num_match[sel] = hi - lo
            starts[sel] = lo
            orders[c] = order
        total = int(num_match.sum())
        owner = torch.repeat_interleave(torch.arange(pattern.shape[0], device=self.device), num_match)
        offset = torch.arange(total, device=self.device) - (num_match.cumsum(0) - num_match)[owner]
        edge_index = torch.empty(total, dtype=torch.long, device=self.device)
        for c, order in orders.items():
            m = code[owner] == c
            edge_index[m] = order[starts[owner[m]] + offset[m]]
        return edge_index, num_match

    def __repr__(self):
        return "Graph(num_node=%d, num_edge=%d, num_relation=%s, device=%s)" % (
            self.num_node, self.num_edge, self.num_relation, self.device)

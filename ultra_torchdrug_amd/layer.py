"""Relational convolution layers of the NBFNet Bellman-Ford iteration, on top of the HIP rspmm.

Mirrors ``/root/reference/ultra/layer.py``:

* :class:`GeneralizedRelationalConvNBF`     -- ``layer.py:14-190``  (relation-graph stack, ``dependent=False``)
* :class:`GeneralizedRelationalConvNBFMod`  -- ``layer.py:193-392`` (entity stack, ``project=True``)

Same constructor arguments, attribute / parameter names (``linear``, ``layer_norm``, ``relation``,
``relation_linear``, ``relation_projection.layers.{0,1}``) and aggregate functions, so reference state dicts load
unchanged.  The two reference classes repeat the same aggregation code; here it lives once in
:class:`_RelationalConvBase`.  ``forward = combine(input, message_and_aggregate(graph, input))`` is the contract
of torchdrug's ``MessagePassingBase``.
"""
import torch
from torch import nn
from torch.nn import functional as F

from . import backend

# First-layer shortcut (csrc/frontier.inc): the first layer of a Bellman-Ford aggregates from the boundary, which is zero
# outside one row per query; with summed DistMult messages only the out-edges of those rows are visited.  Same bits as
# the full kernel; the switch exists so that bench.py can report the step with and without it.
FRONTIER_FIRST_LAYER = True


def _frontier_tables_finite(relation_input):
    """The shortcut's precondition.  ``w * (rel * 0)`` is ``+-0`` only for a FINITE relation entry: the full kernels turn an
    ``inf`` / ``NaN`` entry of the relation table into ``NaN`` at every destination of an edge of that relation (``inf * 0``),
    the frontier kernel never visits those edges.  Eager calls therefore test the table (one tiny reduction + a host
    read, like the reference's own per-call index asserts, ``ultra/model.py:174-175``) and send a non-finite table
    through the full kernel -- the reference's semantics.  While a hipGraph is being captured no host read is possible:
    the capture goes by what ``engine.capture_semantics`` found in the model's parameters when it started
    (``functional.CAPTURE_ASSUMES_FINITE``), as it goes by the triples' range check for the indices."""
    from . import functional
    return functional.tables_finite(relation_input)


class _TallLinear(torch.autograd.Function):
    """``F.linear`` whose weight gradient is reduced in row slices.  The relation projections see ``B * R`` rows (7 584 for
    an FB15k237-sized vocabulary at B = 16) and 64 outputs: ``d_weight = grad^T . input`` is then a 64 x 64 GEMM with
    K = 7 584, for which the BLAS library picks one or two output tiles -- 68 us on one or two CUs, twelve times per
    step.  Cut into 64 row slices it is a batched GEMM over 64 workgroups followed by one small sum."""
    SLICES = 64

    @staticmethod
    def forward(ctx, input, weight, bias):
        ctx.save_for_backward(input, weight)
        return F.linear(input, weight, bias)

    @staticmethod
    def backward(ctx, grad):
        input, weight = ctx.saved_tensors
        g = grad.reshape(-1, grad.shape[-1])
        x = input.reshape(-1, input.shape[-1])
        d_input = (g @ weight).view_as(input) if ctx.needs_input_grad[0] else None
        d_weight = d_bias = None
        if ctx.needs_input_grad[1]:
            rows, s = g.shape[0], _TallLinear.SLICES
            per = -(-rows // s)
            if per * s != rows:         # zero rows add nothing
                g_p = torch.zeros(per * s, g.shape[1], dtype=g.dtype, device=g.device)
                x_p = torch.zeros(per * s, x.shape[1], dtype=x.dtype, device=x.device)
                g_p[:rows], x_p[:rows] = g, x
            else:
                g_p, x_p = g, x
            parts = torch.bmm(g_p.view(s, per, -1).transpose(1, 2), x_p.view(s, per, -1))          # (s, out, in)
            # the sum over the slices as a (1, s) x (s, out * in) product: ATen's strided `sum(0)` of this shape is a
            # multi-block reduction that zeroes its semaphores with a memset, and a memset NODE inside a captured
            # training step is one more node kind whose replay has misbehaved here (see ultra_rspmm_frontier_f32)
            # (as a matrix-VECTOR product: for the (1, s) x (s, out * in) matrix product the BLAS library picks a 47 us kernel)
            ones = torch.ones(s, dtype=parts.dtype, device=parts.device)
            d_weight = torch.mv(parts.view(s, -1).t(), ones).view(parts.shape[1], parts.shape[2])
        if ctx.needs_input_grad[2]:          # (a product for the same reason as above)
            d_bias = (torch.ones(1, g.shape[0], dtype=g.dtype, device=g.device) @ g).view(-1)
        return d_input, d_weight, d_bias


class MLP(nn.Module):
    """``torchdrug.layers.MLP`` as the reference uses it: Linear layers in ``self.layers``, activation between
    them, none after the last (``ultra/layer.py:228``, ``ultra/model.py:53``, ``ultra/rel_model.py:263``)."""

    def __init__(self, input_dim, hidden_dims, short_cut=False, activation="relu"):
        super().__init__()
        if isinstance(hidden_dims, int):
            hidden_dims = [hidden_dims]
        self.dims = [input_dim] + list(hidden_dims)
        self.short_cut = short_cut
        self.activation = getattr(F, activation) if isinstance(activation, str) else activation
        self.layers = nn.ModuleList(nn.Linear(self.dims[i], self.dims[i + 1]) for i in range(len(self.dims) - 1))

    def _library_linear(self, layer, x, relu):
        """Inference on the GPU: the layer runs in libultra_rspmm's documented summation order (bit-identical to the
        CPU oracle); training and unsupported shapes keep nn.Linear."""
        ops = backend.get()
        if not (ops.accepts(x) and x.dtype == torch.float32 and layer.bias is not None
                and ops.linear_supported(layer.in_features, layer.out_features)):
            return None
        if torch.is_grad_enabled() and (x.requires_grad or layer.weight.requires_grad):
            return None
        return ops.linear_forward(x, layer.weight, layer.bias, relu=relu)

    def forward(self, input):
        layer_input = input
        for i, layer in enumerate(self.layers):
            act = i < len(self.layers) - 1 and bool(self.activation)
            fused_relu = act and self.activation is F.relu
            hidden = self._library_linear(layer, layer_input, fused_relu)
            if hidden is None:
                rows = layer_input.numel() // max(layer_input.shape[-1], 1)
                if (layer_input.is_cuda and torch.is_grad_enabled() and layer.weight.requires_grad and layer.bias is not None
                        and rows >= 2048 and layer.out_features <= 128):
                    hidden = _TallLinear.apply(layer_input, layer.weight, layer.bias)
                else:
                    hidden = layer(layer_input)
                fused_relu = False
            if act and not fused_relu:
                hidden = self.activation(hidden)
            if self.short_cut and hidden.shape == layer_input.shape:
                hidden = hidden + layer_input
            layer_input = hidden
        return hidden


class _RelationalConvBase(nn.Module):
    eps = 1e-6
    message2mul = {"transe": "add", "distmult": "mul"}   # layer.py:18-21

    def _init_common(self, input_dim, output_dim, num_relation, query_input_dim, message_func, aggregate_func,
                     layer_norm, activation):
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.num_relation = num_relation
        self.query_input_dim = query_input_dim
        self.message_func = message_func
        self.aggregate_func = aggregate_func
        self.layer_norm = nn.LayerNorm(output_dim) if layer_norm else None
        self.activation = getattr(F, activation) if isinstance(activation, str) else activation
        width = 13 if aggregate_func in ("pna", "pna_nobound") else 2    # layer.py:43-46
        self.linear = nn.Linear(input_dim * width, output_dim)

    # ---- subclasses provide the (num_relation, batch, dim)-shaped relation table ------------------------
    def _relation_table(self, graph, batch_size):
        raise NotImplementedError

    def forward(self, graph, input, shortcut=False, input_is_boundary=False, grad_tiles=None, grad_rows=None):
        """``MessagePassingBase.forward``: ``combine(input, message_and_aggregate(graph, input))``.  ``shortcut``
        additionally adds ``input`` (the caller's ``hidden + layer_input``, ``ultra/model.py:126-127``) so that the
        inference path can run combine + shortcut as ONE HIP kernel.  ``input_is_boundary``: the caller's promise that
        ``input`` is ``graph.boundary`` (the first layer of a Bellman-Ford, ``ultra/model.py:116-120``)."""
        fused = self._sum_layer(graph, input, shortcut, input_is_boundary, grad_tiles, grad_rows)
        if fused is not None:
            return fused
        update = self.message_and_aggregate(graph, input, input_is_boundary=input_is_boundary)
        if self._fusable(input, update):
            ln = self.layer_norm
            return backend.get().combine(input, update, self.linear.weight, self.linear.bias,
                                      ln.weight if ln else None, ln.bias if ln else None,
                                      ln.eps if ln else 1e-5, relu=self.activation is F.relu, shortcut=shortcut,
                                      reuse_update=True)       # `update` is this call's own temporary
        output = self.combine(input, update)
        return output + input if shortcut else output

    def _sum_layer(self, graph, input, shortcut, input_is_boundary=False, grad_tiles=None, grad_rows=None):
        """Training with summed messages and the shipped 64 -> 64 epilogue: the whole layer as one autograd node
        (``backend.sum_layer``), so that the two gradients of ``input`` (through the edges, through the epilogue) are
        produced by one kernel sequence without an add pass.  ``None``: not applicable, take the general path."""
        if graph.requires_grad or self.message_func not in self.message2mul or self.aggregate_func != "sum":
            return None
        if not self._fusable(input, input) or not torch.is_grad_enabled():
            return None
        ops = backend.get()
        batch_size = len(graph.query)
        tables = getattr(graph, "relation_tables", None)
        if tables is not None and id(self) in tables:
            relation_input = tables[id(self)]
        else:
            relation_input = self._relation_table(graph, batch_size).flatten(1)
        if not any(t.requires_grad for t in (input, relation_input, self.linear.weight)):
            return None
        ln = self.layer_norm
        return ops.sum_layer(graph.relcsr, relation_input, input, graph.boundary, getattr(graph, "boundary_sparse", None),
                             self.message2mul[self.message_func], self.linear.weight, self.linear.bias,
                             ln.weight if ln else None, ln.bias if ln else None, ln.eps if ln else 1e-5,
                             relu=self.activation is F.relu, shortcut=shortcut, input_is_boundary=input_is_boundary,
                             grad_tiles=grad_tiles, grad_rows=grad_rows)

    def _no_grad(self, *tensors):
        return not torch.is_grad_enabled() or not any(t.requires_grad for t in tensors if t is not None)

    def _fusable(self, input, update):
        """The fused epilogue kernels (forward and backward) cover the shipped layer shape: 64 -> 64, concat of 2,
        relu or no activation; anything else runs the reference's ATen ops."""
        ln = self.layer_norm
        return (backend.get().accepts(input) and input.dtype == torch.float32 and input.shape == update.shape
                and input.shape[-1] == 64 and self.output_dim == 64 and tuple(self.linear.weight.shape) == (64, 128)
                and (self.activation is F.relu or not self.activation)
                and (ln is None or (ln.elementwise_affine and ln.bias is not None)))

    # ---- O(E) definition, used for rotate / graphs that require grad (layer.py:52-109, :232-296) ---------
    def message(self, graph, input):
        batch_size = len(graph.query)
        node_in, node_out, relation = graph.edge_list.t()
        relation_input = self._relation_table(graph, batch_size)          # (R, B, D)
        node_input = input[node_in]
        edge_input = relation_input[relation]
        if self.message_func == "transe":
            message = edge_input + node_input
        elif self.message_func == "distmult":
            message = edge_input * node_input
        elif self.message_func == "rotate":
            node_re, node_im = node_input.chunk(2, dim=-1)
            edge_re, edge_im = edge_input.chunk(2, dim=-1)
            message = torch.cat([node_re * edge_re - node_im * edge_im, node_re * edge_im + node_im * edge_re], dim=-1)
        else:
            raise ValueError("Unknown message function `%s`" % self.message_func)
        return torch.cat([message, graph.boundary])

    def aggregate(self, graph, message):
        n = graph.num_node
        node_out = torch.cat([graph.edge_list[:, 1], torch.arange(n, device=graph.device)])
        edge_weight = torch.cat([graph.edge_weight, torch.ones(n, device=graph.device)]).unsqueeze(-1).unsqueeze(-1)
        degree_out = graph.degree_out.unsqueeze(-1).unsqueeze(-1) + 1
        index = node_out.view(-1, 1, 1).expand_as(message)
        weighted = message * edge_weight

        def scatter(src, reduce):
            # the fill never enters the result (include_self=False; every node receives its boundary message), but ATen's
            # amax / amin backward counts a fill that EQUALS the result as one more tied element: fill with the identity
            fill = {"amax": float("-inf"), "amin": float("inf")}.get(reduce, 0.0)
            out = torch.full((n,) + tuple(message.shape[1:]), fill, device=message.device, dtype=message.dtype)
            return out.scatter_reduce(0, index, src, reduce=reduce, include_self=False)

        if self.aggregate_func == "sum":
            return scatter(weighted, "sum")
        if self.aggregate_func == "mean":
            return scatter(weighted, "mean")
        if self.aggregate_func == "max":
            return scatter(weighted, "amax")
        if self.aggregate_func == "pna":
            mean = scatter(weighted, "mean")
            sq_mean = scatter(message ** 2 * edge_weight, "mean")
            return self._pna(mean, sq_mean, scatter(weighted, "amax"), scatter(weighted, "amin"), degree_out)
        raise ValueError("Unknown aggregation function `%s`" % self.aggregate_func)

    def _pna(self, mean, sq_mean, max, min, degree_out):
        # layer.py:147-153 / :172-178
        std = (sq_mean - mean ** 2).clamp(min=self.eps).sqrt()
        features = torch.cat([mean.unsqueeze(-1), max.unsqueeze(-1), min.unsqueeze(-1), std.unsqueeze(-1)], dim=-1)
        features = features.flatten(-2)
        scale = degree_out.log()
        scale = scale / scale.mean()
        scales = torch.cat([torch.ones_like(scale), scale, 1 / scale.clamp(min=1e-2)], dim=-1)
        return (features.unsqueeze(-1) * scales.unsqueeze(-2)).flatten(-2)

    # ---- rspmm path (layer.py:111-182, :298-384) -------------------------------------------------------
    def message_and_aggregate(self, graph, input, input_is_boundary=False):
        if graph.requires_grad or self.message_func == "rotate":
            return self.aggregate(graph, self.message(graph, input))
        if self.message_func not in self.message2mul:
            raise ValueError("Unknown message function `%s`" % self.message_func)
        mul = self.message2mul[self.message_func]

        batch_size = len(graph.query)
        input = input.flatten(1)
        boundary = graph.boundary.flatten(1)
        tables = getattr(graph, "relation_tables", None)       # all layers' tables from one launch (model.bellmanford)
        if tables is not None and id(self) in tables:
            relation_input = tables[id(self)]
        else:
            relation_input = self._relation_table(graph, batch_size).flatten(1)    # (R, B*D)
        adjacency = graph.relcsr        # cached plans of graph.adjacency.transpose(0, 1)
        func = self.aggregate_func
        bound = not func.endswith("_nobound")
        kind = func[:-len("_nobound")] if not bound else func
        ops = backend.get()
        rspmm = ops.generalized_rspmm

        if kind not in ("sum", "mean", "max", "pna"):
            raise ValueError("Unknown aggregation function `%s`" % self.aggregate_func)
        degree_out = None
        if kind in ("mean", "pna"):
            degree_out = graph.degree_out.unsqueeze(-1) + 1
        # inference: `update + boundary` / `max(update, boundary)` ride along in the rspmm kernel (bit-identical)
        fuse_bound = bound and ops.accepts(input) and self._no_grad(input, relation_input, boundary)
        # the boundary in its sparse form (node per query, value per query), when the caller attached it
        sparse_bound = getattr(graph, "boundary_sparse", None) if fuse_bound else None
        bound_args = dict(add_rows=boundary) if sparse_bound is None else dict(boundary=sparse_bound)
        if kind in ("sum", "mean"):
            if (fuse_bound and input_is_boundary and sparse_bound is not None and FRONTIER_FIRST_LAYER
                    and ops.frontier_supported("add", mul, input.shape[1]) and sparse_bound[1].shape[-1] == 64
                    and _frontier_tables_finite(relation_input)):
                # first layer: only the out-edges of the boundary rows carry non-zero messages (csrc/frontier.inc)
                update = ops.rspmm_frontier(adjacency, relation_input, sparse_bound)
            elif fuse_bound:
                update = ops.rspmm_forward(adjacency, relation_input, input, "add", mul, **bound_args)
            elif bound and ops.accepts(input):      # training
                sparse_train = getattr(graph, "boundary_sparse", None)
                if sparse_train is not None:
                    update = ops.rspmm_sum_plus(adjacency, relation_input, input, None, mul=mul, boundary=sparse_train)
                else:
                    update = ops.rspmm_sum_plus(adjacency, relation_input, input, boundary, mul=mul)
            else:
                update = rspmm(adjacency, relation_input, input, sum="add", mul=mul)
                if bound:
                    update = update + boundary
            if kind == "mean":
                update = update / degree_out
        elif kind == "max":
            if fuse_bound:
                update = ops.rspmm_forward(adjacency, relation_input, input, "max", mul, **bound_args)
            else:
                update = rspmm(adjacency, relation_input, input, sum="max", mul=mul)
                if bound:
                    update = torch.max(update, boundary)
        else:
            sum = rspmm(adjacency, relation_input, input, sum="add", mul=mul)
            sq_sum = rspmm(adjacency, relation_input ** 2, input ** 2, sum="add", mul=mul)
            max = rspmm(adjacency, relation_input, input, sum="max", mul=mul)
            min = rspmm(adjacency, relation_input, input, sum="min", mul=mul)
            if bound:
                sum, sq_sum = sum + boundary, sq_sum + boundary ** 2
                max, min = torch.max(max, boundary), torch.min(min, boundary)
            update = self._pna(sum / degree_out, sq_sum / degree_out, max, min, degree_out)
        return update.view(len(update), batch_size, -1)

    def combine(self, input, update):
        # layer.py:184-190 / :386-392
        output = self.linear(torch.cat([input, update], dim=-1))
        if self.layer_norm:
            output = self.layer_norm(output)
        if self.activation:
            output = self.activation(output)
        return output


class GeneralizedRelationalConvNBF(_RelationalConvBase):
    """``ultra/layer.py:14-190``.  ``dependent``: relation table is a linear map of the query (``:48,58,123``);
    otherwise a learned ``nn.Embedding`` shared by the whole batch (``:50,60,126``)."""

    def __init__(self, input_dim, output_dim, num_relation, query_input_dim, message_func="distmult",
                 aggregate_func="pna", layer_norm=False, activation="relu", dependent=True):
        super().__init__()
        self._init_common(input_dim, output_dim, num_relation, query_input_dim, message_func, aggregate_func,
                          layer_norm, activation)
        self.dependent = dependent
        if dependent:
            self.relation_linear = nn.Linear(query_input_dim, num_relation * input_dim)
        else:
            self.relation = nn.Embedding(num_relation, input_dim)

    def _relation_table(self, graph, batch_size):
        assert graph.num_relation == self.num_relation     # layer.py:53,115
        if self.dependent:
            table = self.relation_linear(graph.query).view(batch_size, self.num_relation, self.input_dim)
            return table.transpose(0, 1)
        return self.relation.weight.unsqueeze(1).expand(-1, batch_size, -1)


class GeneralizedRelationalConvNBFMod(_RelationalConvBase):
    """``ultra/layer.py:193-392``.  ``self.relation`` is installed by the caller before every forward
    (``ultra/model.py:149-156``): either ``(R, D)`` or per-query ``(B, R, D)``; ``project`` passes it through a
    2-layer MLP first (``:228,318-319``)."""

    def __init__(self, input_dim, output_dim, num_relation, query_input_dim, message_func="distmult",
                 aggregate_func="pna", layer_norm=False, activation="relu", project=True):
        super().__init__()
        self._init_common(input_dim, output_dim, num_relation, query_input_dim, message_func, aggregate_func,
                          layer_norm, activation)
        self.project = project
        self.relation_projection = MLP(input_dim=query_input_dim, hidden_dims=[input_dim, input_dim])
        self.relation = None

    def _relation_table(self, graph, batch_size):
        relation_input = self.relation if isinstance(self.relation, torch.Tensor) else self.relation.weight
        if self.project:
            relation_input = self.relation_projection(relation_input)
        if relation_input.dim() == 2:                          # (R, D) shared by the batch  (layer.py:323-324)
            return relation_input.unsqueeze(1).expand(-1, batch_size, -1)
        return relation_input.transpose(1, 0)                  # (B, R, D) -> (R, B, D)      (layer.py:325-326)

"""Relation-graph model (hot-path caller), mirroring ``/root/reference/ultra/rel_model.py``.

* :func:`construct_relation_graph` -- ``rel_model.py:91-181`` for the shipped configuration
  (``multirelational=True``: 2R relation nodes, 4 edge types hh / tt / ht / th).  One-off preprocessing, done
  with ATen sparse products exactly as the reference does.
* :class:`CustomNBFNetFull` -- ``rel_model.py:227-378``: 6 x ``GeneralizedRelationalConvNBF(dependent=False)``
  with one Bellman-Ford per query relation.
* :class:`RelNBFNet` -- ``rel_model.py:381-416``; :class:`RelationModelList` -- ``:209-223``.

Parameter names follow the reference (``rel_models.0.model.layers.{i}.{linear,layer_norm,relation}``,
``rel_models.0.model.mlp.layers.{0,1}``) so checkpoints load unchanged.
"""
import os
from collections.abc import Sequence

import torch
from torch import nn

from . import layer
from .graph import Graph


def construct_relation_graph(graph):
    """Graph of relations: nodes = 2R relations (with inverses), edge type 0..3 = head-head, tail-tail,
    head-tail, tail-head co-occurrence of two relations at an entity (``rel_model.py:91-143``)."""
    graph = graph.undirected(add_inverse=True)
    device = graph.device
    n_rel, n_node = graph.num_relation, graph.num_node
    if graph.edge_list.is_cuda and 0 < n_rel <= 32768:
        # on the MI355X: the same four index patterns from one native launch over the entities' relation lists
        # (functional.relation_graph_blocks); nonzero() lists a block in the (row, column) order coalesce() gives
        from . import functional
        blocks = functional.relation_graph_blocks(graph.edge_list, n_node, n_rel)
        edges = []
        for etype in range(4):
            idx = blocks[etype].nonzero()
            edges.append(torch.cat([idx, torch.full((idx.shape[0], 1), etype, dtype=torch.long, device=device)], dim=1))
        return Graph(torch.cat(edges, dim=0), num_node=n_rel, num_relation=4)

    def incidence(col):
        pairs = graph.edge_list[:, [col, 2]].unique(dim=0)                     # (entity, relation)
        degree = torch.zeros(n_node, dtype=torch.long, device=device).index_add_(
            0, pairs[:, 0], torch.ones_like(pairs[:, 1]))
        assert not (degree[pairs[:, 0]] == 0).any()
        ones = torch.ones(pairs.shape[0], device=device)
        normalised_t = torch.sparse_coo_tensor(pairs.flip(1).t(), ones / degree[pairs[:, 0]], (n_rel, n_node))
        plain = torch.sparse_coo_tensor(pairs.t(), ones, (n_node, n_rel))
        return normalised_t, plain

    EhT, Eh = incidence(0)
    EtT, Et = incidence(1)
    blocks = [torch.sparse.mm(EhT, Eh), torch.sparse.mm(EtT, Et), torch.sparse.mm(EhT, Et), torch.sparse.mm(EtT, Eh)]
    edges = []
    for etype, block in enumerate(blocks):
        idx = block.coalesce().indices().t()
        edges.append(torch.cat([idx, torch.full((idx.shape[0], 1), etype, dtype=torch.long, device=device)], dim=1))
    return Graph(torch.cat(edges, dim=0), num_node=n_rel, num_relation=4)


TILED_TABLES_TRAIN = os.environ.get("ULTRA_TILED_TABLES_TRAIN", "1") != "0"


class _TiledTables(torch.autograd.Function):
    """``weight.unsqueeze(1).expand(-1, B, -1).flatten(1)`` (layer.py:125-126) of all layers' relation embeddings: two
    launches forward, two backward (the sum over the B copies of every layer's table gradient at once)."""

    @staticmethod
    def forward(ctx, batch_size, *weights):
        n_rel, dim = weights[0].shape
        ctx.batch_size = batch_size
        tiled = torch.stack(weights).unsqueeze(2).expand(-1, -1, batch_size, -1).reshape(len(weights), n_rel, batch_size * dim)
        return tuple(tiled.unbind(0))

    @staticmethod
    def backward(ctx, *grads):
        live = next(g for g in grads if g is not None)
        dim = live.shape[1] // ctx.batch_size
        total = torch.stack([torch.zeros_like(live) if g is None else g for g in grads])
        total = total.view(len(grads), live.shape[0], ctx.batch_size, dim).sum(2)
        return (None,) + tuple(None if g is None else t for g, t in zip(grads, total.unbind(0)))


class CustomNBFNet(nn.Module):
    """``rel_model.py:227-339`` (constructor and module tree; the shipped model is the ``Full`` subclass)."""

    def __init__(self, input_dim, hidden_dims, num_relation=None, symmetric=False, message_func="distmult",
                 aggregate_func="pna", short_cut=False, layer_norm=False, activation="relu", concat_hidden=False,
                 num_mlp_layer=2, dependent=False, remove_one_hop=False, num_beam=10, path_topk=10,
                 separate_remove_one_hop=False):
        super().__init__()
        if not isinstance(hidden_dims, Sequence):
            hidden_dims = [hidden_dims]
        num_relation = 1 if num_relation is None else int(num_relation)
        self.dims = [input_dim] + list(hidden_dims)
        self.num_relation = num_relation
        self.symmetric = symmetric
        self.short_cut = short_cut
        self.concat_hidden = concat_hidden
        self.remove_one_hop = remove_one_hop
        self.layers = nn.ModuleList(
            layer.GeneralizedRelationalConvNBF(self.dims[i], self.dims[i + 1], num_relation, self.dims[0],
                                               message_func, aggregate_func, layer_norm, activation, dependent)
            for i in range(len(self.dims) - 1))
        feature_dim = hidden_dims[-1] * (len(hidden_dims) if concat_hidden else 1) + input_dim
        self.mlp = layer.MLP(feature_dim, [feature_dim] * (num_mlp_layer - 1) + [hidden_dims[-1]])   # unused in forward

    def _tiled_tables(self, graph, batch_size):
        """Inference: the query-independent relation tables of all layers (``relation.weight.repeat(1, B)``,
        layer.py:125-126) tiled by ONE copy instead of one per layer."""
        convs = list(self.layers)
        if not convs or any(conv.dependent for conv in convs):
            return None
        weights = [conv.relation.weight for conv in convs]
        if any(w.shape != weights[0].shape or not w.is_cuda for w in weights):
            return None
        if torch.is_grad_enabled():
            # training: the same tiling as ONE autograd node -- per layer it is a copy forward and a reduction backward
            if not TILED_TABLES_TRAIN or any(w.dtype != torch.float32 for w in weights):
                return None
            return {id(conv): table for conv, table in zip(convs, _TiledTables.apply(batch_size, *weights))}
        n_rel, dim = weights[0].shape
        tiled = torch.stack(weights).unsqueeze(2).expand(-1, -1, batch_size, -1).reshape(len(convs), n_rel, batch_size * dim)
        return {id(conv): tiled[i] for i, conv in enumerate(convs)}

    def _run_layers(self, graph, boundary):
        graph.relation_tables = self._tiled_tables(graph, boundary.shape[1])
        layer_input = boundary
        for conv in self.layers:
            # shortcut (rel_model.py:371-372) applied inside the layer call
            hidden = conv(graph, layer_input, shortcut=self.short_cut and conv.output_dim == layer_input.shape[-1],
                          input_is_boundary=layer_input is boundary)
            layer_input = hidden
        return layer_input

    def bellmanford(self, graph, h_index, separate_grad=False):
        """``rel_model.py:268-302``: ONE graph in which all query relations are labelled together."""
        dev = h_index.device
        query = torch.ones(h_index.shape[0], self.dims[0], device=dev)
        boundary = torch.zeros(graph.num_node, query.shape[1], device=dev)
        boundary[h_index] = 1.0
        graph.query = query.unsqueeze(0)
        graph.boundary = boundary.unsqueeze(1)
        graph.boundary_sparse = None
        return {"node_feature": self._run_layers(graph, boundary.unsqueeze(1)).squeeze(1)}

    def forward(self, graph, h_index, t_index=None, r_index=None, all_loss=None, metric=None):
        if not graph.num_relation:
            relation = torch.zeros(graph.num_edge, 1, dtype=torch.long, device=graph.device)
            graph = Graph(torch.cat([graph.edge_list[:, :2], relation], dim=-1), graph.edge_weight, graph.num_node, 1)
        return self.bellmanford(graph, h_index)["node_feature"]


class CustomNBFNetFull(CustomNBFNet):
    """``rel_model.py:343-378``: every query relation gets its own labelled copy -> ``(batch, 2R, dim)``."""

    def __init__(self, learn_query=False, **kwargs):
        super().__init__(**kwargs)
        self.learn_query = learn_query
        if learn_query:
            self.learnable_q = nn.Embedding(1, self.dims[0])

    def _fast_bellmanford(self, graph, h_index):
        """Inference on the HIP backend with the shipped relation model (64-d DistMult / sum layers over the 4 edge types,
        shared relation embeddings, LayerNorm, shortcut): the sequence :meth:`bellmanford` runs through the layers, with
        the boundary never materialised (frontier kernel + sparse-input epilogue) and the relation tables of all layers
        tiled by one copy.  Same kernels on the same operands: identical bits.  ``None``: not applicable."""
        F = torch.nn.functional
        ops = layer.backend.get()
        if (not getattr(ops, "FAST_INFERENCE", False) or torch.is_grad_enabled() or self.learn_query or self.concat_hidden
                or not layer.FRONTIER_FIRST_LAYER or not ops.accepts(h_index) or not self.layers
                or graph.requires_grad):                   # layer.py:299: such a graph takes message + aggregate
            return None
        for conv in self.layers:
            ok = (isinstance(conv, layer.GeneralizedRelationalConvNBF) and not conv.dependent and conv.message_func == "distmult"
                  and conv.aggregate_func == "sum" and conv.input_dim == 64 and conv.output_dim == 64
                  and tuple(conv.linear.weight.shape) == (64, 128) and conv.linear.bias is not None
                  and (conv.activation is F.relu or not conv.activation) and conv.relation.weight.is_cuda
                  and conv.relation.weight.dtype == torch.float32 and conv.linear.weight.dtype == torch.float32
                  and (conv.layer_norm is None or (conv.layer_norm.elementwise_affine and conv.layer_norm.bias is not None)))
            if not ok:
                return None
            assert graph.num_relation == conv.num_relation              # layer.py:53,115
        n_query = h_index.shape[0]
        if not ops.frontier_supported("add", "mul", 64 * n_query):
            return None
        # (nothing is cached across calls: a captured hipGraph holds raw pointers to every tensor it read, and a cache
        # rebuilt for another batch size -- or stale after a weight update -- would silently invalidate it)
        weights = [conv.relation.weight for conv in self.layers]
        if len(weights) > 8 or self.dims[0] != 64 or not n_query:
            return None
        tables, query, node32 = ops.relation_stack_inputs(weights, h_index)      # tiled tables, ones, int32 nodes: one launch
        if not layer._frontier_tables_finite(tables[0]):
            return None
        boundary = (node32, query)
        csr = graph.relcsr
        n_node = graph.num_node
        hidden = None
        for i, conv in enumerate(self.layers):
            ln = conv.layer_norm
            args = (conv.linear.weight, conv.linear.bias, ln.weight if ln else None, ln.bias if ln else None,
                    ln.eps if ln else 1e-5, conv.activation is F.relu, self.short_cut)
            if i == 0:
                update = ops.rspmm_frontier(csr, tables[0], boundary).view(n_node, n_query, 64)
                hidden = ops.combine_forward(None, update, *args, reuse_update=True, input_boundary=boundary)
            else:
                # dense relation graphs: the layer as one launch on the matrix cores (csrc/relgraph_dense.hip); same bits
                fused = ops.dense_layer_forward(csr, tables[i], hidden, boundary, *args)
                if fused is not None:
                    hidden = fused
                    continue
                update = ops.rspmm_forward(csr, tables[i], hidden.flatten(1), "add", "mul", boundary=boundary)
                hidden = ops.combine_forward(hidden, update.view(n_node, n_query, 64), *args, reuse_update=True)
        return {"node_feature": hidden.transpose(1, 0)}

    def bellmanford(self, graph, h_index, separate_grad=False):
        fast = self._fast_bellmanford(graph, h_index)
        if fast is not None:
            return fast
        dev = h_index.device
        if self.learn_query:
            query = self.learnable_q.weight.expand(h_index.shape[0], self.dims[0])
        else:
            query = torch.ones(h_index.shape[0], self.dims[0], device=dev)
        index = h_index.unsqueeze(-1).expand_as(query)
        boundary = torch.zeros(graph.num_node, *query.shape, device=dev)
        boundary.scatter_add_(0, index.unsqueeze(0), query.unsqueeze(0))
        graph.query = query
        graph.boundary = boundary
        graph.boundary_sparse = (h_index.to(torch.int32), query)
        return {"node_feature": self._run_layers(graph, boundary).transpose(1, 0)}


class RelNBFNet(nn.Module):
    """``rel_model.py:381-416`` (+ the ``RelationModel`` base, ``:54-89``).  ``forward(graph, None, r_idx)``."""

    def __init__(self, input_dim, hidden, num_layers=6, input_type="ones", num_relation=None, **kwargs):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden
        self.input_type = input_type
        self.num_relation = num_relation
        self.ablation_etypes = kwargs.get("ablation_etypes", False)
        self.model = CustomNBFNetFull(input_dim=input_dim, hidden_dims=[hidden] * num_layers,
                                      num_relation=4 if not self.ablation_etypes else None, aggregate_func="sum",
                                      layer_norm=True, short_cut=True, learn_query=kwargs.get("learn_query", False))
        if self.hidden_dim != self.input_dim:
            self.input_transform_linear = nn.Linear(self.input_dim, self.hidden_dim)

    def construct_relation_graph(self, graph):
        return construct_relation_graph(graph)

    def forward(self, graph, input, r_idx, all_loss=None, metric=None):
        assert input is None                                    # rel_model.py:21
        x = self.model(graph, h_index=r_idx)
        return {"graph_feature": None, "node_feature": x}      # (batch, 2R, dim)


class RelationModelList(nn.ModuleList):
    """``rel_model.py:209-223``: ``num_rel_models`` copies built from the ``rel_model`` config dict."""

    def __init__(self, num_rel_models=1, num_relation=None, rel_model=None, **kwargs):
        super().__init__()
        self.num_rel_models = num_rel_models
        self.num_relation = num_relation
        cfg = dict(rel_model or {})
        cls = cfg.pop("class_str", "RelNBFNet")
        if cls != "RelNBFNet":
            raise ValueError("only RelNBFNet relation models are shipped, got `%s`" % cls)
        for _ in range(num_rel_models):
            self.append(RelNBFNet(**cfg, num_relation=num_relation))

"""Test infrastructure: run the package's model stack through the REFERENCE'S OWN O(E) definition of the operator, in ATen ops,
on the GPU -- no HIP kernel of this package and no C oracle anywhere in the computation.

``generalized_rspmm`` is torchdrug's, but the reference carries its definition as plain tensor code: ``message`` + ``aggregate``
(``/root/reference/ultra/layer.py:232-296``, twin ``:52-109``), the branch its layers take when ``graph.requires_grad``
(``:299``).  The package's layers mirror that branch (``layer.message`` / ``layer.aggregate``).  :func:`aten_definition`
routes EVERYTHING there:

* an operator backend whose ``accepts()`` is False, so every fused path of the package (epilogue, projections, score head,
  frontier, candidate tiles, fused loss, device ranking) declines and the callers run the reference's ATen chains
  (``combine`` = cat + Linear + LayerNorm + relu, ``nn.Linear`` MLPs, dense filter masks, ``F.binary_cross_entropy_with_logits``);
  it has no rspmm at all -- a layer that tried to call one would fail loudly;
* ``requires_grad = True`` on the graphs the layers see (entity graph with inverse edges, relation graphs), which is the
  reference's own switch to ``message`` + ``aggregate``;
* edge removal as the reference does it: a NEW graph without the batch's edges (``ultra/model.py:57-74``).

Run it in fp32 and in fp64 (``double=True``: parameters and default dtype): the fp64 run is the truth, and the HIP path must be
as close to it as the fp32 ATen run is."""
import contextlib

import torch

from ultra_torchdrug_amd import backend
from ultra_torchdrug_amd.graph import Graph


class AtenDefinition:
    """Backend interface of ``ultra_torchdrug_amd/backend.py`` that computes nothing itself."""
    FAST_INFERENCE = False

    @staticmethod
    def accepts(tensor):
        return False

    @staticmethod
    def score_candidates_supported(*args):
        return False

    @staticmethod
    def candidate_tiles(*args, **kwargs):
        return None

    @staticmethod
    def candidate_rows(*args, **kwargs):
        return None

    @staticmethod
    def remove_triples(graph, h, t, r, n_base_rel):
        """``remove_easy_edges`` on the graph with inverse edges (``ultra/model.py:57-74,166``): a new, smaller graph."""
        n, rels = graph.num_node, graph.num_relation
        h, t, r = h.reshape(-1), t.reshape(-1), r.reshape(-1)
        gone = torch.cat([(h * n + t) * rels + r, (t * n + h) * rels + r + n_base_rel])
        e = graph.edge_list
        keep = ~torch.isin((e[:, 0] * n + e[:, 1]) * rels + e[:, 2], gone)
        out = Graph(e[keep], graph.edge_weight[keep], n, rels)
        out.requires_grad = True
        return out


@contextlib.contextmanager
def aten_definition(task, double=False):
    """``task`` computes through the ATen definition inside the context (see the module docstring).  ``double``: the
    parameters are converted to fp64 for the duration and the default dtype is fp64 (``torch.ones`` / ``zeros`` of the
    boundary conditions); they are restored to the SAME fp32 values afterwards."""
    graphs = []
    for ctx in task.contexts.values():
        graphs.append(task.model._undirected(ctx["fact_graph"]))
        graphs.extend(ctx["rel_graphs"])
    saved_flags = [g.requires_grad for g in graphs]
    saved_params = [p.detach().clone() for p in task.parameters()] if double else None
    saved_dtype = torch.get_default_dtype()
    try:
        for g in graphs:
            g.requires_grad = True
        if double:
            task.double()
            torch.set_default_dtype(torch.float64)
        with backend.use(AtenDefinition()):
            yield task
    finally:
        torch.set_default_dtype(saved_dtype)
        for g, flag in zip(graphs, saved_flags):
            g.requires_grad = flag
        if double:
            task.float()
            with torch.no_grad():
                for p, keep in zip(task.parameters(), saved_params):
                    p.copy_(keep)

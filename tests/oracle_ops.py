"""Test infrastructure: run the package's model stack on CPU with every operator replaced by the CPU oracle.

The product's operators refuse CPU tensors (no fallback).  For end-to-end parity tests an oracle-backed object with
the complete backend interface (``ultra_torchdrug_amd/backend.py``) is installed with ``backend.use``, so the SAME
Python model code yields the CPU-oracle numbers that the HIP path is compared with.
"""
import contextlib

import numpy as np
import torch

from oracle import oracle as O


class _OracleRSPMM(torch.autograd.Function):

    @staticmethod
    def forward(ctx, relation, input, csr_o, sum, mul, piece, dense_relation=False):
        out = O.rspmm_forward(csr_o, relation.detach().numpy(), input.detach().numpy(), sum, mul, piece=piece)
        ctx.csr_o, ctx.sum, ctx.mul, ctx.piece, ctx.dense_relation = csr_o, sum, mul, piece, dense_relation
        out = torch.from_numpy(out)
        ctx.save_for_backward(relation, input, out)
        return out

    @staticmethod
    def backward(ctx, grad):
        relation, input, out = ctx.saved_tensors
        d_rel, d_x = O.rspmm_backward(ctx.csr_o, relation.detach().numpy(), input.detach().numpy(), out.numpy(),
                                      grad.contiguous().numpy(), ctx.sum, ctx.mul, piece=ctx.piece,
                                      dense_relation=ctx.dense_relation)
        return torch.from_numpy(d_rel), torch.from_numpy(d_x), None, None, None, None, None


class OracleFunctional:
    """The backend interface of ``ultra_torchdrug_amd/backend.py`` on CPU tensors, every operator through the oracle
    (C restatement) or the reference's own torch formulation."""

    def __init__(self, piece):
        self.piece = piece
        self._cache = {}

    def _csr(self, relcsr):
        key = id(relcsr)
        if key not in self._cache:
            n_dst, n_src, n_rel = relcsr.shape
            w = None if relcsr.unit_weight else relcsr.weight.cpu().numpy()
            self._cache[key] = (relcsr, O.coalesce_csr(relcsr.dst.cpu().numpy(), relcsr.src.cpu().numpy(),
                                                       relcsr.rel_id.cpu().numpy(), w, n_dst, n_src, n_rel))
        return self._cache[key][1]

    @staticmethod
    def accepts(tensor):
        return not tensor.is_cuda

    def combine(self, input, update, weight, bias, ln_weight=None, ln_bias=None, ln_eps=1e-5, relu=True, shortcut=False,
                reuse_update=False):
        """Layer epilogue on the CPU: the oracle's C restatement (the HIP kernel's documented order) for inference,
        the reference's own torch chain (layer.py:386-392, model.py:126-127) when autograd is needed."""
        tensors = [t for t in (input, update, weight, bias, ln_weight, ln_bias) if t is not None]
        if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
            out = torch.nn.functional.linear(torch.cat([input, update], dim=-1), weight, bias)
            if ln_weight is not None:
                out = torch.nn.functional.layer_norm(out, (64,), ln_weight, ln_bias, ln_eps)
            if relu:
                out = torch.relu(out)
            return out + input if shortcut else out
        out = O.combine_forward(input.detach().numpy(), update.detach().numpy(), weight.detach().numpy(),
                                bias.detach().numpy(), None if ln_weight is None else ln_weight.detach().numpy(),
                                None if ln_bias is None else ln_bias.detach().numpy(), ln_eps, relu, shortcut)
        return torch.from_numpy(out).view_as(input)

    @staticmethod
    def linear_supported(in_dim, out_dim):
        return (in_dim, out_dim) in ((64, 64), (128, 128)) or (out_dim == 1 and in_dim % 4 == 0)

    def linear_forward(self, input, weight, bias, relu=False):
        out = O.linear_forward(input.detach().numpy(), weight.detach().numpy(), bias.detach().numpy(), relu)
        return torch.from_numpy(out)

    def score_all_entities(self, hidden, query, w1, b1, w2, b2):
        """(N, B, 64), (B, 64) -> (B, N): the reference's cat / transpose / mlp chain (model.py:134-138,177-193) in the
        documented order (the queries' half of the first layer once per query: oracle.score_head_forward)."""
        out = O.score_head_forward(hidden.detach().numpy(), query.detach().numpy(), w1.detach().numpy(), b1.detach().numpy(),
                                   w2.detach().numpy(), b2.detach().numpy())
        return torch.from_numpy(out)

    @staticmethod
    def bce_adversarial_loss(pred, temperature):
        """task.py:169-180 as the reference writes it (ATen ops, autograd)."""
        target = torch.zeros_like(pred)
        target[:, 0] = 1
        loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, target, reduction="none")
        neg_weight = torch.ones_like(pred)
        if temperature > 0:
            with torch.no_grad():
                neg_weight[:, 1:] = torch.nn.functional.softmax(pred[:, 1:] / temperature, dim=-1)
        else:
            neg_weight[:, 1:] = 1 / (pred.shape[1] - 1)
        return (loss * neg_weight).sum(dim=-1) / neg_weight.sum(dim=-1)

    @staticmethod
    def score_candidates_supported(hidden, query, t_index, w1, w2):
        return False                                    # the oracle path scores through the model's own mlp (ATen)

    @staticmethod
    def score_candidates(hidden, query, t_index, w1, b1, w2, b2):
        """model.py:177-183,193 as the reference writes it."""
        rows = torch.arange(hidden.shape[1], device=hidden.device).unsqueeze(-1)
        feature = torch.cat([hidden[t_index, rows], query.unsqueeze(1).expand(-1, t_index.shape[1], -1)], dim=-1)
        hid = torch.relu(torch.nn.functional.linear(feature, w1, b1))
        return torch.nn.functional.linear(hid, w2.view(1, -1), b2).squeeze(-1)

    @staticmethod
    def candidate_tiles(t_index, n_query, n_node=None):
        return None                                     # a hint for the HIP epilogue's backward; the oracle path computes every row

    @staticmethod
    def candidate_rows(t_index, n_node):
        return None                                     # likewise: a hint for the HIP rspmm backward

    @staticmethod
    def statistics(values, repeated=None, repeat=0):
        """(norm, mean, std) of `values` with `repeated` taken `repeat` times, as the reference computes them: ATen
        reductions over the materialised tensor (model.py:158-160,178-181)."""
        flat = values.detach().reshape(-1)
        if repeated is not None:
            flat = torch.cat([flat, repeated.detach().reshape(1, -1).expand(int(repeat), -1).reshape(-1)])
        return torch.stack([flat.norm(), flat.mean(), flat.std()])

    def generalized_rspmm(self, sparse, relation, input, sum="add", mul="mul"):
        # None: the order the library uses for this adjacency and call (each plan's own piece length; the reference order and
        # the documented d_relation order where the plans carry their dense form)
        piece, dense_relation = sparse.kernel_order(sum, mul, input.shape[1]) if self.piece is None else (self.piece, False)
        return _OracleRSPMM.apply(relation.contiguous(), input.contiguous(), self._csr(sparse), sum, mul, piece, dense_relation)

    @staticmethod
    def _dense_boundary(boundary, n_rows):
        """(node (B,), value (B, D)) -> the (N, B * D) tensor scatter_add_ builds in ultra/model.py:106-107."""
        node, value = boundary
        dense = torch.zeros(n_rows, *value.shape, dtype=value.dtype)
        index = node.long().view(1, -1, 1).expand(1, *value.shape)
        return dense.scatter_add(0, index, value.unsqueeze(0)).flatten(1)

    def rspmm_forward(self, csr, relation, input, sum="add", mul="mul", add_rows=None, boundary=None):
        """The fused boundary epilogue of the HIP kernels, unfused: layer.py:156,162,358,364 as written."""
        out = self.generalized_rspmm(csr, relation, input, sum=sum, mul=mul)
        if boundary is not None:
            add_rows = self._dense_boundary(boundary, out.shape[0])
        if add_rows is None:
            return out
        return out + add_rows if sum == "add" else (torch.max(out, add_rows) if sum == "max" else torch.min(out, add_rows))

    def rspmm_sum_plus(self, sparse, relation, input, add_rows, mul="mul", boundary=None):
        out = self.generalized_rspmm(sparse, relation, input, sum="add", mul=mul)
        return out + (add_rows if boundary is None else self._dense_boundary(boundary, out.shape[0]))

    def sum_layer(self, csr, relation, input, boundary_dense, boundary_sparse, mul, weight, bias, ln_weight=None,
                  ln_bias=None, ln_eps=1e-5, relu=True, shortcut=False, input_is_boundary=False, grad_tiles=None,
                  grad_rows=None):
        """The layer the reference's way, op by op (layer.py:298-392, model.py:126-127)."""
        shape = input.shape
        if boundary_sparse is not None:
            update = self.rspmm_sum_plus(csr, relation, input.flatten(1), None, mul=mul, boundary=boundary_sparse)
        else:
            update = self.rspmm_sum_plus(csr, relation, input.flatten(1), boundary_dense.flatten(1), mul=mul)
        return self.combine(input, update.view(shape), weight, bias, ln_weight, ln_bias, ln_eps, relu, shortcut)

    @staticmethod
    def frontier_supported(sum, mul, F):
        return sum == "add" and mul == "mul" and F % 64 == 0

    def rspmm_frontier(self, csr, relation, boundary):
        """The first layer the long way round: the dense boundary as input (what the reference does)."""
        dense = self._dense_boundary(boundary, csr.shape[1])
        return self.rspmm_forward(csr, relation, dense, "add", "mul", boundary=boundary)

    def relation_project(self, relation, weights):
        """layer.py:318-319,325-326 per layer: two linear layers and the (B, R, D) -> (R, B * D) transpose."""
        outs = []
        for w1, b1, w2, b2 in weights:
            hidden = self.linear_forward(relation.contiguous(), w1, b1, relu=True)
            outs.append(self.linear_forward(hidden, w2, b2).transpose(0, 1).flatten(1).contiguous())
        return outs

    @staticmethod
    def relation_project_train(relation, weights):
        """The reference's own chain with autograd (layer.py:318-319,325-326): the yardstick of the one-launch backward."""
        import torch.nn.functional as F
        return [F.linear(F.relu(F.linear(relation, w1, b1)), w2, b2).transpose(0, 1).flatten(1) for w1, b1, w2, b2 in weights]

    @staticmethod
    def remove_triples(graph, h, t, r, n_base_rel):
        """model.py:57-74 + :166 on the graph with inverse edges: weight 0 for (h, t, r) and (t, h, r + R)."""
        n, rels = graph.num_node, graph.num_relation
        key = lambda a, b, c: (a * n + b) * rels + c
        gone = torch.cat([key(h.reshape(-1), t.reshape(-1), r.reshape(-1)),
                          key(t.reshape(-1), h.reshape(-1), r.reshape(-1) + n_base_rel)])
        e = graph.edge_list
        keep = ~torch.isin(key(e[:, 0], e[:, 1], e[:, 2]), gone)
        return graph.reweighted(graph.edge_weight * keep)

    @staticmethod
    def strict_negatives(keys, anchor, rel, n_rel, n_node, rand):
        """task.py:102-118 literally: dense mask, nonzero, variadic_sample with the given uniform numbers."""
        rows = len(anchor)
        mask = torch.ones(rows, n_node, dtype=torch.bool)
        base = (anchor * n_rel + rel) * n_node
        lo, hi = torch.searchsorted(keys, base), torch.searchsorted(keys, base + n_node)
        for q in range(rows):
            mask[q, keys[lo[q]:hi[q]] - base[q]] = False
        candidates, sizes = mask.nonzero()[:, 1], mask.sum(dim=-1)
        index = (rand * sizes.unsqueeze(-1)).long()
        index = torch.minimum(index, (sizes - 1).unsqueeze(-1)) + (sizes.cumsum(0) - sizes).unsqueeze(-1)
        return candidates[index]

    @staticmethod
    def filtered_rank_keys(pred, target, keys, anchor, rel, n_rel, n_node=None):
        """task.py:307-315 with the dense mask of task.py:65-100 rebuilt from the keys."""
        assert n_node is None or n_node == pred.shape[1]
        rows, n_node = pred.shape
        mask = np.ones((rows, n_node), dtype=bool)
        if keys is not None:
            base = (anchor * n_rel + rel) * n_node
            lo, hi = torch.searchsorted(keys, base), torch.searchsorted(keys, base + n_node)
            for q in range(rows):
                mask[q, (keys[lo[q]:hi[q]] - base[q]).numpy()] = False
        return torch.from_numpy(O.filtered_rank(pred.contiguous().numpy(), mask, target.numpy()))


@contextlib.contextmanager
def oracle_rspmm(piece=None):
    """Inside the context the layers aggregate with the CPU oracle: ``piece=None`` -> the kernels' summation order
    (every RelCSR's own ``piece_len``), ``piece=0`` -> the reference's strictly sequential order."""
    from ultra_torchdrug_amd import backend
    with backend.use(OracleFunctional(piece)) as ops:
        yield ops

"""Test infrastructure: run the package's model stack on CPU with the rspmm operator replaced by the CPU oracle.

The product's ``generalized_rspmm`` refuses CPU tensors (no fallback).  For end-to-end parity tests the layers'
``functional`` module attribute is swapped for this oracle-backed operator, so the SAME Python model code yields
the CPU-oracle scores that the HIP path is compared with.
"""
import contextlib

import numpy as np
import torch

from oracle import oracle as O


class _OracleRSPMM(torch.autograd.Function):

    @staticmethod
    def forward(ctx, relation, input, csr_o, sum, mul, piece):
        out = O.rspmm_forward(csr_o, relation.detach().numpy(), input.detach().numpy(), sum, mul, piece=piece)
        ctx.csr_o, ctx.sum, ctx.mul, ctx.piece = csr_o, sum, mul, piece
        out = torch.from_numpy(out)
        ctx.save_for_backward(relation, input, out)
        return out

    @staticmethod
    def backward(ctx, grad):
        relation, input, out = ctx.saved_tensors
        d_rel, d_x = O.rspmm_backward(ctx.csr_o, relation.detach().numpy(), input.detach().numpy(), out.numpy(),
                                      grad.contiguous().numpy(), ctx.sum, ctx.mul, piece=ctx.piece)
        return torch.from_numpy(d_rel), torch.from_numpy(d_x), None, None, None, None


class OracleFunctional:
    """Drop-in for ``ultra_torchdrug_amd.functional`` inside ``layer.py`` (only ``generalized_rspmm`` is used)."""

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

    cpu_ok = True      # lets the layers route CPU tensors to `combine` below (the product itself is GPU-only)

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
        """(N, B, 64), (B, 64) -> (B, N): the reference's cat / transpose / mlp chain (model.py:134-138,177-193),
        each nn.Linear in the documented order."""
        n_node, batch, _ = hidden.shape
        feature = torch.cat([hidden, query.expand(n_node, -1, -1)], dim=-1).transpose(0, 1)      # (B, N, 128)
        h = self.linear_forward(feature.contiguous(), w1, b1, relu=True)
        return self.linear_forward(h, w2, b2, relu=False).squeeze(-1)

    def generalized_rspmm(self, sparse, relation, input, sum="add", mul="mul"):
        piece = sparse.piece_len if self.piece is None else self.piece       # None: each plan's own piece length
        return _OracleRSPMM.apply(relation.contiguous(), input.contiguous(), self._csr(sparse), sum, mul, piece)


@contextlib.contextmanager
def oracle_rspmm(piece=None):
    """Inside the context the layers aggregate with the CPU oracle: ``piece=None`` -> the kernels' summation order
    (every RelCSR's own ``piece_len``), ``piece=0`` -> the reference's strictly sequential order."""
    from ultra_torchdrug_amd import layer
    saved = layer.functional
    layer.functional = OracleFunctional(piece)
    try:
        yield
    finally:
        layer.functional = saved

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

    def generalized_rspmm(self, sparse, relation, input, sum="add", mul="mul"):
        return _OracleRSPMM.apply(relation.contiguous(), input.contiguous(), self._csr(sparse), sum, mul, self.piece)


@contextlib.contextmanager
def oracle_rspmm(piece):
    """Inside the context the layers aggregate with the CPU oracle (kernel summation order when piece > 0)."""
    from ultra_torchdrug_amd import layer
    saved = layer.functional
    layer.functional = OracleFunctional(piece)
    try:
        yield
    finally:
        layer.functional = saved

"""The operator backend the layers, models and the task call.

The product has exactly ONE backend: :mod:`ultra_torchdrug_amd.functional`, i.e. ``libultra_rspmm.so`` on an MI355X
(there is no CPU / PyTorch fallback behind it -- CPU tensors make its operators raise).  The callers never probe for
optional functions: a backend is an object with this complete interface

    accepts(tensor) -> bool            tensors this backend computes on (HIP library: ``tensor.is_cuda``)
    generalized_rspmm, rspmm_forward, rspmm_sum_plus, rspmm_frontier, frontier_supported, first_layer_forward, sum_layer,
    remove_triples
    combine, linear_supported, linear_forward, relation_project, relation_project_train, score_all_entities
    filtered_rank, filtered_rank_keys, strict_negatives, statistics, bce_adversarial_loss, candidate_tiles, candidate_rows,
    score_candidates_supported, score_candidates

The parity tests install a second implementation of the same interface (``tests/oracle_ops.py``: the CPU oracle
behind every operator) with :func:`use`, so that the SAME model code yields the oracle-side numbers; nothing under
``oracle/`` is imported from the package.
"""
import contextlib

from . import functional as _hip

_active = _hip


def get():
    """The backend in force (the HIP library unless a test installed another one)."""
    return _active


@contextlib.contextmanager
def use(backend):
    """Install ``backend`` for the duration of the context (test infrastructure)."""
    global _active
    saved = _active
    _active = backend
    try:
        yield backend
    finally:
        _active = saved

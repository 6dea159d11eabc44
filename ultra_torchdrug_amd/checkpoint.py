"""Loading / saving in the reference's checkpoint layout (``/root/reference/ultra/util.py:233-325``).

A reference checkpoint is ``{"model": task.state_dict(), "optimizer": ...}`` written by ``clean_save`` -- the task's
state dict minus the non-tensor graph buffers -- and read back by ``safe_load`` with ``strict=False``.  The module tree
of this package uses the same parameter names (``model.layers.{i}.linear.weight`` ...,
``rel_models.0.model.layers.{i}.relation.weight`` ...), so ``td_ultra_3g.pth`` / ``td_ultra_4g.pth``
load unchanged.  Graph objects that an older checkpoint may still carry are dropped exactly as ``safe_load`` does.
"""
import os

import torch

# util.py:241-244 -- entries of the model dict that are graphs, not tensors
GRAPH_KEYS = ("fact_graph", "train_graph", "valid_graph", "test_graph", "train_rel_graph", "valid_rel_graph",
              "test_rel_graph", "graph", "inductive_graph", "train_rel_graphs", "valid_rel_graphs", "test_rel_graphs",
              "rel_graphs")
# util.py:250-252 -- relation-specific weights kept from the current model when fix_reasoner is set
_REASONER_MARKS = ("relation.weight", "relation_projection", "relation_linear", "query.weight")


def load_checkpoint(task, checkpoint, fix_reasoner=False, optimizer=None, map_location=None):
    """``util.safe_load``: returns the ``(missing_keys, unexpected_keys)`` of the non-strict load."""
    if isinstance(checkpoint, (str, os.PathLike)):
        state = torch.load(os.path.expanduser(checkpoint), map_location=map_location or task.device,
                           weights_only=False)
    else:
        state = checkpoint
    model_state = dict(state["model"])
    for key in GRAPH_KEYS:
        if key in model_state and not torch.is_tensor(model_state[key]):
            model_state.pop(key)
    if fix_reasoner:
        current = task.state_dict()
        for key in [k for k in model_state if any(mark in k for mark in _REASONER_MARKS)]:
            model_state.pop(key)
            if key in current:
                model_state[key] = current[key]
    result = task.load_state_dict(model_state, strict=False)
    if optimizer is not None and not fix_reasoner and "optimizer" in state:
        try:
            optimizer.load_state_dict(state["optimizer"])
        except ValueError:
            print("warning: loaded optimizer state has a different number of parameter groups")
    return result.missing_keys, result.unexpected_keys


def save_checkpoint(task, checkpoint, optimizer=None):
    """``util.clean_save``: tensors only (graphs are plain attributes here, never part of the state dict)."""
    state = {"model": {k: v.detach().cpu() for k, v in task.state_dict().items()},
             "optimizer": optimizer.state_dict() if optimizer is not None else None}
    torch.save(state, os.path.expanduser(checkpoint))
    return state

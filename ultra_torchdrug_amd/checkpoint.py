"""Loading / saving in the reference's checkpoint layout (``/root/reference/ultra/util.py:233-325``).

A reference checkpoint is ``{"model": task.state_dict(), "optimizer": ...}`` written by ``clean_save`` -- the task's
state dict minus the non-tensor graph buffers -- and read back by ``safe_load`` with ``strict=False``.  The module tree
of this package uses the same parameter names (``model.layers.{i}.linear.weight`` ...,
``rel_models.0.model.layers.{i}.relation.weight`` ...), so ``td_ultra_3g.pth`` / ``td_ultra_4g.pth``
load unchanged.  Graph objects that an older checkpoint may still carry are dropped exactly as ``safe_load`` does.
"""
import os
import pickle

import torch

# util.py:241-244 -- entries of the model dict that are graphs, not tensors
GRAPH_KEYS = ("fact_graph", "train_graph", "valid_graph", "test_graph", "train_rel_graph", "valid_rel_graph",
              "test_rel_graph", "graph", "inductive_graph", "train_rel_graphs", "valid_rel_graphs", "test_rel_graphs",
              "rel_graphs")
# util.py:250-252 -- relation-specific weights kept from the current model when fix_reasoner is set
_REASONER_MARKS = ("relation.weight", "relation_projection", "relation_linear", "query.weight")


class _Placeholder:
    """Stands in for a pickled object of a package that is not installed (torchdrug ``Graph`` / ``PackedGraph``
    buffers of an un-cleaned checkpoint); such entries are dropped right after loading."""

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        pass


_STORAGES = ("FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage", "ShortStorage",
             "CharStorage", "ByteStorage", "BoolStorage", "ComplexFloatStorage", "ComplexDoubleStorage")
_DTYPES = ("float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool", "complex64",
           "complex128")
# every global a tensors-only checkpoint (state dict + optimizer state) can refer to -- (module, name) pairs, nothing
# resolved by package root: ``torch.*`` / ``numpy.*`` / ``builtins.*`` as roots also hold callables that run commands
# (``torch.utils.collect_env.run``, ``numpy.testing._private.utils.runstring``, ``builtins.breakpoint`` ...)
_ALLOWED_GLOBALS = frozenset(
    [("collections", "OrderedDict"), ("collections", "defaultdict"),
     ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
     ("torch", "Size"), ("torch", "device"), ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"),
     ("numpy", "ndarray"), ("numpy", "dtype"),
     ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
     ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
     ("_codecs", "encode")]
    + [("torch", name) for name in _STORAGES + _DTYPES]
    + [("builtins", name) for name in ("set", "frozenset", "list", "dict", "tuple", "int", "float", "bool", "str", "bytes",
                                       "bytearray", "complex", "slice", "range")])


class _TensorsOnlyUnpickler(pickle.Unpickler):
    """Resolves exactly the globals of ``_ALLOWED_GLOBALS``; every ``torchdrug.*`` global becomes an inert placeholder
    and anything else is refused -- a checkpoint is data, not code."""

    def find_class(self, module, name):
        if module.split(".")[0] == "torchdrug":
            return _Placeholder
        if (module, name) not in _ALLOWED_GLOBALS:
            raise pickle.UnpicklingError("checkpoint refers to %s.%s, which a tensors-only checkpoint never needs"
                                         % (module, name))
        return super().find_class(module, name)


class _tensors_only_pickle:
    """``pickle_module`` for ``torch.load``: same interface as ``pickle`` with the restricted unpickler."""
    __name__ = "pickle"
    Unpickler = _TensorsOnlyUnpickler
    load = staticmethod(lambda f, **kw: _TensorsOnlyUnpickler(f, **kw).load())


def read_checkpoint(path, map_location=None):
    """``torch.load`` of a reference checkpoint as DATA.  First ``weights_only=True`` (what ``util.clean_save`` writes,
    ``ultra/util.py:278-325``).  A file that still carries torchdrug graph objects (``util.py:241-247`` drops them
    AFTER unpickling, which needs torchdrug installed) makes that loader refuse the ``torchdrug.*`` global; such a file
    -- and only an ``UnpicklingError``, with a warning -- is re-read with an unpickler that resolves an explicit list of
    ``(module, name)`` globals (tensor / storage rebuilders, ``OrderedDict``, numpy array rebuilders, plain builtin
    containers), maps ``torchdrug.*`` classes to inert placeholders and refuses everything else.  Neither reader can be
    made to call a function the file names."""
    path = os.path.expanduser(path)
    try:
        return torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError:
        import warnings
        warnings.warn("%s is not a tensors-only file for torch.load(weights_only=True); re-reading it with the "
                      "allow-listed unpickler (torchdrug objects become placeholders, other globals are refused)" % path)
        return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_tensors_only_pickle)


def load_checkpoint(task, checkpoint, fix_reasoner=False, optimizer=None, map_location=None):
    """``util.safe_load``: returns the ``(missing_keys, unexpected_keys)`` of the non-strict load."""
    if isinstance(checkpoint, (str, os.PathLike)):
        state = read_checkpoint(checkpoint, map_location=map_location or task.device)
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

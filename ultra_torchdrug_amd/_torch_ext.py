"""Loader of ``libultra_torch_ext.so`` -- the PyTorch-ROCm C++ extension that registers ``torch.ops.ultra_mi.*``
(``csrc/torch_ext.cpp``): the form of the operator boundary the reference itself uses (torchdrug JIT-builds a C++
extension and calls it through the dispatcher, ``/root/reference/README.md:43-45``).  The ctypes binding (``_lib.py``)
stays as the torch-free path onto the same C ABI; ``ULTRA_BINDING=ctypes`` selects it for the rspmm operators.
"""
import os

import torch

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
EXT_PATH = os.environ.get("ULTRA_TORCH_EXT") or os.path.join(_HERE, "libultra_torch_ext.so")
OPS = ("build_relcsr", "rspmm_fwd", "rspmm_bwd", "rspmm_plan_fwd", "rspmm_plan_bwd", "abi_version")
_loaded = None


def load():
    """``torch.ops.ultra_mi`` with the library loaded; raises :class:`_lib.UltraLibraryError` if it is absent."""
    global _loaded
    if _loaded is not None:
        return _loaded
    _lib.load()                                    # the C ABI library first (same file the extension links)
    if not os.path.exists(EXT_PATH):
        raise _lib.UltraLibraryError("%s not found: build it with `make -C ultra_torchdrug_amd/csrc` "
                                     "(hipcc against the torch headers)" % EXT_PATH)
    try:
        torch.ops.load_library(EXT_PATH)
    except OSError as err:
        raise _lib.UltraLibraryError("cannot load %s: %s" % (EXT_PATH, err)) from err
    ops = torch.ops.ultra_mi
    missing = [name for name in OPS if not hasattr(ops, name)]
    if missing:
        raise _lib.UltraLibraryError("%s does not register %s" % (EXT_PATH, missing))
    if int(ops.abi_version()) != _lib.ABI_VERSION:
        raise _lib.UltraLibraryError("ABI mismatch: torch extension %d, binding %d" % (int(ops.abi_version()), _lib.ABI_VERSION))
    _loaded = ops
    return ops


def binding():
    """"torch" (dispatcher ops of the C++ extension; default) or "ctypes" (``ULTRA_BINDING=ctypes``)."""
    choice = os.environ.get("ULTRA_BINDING", "torch")
    if choice not in ("torch", "ctypes"):
        raise ValueError("ULTRA_BINDING must be `torch` or `ctypes`, got `%s`" % choice)
    return choice


def available():
    return os.path.exists(EXT_PATH)

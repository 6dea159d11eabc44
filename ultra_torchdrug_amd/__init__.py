"""ultra_torchdrug_amd -- MI355X-native relational message passing (rspmm) for ULTRA / NBFNet.

The package holds only what the hot path of DeepGraphLearning/ultra_torchdrug needs:
``csrc/`` (HIP kernels + the C ABI of ``libultra_rspmm.so``), the ctypes binding, the RelCSR plan builder and the
host-side mirror of the reference's operator / layer / model interface for that path.
"""
from . import _lib
from .relcsr import RelCSR, Segments
from .functional import generalized_rspmm, rspmm_forward

__all__ = ["generalized_rspmm", "rspmm_forward", "RelCSR", "Segments", "library_path", "require_library"]
__version__ = "0.1.0"


def library_path():
    return _lib.LIB_PATH


def require_library():
    """Load ``libultra_rspmm.so`` -- and the PyTorch extension on top of it unless ``ULTRA_BINDING=ctypes`` -- now
    (raises if one is missing: there is no fallback path)."""
    lib = _lib.load()
    from . import _torch_ext
    if _torch_ext.binding() == "torch":
        _torch_ext.load()
    return lib

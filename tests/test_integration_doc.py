"""INTEGRATION.md 1b -- the ctypes stub a maintainer pastes -- held to the library, VERBATIM.

Round 5's stub declared ``struct ultra_segments`` one field short of the header (no ``packed_dead``); pasted, it handed the
library a struct 8 bytes too small and the forward read a garbage pointer past its end.  Three guards since ABI 8:
the struct carries ``struct_bytes`` / ``abi_version`` and every entry point refuses a foreign one (``ULTRA_ERR_ABI``);
``ultra_segments_bytes()`` lets a binding check itself at import; and these tests extract the code block from the document
and run it as it stands (CPU: the struct's size and the refusal; GPU: the forward against the oracle).
"""
import ctypes
import os
import re
import types

import numpy as np
import pytest
import torch

from graphs import random_graph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("### 1b."):]
    block = re.search(r"```python\n(.*?)```", section, flags=re.S)
    assert block, "INTEGRATION.md 1b lost its python block"
    return block.group(1)


def _run_stub():
    """exec the document's block as a module.  Its `ctypes.CDLL("libultra_rspmm.so")` finds the in-tree build because the
    package has loaded it already and the library carries that soname (csrc/Makefile)."""
    from ultra_torchdrug_amd import _lib
    _lib.load()
    module = types.ModuleType("integration_1b")
    exec(compile(_stub_source(), "INTEGRATION.md#1b", "exec"), module.__dict__)
    return module


def test_documented_struct_is_the_librarys_struct():
    from ultra_torchdrug_amd import _lib
    lib = _lib.load()
    stub = _run_stub()                                        # (its own assert compares ABI and size at import)
    assert ctypes.sizeof(stub._Segments) == lib.ultra_segments_bytes() == ctypes.sizeof(_lib.UltraSegments)
    assert [name for name, _ in stub._Segments._fields_] == [name for name, _ in _lib.UltraSegments._fields_]
    header = open(os.path.join(ROOT, "include", "ultra_rspmm.h")).read()
    body = header[header.index("typedef struct ultra_segments {"):header.index("} ultra_segments;")]
    declared = re.findall(r"^\s*(?:const\s+)?(?:u?int\d+_t|float)\s*\*?\s*(\w+);", body, flags=re.M)
    assert declared == [name for name, _ in stub._Segments._fields_], "header fields vs the document's struct"
    assert stub.ABI == _lib.ABI_VERSION


def test_a_struct_from_another_header_is_refused():
    """No GPU involved: the fence is checked before anything else is read or any HIP call is made."""
    from ultra_torchdrug_amd import _lib
    lib = _lib.load()
    ERR_ABI = 7
    assert b"struct_bytes" in lib.ultra_rspmm_status_string(ERR_ABI)

    class Round5Segments(ctypes.Structure):                  # INTEGRATION.md 1b as round 5 shipped it: no fence, no packed_dead
        _fields_ = [(name, kind) for name, kind in _lib.UltraSegments._fields_
                    if name not in ("struct_bytes", "abi_version", "packed_dead")]

    stale = Round5Segments()
    stale.n_rows, stale.n_edges = 4, 4                      # lands where the fence lives: a size nobody would declare
    as_plan = ctypes.cast(ctypes.pointer(stale), ctypes.POINTER(_lib.UltraSegments))
    assert lib.ultra_rspmm_workspace_bytes(as_plan, 64) == 0
    args = (None, None, None, None, None, 0, 4, 4, 64, 0, 0, None)
    assert lib.ultra_rspmm_forward_f32(as_plan, *args) == ERR_ABI
    # right size, wrong version; right version, wrong size
    seg = _lib.UltraSegments()
    assert seg.struct_bytes == ctypes.sizeof(_lib.UltraSegments) and seg.abi_version == _lib.ABI_VERSION
    seg.abi_version = _lib.ABI_VERSION - 1
    assert lib.ultra_rspmm_forward_f32(ctypes.byref(seg), *args) == ERR_ABI
    seg.abi_version, seg.struct_bytes = _lib.ABI_VERSION, ctypes.sizeof(_lib.UltraSegments) - 8
    assert lib.ultra_rspmm_forward_f32(ctypes.byref(seg), *args) == ERR_ABI
    assert lib.ultra_rspmm_backward_f32(ctypes.byref(seg), ctypes.byref(seg), None, None, None, None, None, None, None, 0,
                                        4, 4, 4, 64, 0, 0, None) in (ERR_ABI, 3)     # (NULL operands are refused first there)
    assert lib.ultra_dense_layer_supported(ctypes.byref(seg), 2) == 0
    # the genuine struct passes the fence (an empty plan: nothing to launch, no GPU touched)
    ok = _lib.UltraSegments()
    assert lib.ultra_rspmm_forward_f32(ctypes.byref(ok), None, None, None, None, None, 0, 0, 0, 64, 0, 0, None) == 0


@pytest.mark.gpu
def test_documented_stub_runs_verbatim_against_the_oracle(oracle):
    from ultra_torchdrug_amd import RelCSR
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    stub = _run_stub()
    n, r, F = 700, 9, 128
    g = random_graph(seed=21, n_node=n, n_edge=9000, n_rel=r)
    g["dst"][:1500] = 5                                        # a hub row: split into pieces (workspace in use)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    csr = RelCSR(t(g["dst"]), t(g["src"]), t(g["rel"]), None, n, n, r)
    rng = np.random.default_rng(3)
    relation = rng.standard_normal((r, F)).astype(np.float32)
    x = rng.standard_normal((n, F)).astype(np.float32)
    csr_o = oracle.coalesce_csr(g["dst"], g["src"], g["rel"], None, n, n, r)
    for sum_op, sum_name in enumerate(("add", "min", "max")):
        for mul_op, mul_name in enumerate(("mul", "add")):
            got = stub.rspmm_forward(csr.fwd, t(relation), t(x), sum_op, mul_op)
            torch.cuda.synchronize()
            want = oracle.rspmm_forward(csr_o, relation, x, sum_name, mul_name, piece=csr.piece_len)
            assert np.array_equal(got.cpu().numpy(), want), (sum_name, mul_name)
    # the stub's own struct, one field short, is refused with the ABI status (what round 5's document would have got)
    short = type("Short", (ctypes.Structure,), {"_fields_": stub._Segments._fields_[:-1]})()
    short.struct_bytes, short.abi_version = ctypes.sizeof(short), stub.ABI
    rc = stub._lib.ultra_rspmm_forward_f32(ctypes.byref(short), None, None, None, None, None, ctypes.c_size_t(0),
                                           ctypes.c_int64(n), ctypes.c_int64(r), ctypes.c_int64(F), 0, 0, None)
    assert rc == 7

"""ctypes binding of ``libultra_rspmm.so`` (C ABI: ``include/ultra_rspmm.h``).

The shared object is the product: there is no Python, PyTorch or CPU fallback behind it.  If it is missing or
cannot be loaded every operator of this package raises immediately.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ULTRA_RSPMM_LIB: load another build of the same ABI (kernel A/B runs, tools/kbench.py)
LIB_PATH = os.environ.get("ULTRA_RSPMM_LIB") or os.path.join(_HERE, "libultra_rspmm.so")
ABI_VERSION = 8

SUM_OPS = {"add": 0, "min": 1, "max": 2}
MUL_OPS = {"mul": 0, "add": 1}


class UltraSegments(ctypes.Structure):
    """``struct ultra_segments`` of include/ultra_rspmm.h (device pointers as integers).  A new instance carries the fence
    of ABI 8: ``struct_bytes`` = the size of THIS declaration, ``abi_version`` = the version this binding was written for;
    the library refuses a struct whose two leading fields are not its own (``ULTRA_ERR_ABI``)."""
    _fields_ = [
        ("struct_bytes", ctypes.c_uint32),
        ("abi_version", ctypes.c_uint32),
        ("n_rows", ctypes.c_int64),
        ("n_edges", ctypes.c_int64),
        ("row", ctypes.c_void_p),
        ("node_a", ctypes.c_void_p),
        ("node_b", ctypes.c_void_p),
        ("rel", ctypes.c_void_p),
        ("weight", ctypes.c_void_p),
        ("n_chunks", ctypes.c_int64),
        ("chunks", ctypes.c_void_p),
        ("n_long_rows", ctypes.c_int64),
        ("long_rows", ctypes.c_void_p),
        ("n_pieces", ctypes.c_int64),
        ("piece_len", ctypes.c_int64),
        ("packed", ctypes.c_void_p),
        ("packed_src_shift", ctypes.c_int64),
        ("n_hot", ctypes.c_int64),
        ("hot_nodes", ctypes.c_void_p),
        ("row_ptr", ctypes.c_void_p),
        ("dense", ctypes.c_void_p),
        ("dense_rows", ctypes.c_int64),
        ("dense_cols", ctypes.c_int64),
        ("packed_dead", ctypes.c_void_p),
    ]

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if not self.struct_bytes:
            self.struct_bytes = ctypes.sizeof(type(self))
        if not self.abi_version:
            self.abi_version = ABI_VERSION


EXPORTS = (
    "ultra_rspmm_abi_version",
    "ultra_segments_bytes",
    "ultra_rspmm_status_string",
    "ultra_rspmm_last_hip_error",
    "ultra_rspmm_device_info",
    "ultra_rspmm_profile_next",
    "ultra_rspmm_event_create",
    "ultra_rspmm_event_destroy",
    "ultra_rspmm_event_elapsed_ms",
    "ultra_rspmm_force_general_path",
    "ultra_rspmm_reserve_cus",
    "ultra_rspmm_workspace_bytes",
    "ultra_rspmm_forward_f32",
    "ultra_rspmm_fwd_f32",
    "ultra_rspmm_forward_boundary_f32",
    "ultra_rspmm_frontier_f32",
    "ultra_first_layer_sparse_supported",
    "ultra_first_layer_sparse_f32",
    "ultra_rspmm_backward_boundary_rows_f32",
    "ultra_rspmm_backward_boundary_rows_workspace",
    "ultra_rspmm_backward_f32",
    "ultra_rspmm_backward_accumulate_f32",
    "ultra_rspmm_backward_weight_f32",
    "ultra_rspmm_backward_active_f32",
    "ultra_node_bitmap",
    "ultra_rspmm_drelation_boundary_f32",
    "ultra_combine_forward_f32",
    "ultra_combine_forward_boundary_f32",
    "ultra_combine_backward_waves",
    "ultra_combine_backward_f32",
    "ultra_combine_dxdu_f32",
    "ultra_combine_backward_fused_waves",
    "ultra_combine_backward_fused_f32",
    "ultra_linear_forward_f32",
    "ultra_score_forward_f32",
    "ultra_relation_project_f32",
    "ultra_relation_project_backward_blocks",
    "ultra_relation_project_backward_f32",
    "ultra_filtered_rank",
    "ultra_filtered_rank_keys",
    "ultra_strict_negative",
    "ultra_edge_removal_weights",
    "ultra_edge_removal_marks",
    "ultra_prepare_queries",
    "ultra_relation_stack_inputs",
    "ultra_statistics_blocks",
    "ultra_statistics_f32",
    "ultra_bce_adversarial_f32",
    "ultra_candidate_tiles",
    "ultra_score_rows_forward_f32",
    "ultra_score_rows_backward_f32",
    "ultra_gather_boundary_rows_f32",
    "ultra_relcsr_coalesce_temp_bytes",
    "ultra_relcsr_coalesce",
    "ultra_relcsr_plan_temp_bytes",
    "ultra_relcsr_plan",
    "ultra_relcsr_dense_bytes",
    "ultra_relcsr_dense",
    "ultra_relation_graph_marks",
    "ultra_calibrate_gather_f32",
    "ultra_first_layer_sparse_train_f32",
    "ultra_first_layer_epilogue_backward_workspace",
    "ultra_first_layer_epilogue_backward_f32",
    "ultra_column_sum_blocks",
    "ultra_column_sum_f32",
    "ultra_dense_layer_supported",
    "ultra_dense_layer_forward_f32",
    "ultra_layer_forward_supported",
    "ultra_layer_forward_f32",
    "ultra_second_layer_sources",
    "ultra_layer_forward_sources_f32",
    "ultra_layer_score_supported",
    "ultra_layer_score_forward_f32",
)

_lib = None


class UltraLibraryError(RuntimeError):
    pass


def load():
    """Load the HIP library once; raise :class:`UltraLibraryError` if it is absent (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UltraLibraryError(
            "%s not found: build it with `make -C ultra_torchdrug_amd/csrc` (hipcc --offload-arch=gfx950) or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU/PyTorch fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as err:  # missing libamdhip64 etc.
        raise UltraLibraryError("cannot load %s: %s" % (LIB_PATH, err)) from err
    missing = [name for name in EXPORTS if not hasattr(lib, name)]
    if missing:
        raise UltraLibraryError("%s lacks symbols %s" % (LIB_PATH, missing))

    vp, i64, i32, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
    seg = ctypes.POINTER(UltraSegments)
    lib.ultra_rspmm_abi_version.restype = i32
    lib.ultra_rspmm_abi_version.argtypes = []
    lib.ultra_layer_forward_supported.restype = i32
    lib.ultra_layer_forward_supported.argtypes = [seg, i64, i64]
    lib.ultra_layer_forward_f32.restype = i32
    lib.ultra_layer_forward_f32.argtypes = [seg, vp, vp, vp, vp, i64, vp, vp, vp, vp, ctypes.c_float, i32, i32, vp, i64, vp]
    lib.ultra_second_layer_sources.restype = i32
    lib.ultra_second_layer_sources.argtypes = [vp, i64, i64, vp, vp, i64, i64, i64, vp, vp, vp, vp]
    lib.ultra_layer_forward_sources_f32.restype = i32
    lib.ultra_layer_forward_sources_f32.argtypes = [seg, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, ctypes.c_float, i32, i32, vp, i64, vp]
    lib.ultra_layer_score_supported.restype = i32
    lib.ultra_layer_score_supported.argtypes = [seg, i64, i64]
    lib.ultra_layer_score_forward_f32.restype = i32
    lib.ultra_layer_score_forward_f32.argtypes = [seg, vp, vp, vp, vp, i64, vp, vp, vp, vp, ctypes.c_float, i32, i32, vp, vp, vp, vp,
                                                  vp, vp, vp, i64, vp]
    lib.ultra_relation_graph_marks.restype = i32
    lib.ultra_relation_graph_marks.argtypes = [vp, vp, vp, vp, i64, i64, vp, vp]
    lib.ultra_segments_bytes.restype = sz
    lib.ultra_segments_bytes.argtypes = []
    lib.ultra_rspmm_status_string.restype = ctypes.c_char_p
    lib.ultra_rspmm_status_string.argtypes = [i32]
    lib.ultra_rspmm_last_hip_error.restype = i32
    lib.ultra_rspmm_last_hip_error.argtypes = []
    lib.ultra_rspmm_device_info.restype = i32
    lib.ultra_rspmm_device_info.argtypes = [i32, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.c_char_p, sz]
    lib.ultra_rspmm_profile_next.restype = i32
    lib.ultra_rspmm_profile_next.argtypes = [vp, vp]
    lib.ultra_rspmm_event_create.restype = i32
    lib.ultra_rspmm_event_create.argtypes = [ctypes.POINTER(vp)]
    lib.ultra_rspmm_event_destroy.restype = i32
    lib.ultra_rspmm_event_destroy.argtypes = [vp]
    lib.ultra_rspmm_event_elapsed_ms.restype = i32
    lib.ultra_rspmm_event_elapsed_ms.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_float)]
    lib.ultra_rspmm_force_general_path.restype = i32
    lib.ultra_rspmm_force_general_path.argtypes = [i32]
    lib.ultra_rspmm_reserve_cus.restype = i32
    lib.ultra_rspmm_reserve_cus.argtypes = [i32]
    lib.ultra_rspmm_workspace_bytes.restype = sz
    lib.ultra_rspmm_workspace_bytes.argtypes = [seg, i64]
    lib.ultra_rspmm_forward_f32.restype = i32
    lib.ultra_rspmm_forward_f32.argtypes = [seg, vp, vp, vp, vp, vp, sz, i64, i64, i64, i32, i32, vp]
    lib.ultra_rspmm_fwd_f32.restype = i32
    lib.ultra_rspmm_fwd_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, i32, vp]
    lib.ultra_rspmm_forward_boundary_f32.restype = i32
    lib.ultra_rspmm_forward_boundary_f32.argtypes = [seg, vp, vp, vp, vp, i64, vp, vp, sz, i64, i64, i64, i32, i32, vp]
    lib.ultra_rspmm_backward_boundary_rows_f32.restype = i32
    lib.ultra_rspmm_backward_boundary_rows_f32.argtypes = [seg, vp, vp, vp, vp, vp, vp, sz, i64, i64, i64, i32, vp]
    lib.ultra_rspmm_backward_boundary_rows_workspace.restype = sz
    lib.ultra_rspmm_backward_boundary_rows_workspace.argtypes = [i64]
    lib.ultra_rspmm_frontier_f32.restype = i32
    lib.ultra_rspmm_frontier_f32.argtypes = [seg, vp, vp, vp, vp, vp, i64, vp, i64, i64, i64, vp]
    lib.ultra_first_layer_sparse_supported.restype = i32
    lib.ultra_first_layer_sparse_supported.argtypes = [i64, i64, i64]
    lib.ultra_first_layer_sparse_f32.restype = i32
    lib.ultra_first_layer_sparse_f32.argtypes = [seg, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, ctypes.c_float, i32, i32, vp, vp, i64, i64,
                                                 vp, i64, i64, vp]
    lib.ultra_rspmm_backward_f32.restype = i32
    lib.ultra_rspmm_backward_f32.argtypes = [seg, seg, vp, vp, vp, vp, vp, vp, vp, sz, i64, i64, i64, i64, i32, i32, vp]
    lib.ultra_rspmm_backward_accumulate_f32.restype = i32
    lib.ultra_rspmm_backward_accumulate_f32.argtypes = [seg, seg, vp, vp, vp, vp, vp, vp, vp, vp, sz, i64, i64, i64, i64, i32,
                                                        i32, vp]
    lib.ultra_rspmm_backward_active_f32.restype = i32
    lib.ultra_rspmm_backward_active_f32.argtypes = [seg, seg, vp, vp, vp, vp, vp, vp, vp, sz, i64, i64, i64, i64, i32, vp, i64, vp, vp]
    lib.ultra_node_bitmap.restype = i32
    lib.ultra_node_bitmap.argtypes = [vp, i64, i64, i64, vp, vp]
    lib.ultra_rspmm_drelation_boundary_f32.restype = i32
    lib.ultra_rspmm_drelation_boundary_f32.argtypes = [seg, vp, i64, vp, vp, vp, vp, vp, vp, vp, sz, i64, i64, i64, vp]
    lib.ultra_rspmm_backward_weight_f32.restype = i32
    lib.ultra_rspmm_backward_weight_f32.argtypes = [seg, vp, vp, vp, vp, vp, i64, i64, i32, i32, vp]
    lib.ultra_combine_forward_f32.restype = i32
    lib.ultra_combine_forward_f32.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_float, i32, i32, vp, vp, i64, i64, vp]
    lib.ultra_combine_forward_boundary_f32.restype = i32
    lib.ultra_combine_forward_boundary_f32.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, ctypes.c_float, i32, i32, vp, i64, i64, vp]
    lib.ultra_prepare_queries.restype = i32
    lib.ultra_prepare_queries.argtypes = [vp, vp, i64, i64, i64, i64, i64, vp, vp, vp, vp, vp]
    lib.ultra_statistics_blocks.restype = i32
    lib.ultra_statistics_blocks.argtypes = [i64]
    lib.ultra_statistics_f32.restype = i32
    lib.ultra_statistics_f32.argtypes = [vp, i64, vp, i64, i64, vp, vp, vp]
    lib.ultra_gather_boundary_rows_f32.restype = i32
    lib.ultra_gather_boundary_rows_f32.argtypes = [vp, vp, i64, vp, vp]
    lib.ultra_score_rows_forward_f32.restype = i32
    lib.ultra_score_rows_forward_f32.argtypes = [vp] * 10 + [i64, i64, vp]
    lib.ultra_score_rows_backward_f32.restype = i32
    lib.ultra_score_rows_backward_f32.argtypes = [vp] * 16 + [i64, i64, i64, vp]
    lib.ultra_candidate_tiles.restype = i32
    lib.ultra_candidate_tiles.argtypes = [vp, i64, i64, i64, i64, vp, vp]
    lib.ultra_bce_adversarial_f32.restype = i32
    lib.ultra_bce_adversarial_f32.argtypes = [vp, i64, i64, ctypes.c_float, vp, vp, vp]
    lib.ultra_relation_stack_inputs.restype = i32
    lib.ultra_relation_stack_inputs.argtypes = [vp, i64, i64, i64, vp, i64, vp, vp, vp, vp]
    lib.ultra_combine_backward_waves.restype = i32
    lib.ultra_combine_backward_waves.argtypes = [i32, i64, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    lib.ultra_combine_backward_f32.restype = i32
    lib.ultra_combine_backward_f32.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_float, i32, vp, vp, vp, vp, vp, vp, i64, i64, vp]
    lib.ultra_combine_dxdu_f32.restype = i32
    lib.ultra_combine_dxdu_f32.argtypes = [vp, vp, vp, vp, vp, i64, i64, vp]
    lib.ultra_combine_backward_fused_waves.restype = i32
    lib.ultra_combine_backward_fused_waves.argtypes = [i32, i64, ctypes.POINTER(i32)]
    lib.ultra_combine_backward_fused_f32.restype = i32
    lib.ultra_combine_backward_fused_f32.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_float, i32, i32, vp, vp, vp, vp, vp, vp, vp,
                                                     vp, vp, sz, vp, i64, i64, i64, vp]
    lib.ultra_linear_forward_f32.restype = i32
    lib.ultra_linear_forward_f32.argtypes = [vp, vp, vp, vp, i64, i64, i64, i32, vp]
    lib.ultra_score_forward_f32.restype = i32
    lib.ultra_score_forward_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, vp]
    lib.ultra_relation_project_f32.restype = i32
    lib.ultra_relation_project_f32.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, vp]
    lib.ultra_relation_project_backward_blocks.restype = i32
    lib.ultra_relation_project_backward_blocks.argtypes = [i32, i64, i64, i64, vp]
    lib.ultra_relation_project_backward_f32.restype = i32
    lib.ultra_relation_project_backward_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i64, i64, i64, i64, vp]
    lib.ultra_filtered_rank.restype = i32
    lib.ultra_filtered_rank.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, vp]
    lib.ultra_filtered_rank_keys.restype = i32
    lib.ultra_filtered_rank_keys.argtypes = [vp, i64, i64, i64, vp, i64, vp, i64, vp, vp, i64, i64, vp, i64, vp]
    lib.ultra_strict_negative.restype = i32
    lib.ultra_strict_negative.argtypes = [vp, i64, vp, vp, i64, i64, i64, vp, i64, vp, vp]
    lib.ultra_edge_removal_weights.restype = i32
    lib.ultra_edge_removal_weights.argtypes = [seg, seg, seg, vp, vp, vp, i64, i64, vp, vp, vp, i64, vp]
    lib.ultra_edge_removal_marks.restype = i32
    lib.ultra_edge_removal_marks.argtypes = [seg, seg, seg, vp, vp, vp, i64, i64, vp, vp, vp, i64, vp, vp, vp, i64, vp]
    lib.ultra_relcsr_coalesce_temp_bytes.restype = sz
    lib.ultra_relcsr_coalesce_temp_bytes.argtypes = [i64]
    lib.ultra_relcsr_coalesce.restype = i32
    lib.ultra_relcsr_coalesce.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, vp, vp, vp, vp, vp,
                                          ctypes.POINTER(i64), ctypes.POINTER(i32), vp, sz, vp]
    lib.ultra_relcsr_plan_temp_bytes.restype = sz
    lib.ultra_relcsr_plan_temp_bytes.argtypes = [i64, i64, i64]
    lib.ultra_relcsr_plan.restype = i32
    lib.ultra_relcsr_plan.argtypes = [vp, vp, vp, i64, i64, i64, i64, i32, i32, i32, i64, i64, i64, vp, i64, vp, i64,
                                      vp, i64, ctypes.POINTER(i64), vp, sz, vp]
    lib.ultra_relcsr_dense_bytes.restype = sz
    lib.ultra_relcsr_dense_bytes.argtypes = [i64, i64, i32]
    lib.ultra_relcsr_dense.restype = i32
    lib.ultra_relcsr_dense.argtypes = [seg, i64, i64, i32, vp, vp]
    lib.ultra_first_layer_sparse_train_f32.restype = i32
    lib.ultra_first_layer_sparse_train_f32.argtypes = [seg, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, ctypes.c_float, i32, i32, vp, vp,
                                                       vp, i64, i64, vp, i64, i64, vp]
    lib.ultra_first_layer_epilogue_backward_workspace.restype = sz
    lib.ultra_first_layer_epilogue_backward_workspace.argtypes = [i32, i64]
    lib.ultra_first_layer_epilogue_backward_f32.restype = i32
    lib.ultra_first_layer_epilogue_backward_f32.argtypes = [vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, ctypes.c_float, i32, i32, vp, vp, vp,
                                                            vp, vp, vp, vp, sz, i64, vp]
    lib.ultra_column_sum_blocks.restype = i32
    lib.ultra_column_sum_blocks.argtypes = []
    lib.ultra_column_sum_f32.restype = i32
    lib.ultra_column_sum_f32.argtypes = [vp, i64, vp, vp, vp, vp]
    lib.ultra_dense_layer_supported.restype = i32
    lib.ultra_dense_layer_supported.argtypes = [seg, i64]
    lib.ultra_dense_layer_forward_f32.restype = i32
    lib.ultra_dense_layer_forward_f32.argtypes = [seg, vp, vp, vp, vp, i64, vp, vp, vp, vp, ctypes.c_float, i32, i32, vp, vp]
    lib.ultra_calibrate_gather_f32.restype = i32
    lib.ultra_calibrate_gather_f32.argtypes = [vp, i64, vp, i64, vp, ctypes.POINTER(i64), vp]
    if lib.ultra_rspmm_abi_version() != ABI_VERSION:
        raise UltraLibraryError("ABI mismatch: library %d, binding %d" % (lib.ultra_rspmm_abi_version(), ABI_VERSION))
    if lib.ultra_segments_bytes() != ctypes.sizeof(UltraSegments):
        raise UltraLibraryError("struct ultra_segments: library %d bytes, binding %d bytes"
                                % (lib.ultra_segments_bytes(), ctypes.sizeof(UltraSegments)))
    _lib = lib
    return lib


def check(status):
    """Turn a non-zero ``ultra_status`` into a RuntimeError (TORCH_CHECK-like behaviour of the reference op)."""
    if status != 0:
        lib = load()
        msg = lib.ultra_rspmm_status_string(status).decode()
        if status == 5:
            msg += " [hipError_t=%d]" % lib.ultra_rspmm_last_hip_error()
        raise RuntimeError("libultra_rspmm: %s" % msg)


def device_info(device=0):
    lib = load()
    n_cu, lds = ctypes.c_int(0), ctypes.c_int(0)
    arch = ctypes.create_string_buffer(64)
    check(lib.ultra_rspmm_device_info(int(device), ctypes.byref(n_cu), ctypes.byref(lds), arch, 64))
    return {"n_cu": n_cu.value, "lds_bytes": lds.value, "arch": arch.value.decode()}

"""Loads libw3hip.so (the HIP extension) through ctypes.  Fails loudly when the
extension is missing: there is no CPU fallback in this package."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# W3HIP_SO: another build of the same library (A/B experiments: `python -m weath3rb0i_amd.build --out libw3hip_x.so -DW3_...`)
SO = os.environ.get("W3HIP_SO") or os.path.join(HERE, "libw3hip.so")

W3_MAX_NODES = 31
W3_NODE_ORDERN, W3_NODE_BEST_OF_TWO, W3_NODE_SLOT_STATE, W3_NODE_APM = 1, 2, 3, 4
W3_APM_ORDER0, W3_APM_ORDER1 = 0, 1
W3_MAX_APM = 4
W3_HIST_NONE, W3_HIST_RAW, W3_HIST_AC, W3_HIST_HUFF = 0, 1, 2, 3
W3_MAX_HUFF = 4
W3_OK, W3_E_INVALID, W3_E_NOSPACE, W3_E_HIP, W3_E_UNSUPPORTED, W3_E_NOMEM, W3_E_FORMAT = 0, -1, -2, -3, -4, -5, -6
W3_OPT_PATH, W3_OPT_TIMING, W3_OPT_CODER, W3_OPT_ACC_LIMIT, W3_OPT_DEBUG_STAMPS, W3_OPT_VARIANT, W3_OPT_SLOT_BUDGET_MB, W3_OPT_VERIFY, W3_OPT_FAULT_BLOCK, W3_OPT_TUNE, W3_OPT_HOST_CHUNK_BLOCKS = 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12
W3_VAR_NO_LDS_ATOMICS, W3_VAR_PARTITION4, W3_VAR_NO_CHAINED_PARTITION, W3_VAR_CM_UNSTAGED, W3_VAR_NO_SIDE_STREAM, W3_VAR_INJECT_LDS_FAULT = 1, 2, 4, 8, 16, 32
W3_VAR_HALF_CU, W3_VAR_FULL_CU = 64, 128
W3_VAR_SLOT_TABLE, W3_VAR_SLOT_SORTED, W3_VAR_DECODE_LANE = 256, 512, 1024
W3_GATHER_AUTO, W3_GATHER_RCCL, W3_GATHER_PEER_COPY = 0, 1, 2
W3_PATH_AUTO, W3_PATH_GENERIC, W3_PATH_TWOPHASE = 0, 1, 2


class Node(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("bits", C.c_uint8), ("align", C.c_uint8), ("history", C.c_uint8),
                ("max_bits", C.c_uint8), ("frozen", C.c_uint8), ("log_cells", C.c_uint8), ("reserved", C.c_uint8), ("table", C.c_uint16 * 8)]


class HuffTable(C.Structure):
    _fields_ = [("code", C.c_uint16 * 256), ("len", C.c_uint8 * 256), ("rem_code", C.c_uint16 * 256), ("rem_len", C.c_uint8 * 256)]


class ModelSpec(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("nodes", Node * W3_MAX_NODES), ("n_huff", C.c_uint32), ("huff", C.POINTER(HuffTable))]


class Timing(C.Structure):
    _fields_ = [("predict_ms", C.c_float), ("coder_ms", C.c_float), ("pack_ms", C.c_float), ("generic_ms", C.c_float),
                ("total_ms", C.c_float), ("path", C.c_uint32), ("n_coder_launches", C.c_uint32),
                ("coder_bytes", C.c_uint64), ("predict_bytes", C.c_uint64), ("n_recoded_blocks", C.c_uint32), ("apm_ms", C.c_float),
                ("slot_ms", C.c_float), ("n_slot_launches", C.c_uint32), ("achash_ms", C.c_float), ("n_parts", C.c_uint32),
                ("n_lds_faults", C.c_uint32), ("n_wide", C.c_uint32), ("part_ms", C.c_float * 4), ("rank_ms", C.c_float * 4),
                ("small_ms", C.c_float)]


EXPORTS = [
    "w3_abi_version", "w3_strerror", "w3_last_error", "w3_ctx_create", "w3_ctx_destroy", "w3_spec_validate",
    "w3_ctx_set_option", "w3_max_compressed_size", "w3_encode_blocks", "w3_decode_blocks", "w3_encode_blocks_device",
    "w3_decode_blocks_device", "w3_encode_submit", "w3_encode_wait", "w3_encode_max_in_flight", "w3_compress_stream", "w3_decompress_stream", "w3_predict_blocks", "w3_stationary_table",
    "w3_get_timing", "w3_selftest_counter_p", "w3_debug_get_stamps", "w3_state_table", "w3_stretch_squash", "w3_huff_tables",
    "w3_shard_range", "w3_encode_blocks_sharded", "w3_encode_blocks_sharded_device", "w3_encode_stats", "w3_encode_stats_device", "w3_sweep_ordern", "w3_sweep_ordern_device", "w3_export_counters",
    "w3_encode_host_submit", "w3_encode_host_wait", "w3_encode_host_max_in_flight", "w3_rccl_library", "w3_rccl_status",
    "w3_encode_sharded_submit", "w3_encode_sharded_wait", "w3_encode_sharded_max_in_flight",
]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise ImportError(
            "weath3rb0i_amd: HIP extension %s is missing - run `python -m weath3rb0i_amd.build` "
            "(or __graft_entry__.build()).  There is no CPU fallback." % SO)
    # torch bundles its own libamdhip64/libhsa-runtime64 (same sonames as /opt/rocm's).  Two HIP runtimes in one
    # process do not work, so let torch load its copy FIRST; libw3hip.so then binds to that already-loaded runtime.
    # Without torch (pure C/C++ hosts) the RUNPATH resolves /opt/rocm/lib.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(SO)
    vp, sz = C.c_void_p, C.c_size_t
    lib.w3_abi_version.restype = C.c_int
    lib.w3_strerror.restype = C.c_char_p
    lib.w3_strerror.argtypes = [C.c_int]
    lib.w3_last_error.restype = C.c_char_p
    lib.w3_last_error.argtypes = [vp]
    lib.w3_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.w3_ctx_destroy.argtypes = [vp]
    lib.w3_ctx_destroy.restype = None
    lib.w3_spec_validate.argtypes = [C.POINTER(ModelSpec)]
    lib.w3_ctx_set_option.argtypes = [vp, C.c_int, C.c_int64]
    lib.w3_max_compressed_size.restype = sz
    lib.w3_max_compressed_size.argtypes = [sz, sz]
    lib.w3_encode_blocks.argtypes = [vp, C.POINTER(ModelSpec), vp, sz, sz, vp, sz, C.POINTER(sz), vp]
    lib.w3_decode_blocks.argtypes = [vp, C.POINTER(ModelSpec), vp, sz, vp, sz, sz, C.c_uint64, vp]
    lib.w3_encode_blocks_device.argtypes = [vp, C.POINTER(ModelSpec), vp, sz, sz, vp, sz, vp, vp, vp]
    lib.w3_decode_blocks_device.argtypes = [vp, C.POINTER(ModelSpec), vp, sz, vp, sz, sz, C.c_uint64, vp, vp]
    lib.w3_encode_submit.argtypes = [vp, C.POINTER(ModelSpec), vp, sz, sz, vp, sz, vp, vp, vp, C.POINTER(C.c_int)]
    lib.w3_encode_wait.argtypes = [vp, C.c_int]
    lib.w3_encode_max_in_flight.argtypes = [C.POINTER(ModelSpec), sz, sz]
    lib.w3_encode_max_in_flight.restype = C.c_int
    lib.w3_compress_stream.argtypes = [vp, C.POINTER(ModelSpec), vp, sz, vp, sz, C.POINTER(sz)]
    lib.w3_decompress_stream.argtypes = [vp, C.POINTER(ModelSpec), vp, sz, vp, sz, C.POINTER(sz)]
    lib.w3_predict_blocks.argtypes = [vp, C.POINTER(ModelSpec), vp, sz, sz, vp]
    lib.w3_stationary_table.argtypes = [vp, sz, vp]
    lib.w3_selftest_counter_p.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.w3_debug_get_stamps.argtypes = [vp, C.POINTER(C.c_uint64 * 8)]
    lib.w3_get_timing.argtypes = [vp, C.POINTER(Timing)]
    lib.w3_state_table.argtypes = [vp]
    lib.w3_stretch_squash.argtypes = [vp, vp]
    lib.w3_huff_tables.argtypes = [vp, sz, C.c_uint8, C.c_uint8, C.POINTER(HuffTable)]
    lib.w3_shard_range.argtypes = [sz, C.c_int, C.c_int, C.POINTER(sz), C.POINTER(sz)]
    lib.w3_encode_blocks_sharded.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(ModelSpec), vp, sz, sz, vp, sz, C.POINTER(sz), vp]
    lib.w3_encode_blocks_sharded_device.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(ModelSpec), C.POINTER(vp), C.POINTER(sz), sz, C.c_int, vp, sz, vp,
                                                    C.POINTER(C.c_uint64), C.c_int]
    lib.w3_encode_stats.argtypes = [vp, C.POINTER(ModelSpec), vp, sz, sz, vp]
    lib.w3_encode_stats_device.argtypes = [vp, C.POINTER(ModelSpec), vp, sz, sz, vp, vp]
    lib.w3_sweep_ordern.argtypes = [vp, vp, sz, sz, vp, vp, sz, vp]
    lib.w3_sweep_ordern_device.argtypes = [vp, vp, sz, sz, vp, vp, sz, vp]
    lib.w3_export_counters.argtypes = [vp, C.POINTER(ModelSpec), vp, sz, vp]
    lib.w3_encode_host_submit.argtypes = [vp, C.POINTER(ModelSpec), vp, sz, sz, vp, sz, vp, C.POINTER(C.c_int)]
    lib.w3_encode_host_wait.argtypes = [vp, C.c_int, C.POINTER(sz)]
    lib.w3_encode_host_max_in_flight.argtypes = [C.POINTER(ModelSpec), sz, sz]
    lib.w3_encode_host_max_in_flight.restype = C.c_int
    lib.w3_encode_sharded_submit.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(ModelSpec), C.POINTER(vp), C.POINTER(sz), sz, C.POINTER(C.c_int)]
    lib.w3_encode_sharded_wait.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, vp, sz, vp, C.POINTER(C.c_uint64), C.c_int]
    lib.w3_encode_sharded_max_in_flight.argtypes = [C.POINTER(ModelSpec), C.POINTER(sz), C.c_int, sz]
    lib.w3_encode_sharded_max_in_flight.restype = C.c_int
    lib.w3_rccl_library.argtypes = [C.c_char_p]
    lib.w3_rccl_status.argtypes = [C.c_char_p, sz]
    _lib = lib
    return lib

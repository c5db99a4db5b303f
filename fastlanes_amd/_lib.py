"""ctypes loader for libfastlanes_amd.so (the C ABI of include/fastlanes_amd.h).

There is no fallback: if the HIP library is missing or fails to load, importing
the codec raises.  Nothing here (or anywhere in this package) touches oracle/.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FL_LIB=<path> selects another build of the same library -- the A/B tools use it to load libfastlanes_amd_full.so
# (make -C fastlanes_amd/csrc FULL=1: every cell-column instance, whatever the dispatch table says).  Never a fallback.
LIB_PATH = os.environ.get("FL_LIB") or os.path.join(_HERE, "libfastlanes_amd.so")

TYPES = ("u8", "u16", "u32", "u64")
CTYPE = {"u8": ctypes.c_uint8, "u16": ctypes.c_uint16, "u32": ctypes.c_uint32, "u64": ctypes.c_uint64}
BITS = {"u8": 8, "u16": 16, "u32": 32, "u64": 64}

# method -> (device-tier argtypes builder, host-tier argtypes builder); P = void*
_P = ctypes.c_void_p
_U = ctypes.c_uint
_Z = ctypes.c_size_t
_Q = ctypes.c_uint64


def _signatures(ty):
    c = CTYPE[ty]
    dev = {
        "pack": [_U, _P, _P, _Z, _P],
        "unpack": [_U, _P, _P, _Z, _P],
        "unpack_single": [_U, _P, _Z, _P, _Z, _P, _P, _P],
        "for_pack": [_U, _P, _P, _Z, _P, _Z, _P],
        "unfor_pack": [_U, _P, _P, _Z, _P, _Z, _P],
        "delta": [_P, _P, _P, _Z, _P],
        "undelta": [_P, _P, _P, _Z, _P],
        "undelta_pack": [_U, _P, _P, _P, _Z, _P],
        "transpose": [_P, _P, _Z, _P],
        "untranspose": [_P, _P, _Z, _P],
        "undelta_pack_untranspose": [_U, _P, _P, _P, _Z, _P],
        "transpose_delta_pack": [_U, _P, _P, _P, _Z, _P],
        "unpack_block_sums": [_U, _P, _Z, _P, _P],
        "block_min_max": [_P, _Z, _P, _P, _P],
        "unpack_compare": [_U, _P, ctypes.c_int, c, _Z, _P, _P],
        "unpack_mixed": [_P, _P, _P, _P],
        "pack_mixed": [_P, _P, _P, _P],
        "unpack_widths": [_P, _P, _P, _Z, _P, _Z, _P, _P],
        "pack_widths": [_P, _P, _P, _P, _Z, _Z, _P, _P],
        "unpack_single_widths": [_P, _P, _P, _Z, _Z, _P, _Z, _P, _P, _P],
        "unfor_pack_widths": [_P, _P, _P, _Z, _P, _Z, _P, _Z, _P, _P],
        "for_pack_widths": [_P, _P, _P, _P, _Z, _P, _Z, _Z, _P, _P],
        "undelta_pack_widths": [_P, _P, _P, _Z, _P, _P, _Z, _P, _P],
        "undelta_pack_untranspose_widths": [_P, _P, _P, _Z, _P, _P, _Z, _P, _P],
        "transpose_delta_pack_widths": [_P, _P, _P, _P, _P, _Z, _Z, _P, _P],
        "for_widths": [_P, _P, _Z, _P, _P],
        "unpack_batch": [_P, _P, _P, _P, _Z, ctypes.c_uint32, _P, _P],
        "pack_batch": [_P, _P, _P, _P, _Z, ctypes.c_uint32, _P, _P],
        "unfor_pack_batch": [_P, _P, _P, _P, _P, _Z, ctypes.c_uint32, _P, _P],
        "for_pack_batch": [_P, _P, _P, _P, _P, _Z, ctypes.c_uint32, _P, _P],
        "undelta_pack_batch": [_P, _P, _P, _P, _P, _Z, ctypes.c_uint32, ctypes.c_int, _P, _P],
        "transpose_delta_pack_batch": [_P, _P, _P, _P, _P, _Z, ctypes.c_uint32, _P, _P],
    }
    host = {
        "pack_host": [_U, _P, _P, _Z],
        "unpack_host": [_U, _P, _P, _Z],
        "unpack_single_host": [_U, _P, _Z, _Q, _P],
        "for_pack_host": [_U, _P, c, _P, _Z],
        "unfor_pack_host": [_U, _P, c, _P, _Z],
        "delta_host": [_P, _P, _P, _Z],
        "undelta_host": [_P, _P, _P, _Z],
        "undelta_pack_host": [_U, _P, _P, _P, _Z],
        "transpose_host": [_P, _P, _Z],
        "untranspose_host": [_P, _P, _Z],
    }
    dev.update(host)
    return dev


# include/fastlanes_amd_internal.h: test / measurement hooks, not part of the stable ABI
INTERNAL_SYMBOLS = ["fl_internal_set_kernel_policy", "fl_internal_get_kernel_policy", "fl_internal_probe_memory_classes",
                    "fl_internal_bare_stream", "fl_internal_bare_stream_shape", "fl_internal_zero_copy_fallbacks",
                    "fl_internal_column_pair_classes", "fl_internal_selftune_check", "fl_internal_choose_chunks",
                    "fl_internal_pair_chunk_cache"]


def exported_symbols():
    """Every symbol include/fastlanes_amd.h and include/fastlanes_amd_internal.h declare."""
    names = ["fl_version", "fl_status_string", "fl_last_hip_error", "fl_packed_len",
             "fl_mixed_plan_create", "fl_mixed_plan_destroy", "fl_mixed_plan_n_blocks",
             "fl_mixed_plan_packed_bytes", "fl_mixed_plan_offsets", "fl_mixed_plan_widths",
             "fl_widths_to_offsets", "fl_fill_random", "fl_host_release", "fl_column_pair_alloc", "fl_column_pair_free"]
    names += INTERNAL_SYMBOLS
    for ty in TYPES:
        names += [f"fl_{ty}_{m}" for m in _signatures(ty)]
    return names


_LIB = None


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C fastlanes_amd/csrc). "
            "fastlanes_amd has no CPU fallback.")
    # ONE HIP runtime per process: PyTorch ships its own libamdhip64.so, libfastlanes_amd.so is linked against /opt/rocm's.  Whichever
    # is loaded first serves both (same SONAME) -- but if this library came first, torch would still load its bundled copy next to
    # it, and a kernel launched through one runtime on memory allocated through the other fails with hipErrorNoDevice (seen in
    # round 4).  The Python mirror exists to be used with torch tensors, so torch goes first whenever it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    lib.fl_version.restype = ctypes.c_char_p
    lib.fl_status_string.restype = ctypes.c_char_p
    lib.fl_status_string.argtypes = [ctypes.c_int]
    lib.fl_last_hip_error.restype = ctypes.c_int
    lib.fl_packed_len.restype = ctypes.c_size_t
    lib.fl_packed_len.argtypes = [_U, _U]
    lib.fl_mixed_plan_create.restype = ctypes.c_int
    lib.fl_mixed_plan_create.argtypes = [_U, _P, _Z, ctypes.POINTER(_P)]
    lib.fl_mixed_plan_destroy.restype = None
    lib.fl_mixed_plan_destroy.argtypes = [_P]
    lib.fl_mixed_plan_n_blocks.restype = ctypes.c_size_t
    lib.fl_mixed_plan_n_blocks.argtypes = [_P]
    lib.fl_mixed_plan_packed_bytes.restype = ctypes.c_uint64
    lib.fl_mixed_plan_packed_bytes.argtypes = [_P]
    lib.fl_mixed_plan_offsets.restype = _P
    lib.fl_mixed_plan_offsets.argtypes = [_P]
    lib.fl_mixed_plan_widths.restype = _P
    lib.fl_mixed_plan_widths.argtypes = [_P]
    lib.fl_internal_set_kernel_policy.restype = None
    lib.fl_internal_set_kernel_policy.argtypes = [ctypes.c_int]
    lib.fl_internal_get_kernel_policy.restype = ctypes.c_int
    lib.fl_internal_get_kernel_policy.argtypes = []
    lib.fl_host_release.restype = None
    lib.fl_host_release.argtypes = []
    lib.fl_fill_random.restype = ctypes.c_int
    lib.fl_fill_random.argtypes = [_P, _Z, _Q, _P]
    lib.fl_internal_probe_memory_classes.restype = ctypes.c_int
    lib.fl_internal_probe_memory_classes.argtypes = [_P, _Z, ctypes.POINTER(ctypes.c_int), _P]
    lib.fl_internal_zero_copy_fallbacks.restype = ctypes.c_uint64
    lib.fl_internal_zero_copy_fallbacks.argtypes = []
    lib.fl_internal_bare_stream.restype = ctypes.c_int
    lib.fl_internal_bare_stream.argtypes = [_P, _Z, _P, _Z, _P, _Z, _Z, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]
    lib.fl_internal_bare_stream_shape.restype = ctypes.c_int
    lib.fl_internal_bare_stream_shape.argtypes = ([ctypes.c_int, _U, _U] + [ctypes.POINTER(ctypes.c_size_t)] * 3 + [ctypes.POINTER(ctypes.c_int)] * 3 +
                                                  [ctypes.POINTER(ctypes.c_uint)])
    lib.fl_column_pair_alloc.restype = ctypes.c_int
    lib.fl_column_pair_alloc.argtypes = [_Z, _Z, _Z, ctypes.c_int, _P] + [ctypes.POINTER(_P)] * 4 + [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint32)]
    lib.fl_column_pair_free.restype = ctypes.c_int
    lib.fl_column_pair_free.argtypes = [_P]
    lib.fl_internal_selftune_check.restype = ctypes.c_int
    lib.fl_internal_selftune_check.argtypes = [ctypes.c_int, _U, _U, _P, _P, _P, _Z, _P, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                               ctypes.POINTER(ctypes.c_int)]
    lib.fl_internal_pair_chunk_cache.restype = ctypes.c_size_t
    lib.fl_internal_pair_chunk_cache.argtypes = [_Z]
    lib.fl_internal_choose_chunks.restype = ctypes.c_size_t
    lib.fl_internal_choose_chunks.argtypes = [ctypes.POINTER(ctypes.c_int), _Z, _Z, _Z, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    lib.fl_internal_column_pair_classes.restype = ctypes.c_char_p
    lib.fl_internal_column_pair_classes.argtypes = [_P]
    lib.fl_widths_to_offsets.restype = ctypes.c_int
    lib.fl_widths_to_offsets.argtypes = [_U, _P, _Z, _P, _P, _P, _P]
    for ty in TYPES:
        for m, argtypes in _signatures(ty).items():
            fn = getattr(lib, f"fl_{ty}_{m}")
            fn.restype = ctypes.c_int
            fn.argtypes = argtypes
    _LIB = lib
    return lib

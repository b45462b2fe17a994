"""ctypes binding of libdig_hip.so (the C ABI declared in include/dig_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C digdriver_amd/csrc``.  There is NO CPU fallback: if the shared library
is missing or a call fails, a ``DigHipError`` is raised.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DIG_HIP_LIB: developer override used to A/B kernel builds -- tools/variant_bench.py; still a HIP library, no fallback)
LIB_PATH = os.environ.get("DIG_HIP_LIB") or os.path.join(_HERE, "lib", "libdig_hip.so")

DIG_F32, DIG_F64, DIG_I16, DIG_BF16 = 0, 1, 2, 3
DIG_PIPE_CONTEXTS, DIG_PIPE_DOT, DIG_PIPE_STATISTICS, DIG_PIPE_WORKLIST_CLEAN, DIG_PIPE_COMPACT_L, DIG_PIPE_RECORDS = 1, 2, 4, 8, 16, 32
DIG_REC_DOUBLES, DIG_REC_MU, DIG_REC_SIGMA, DIG_REC_ROBS_FLAG = 10, 7, 8, 9
GENE_CLASSES = ("SYN", "MIS", "NONS", "SPL", "TRUNC", "NONSYN")
GS_PLANES = tuple("EXP_" + c for c in GENE_CLASSES) + tuple("PVAL_%s_BURDEN" % c for c in GENE_CLASSES) + \
    tuple("PVAL_%s_BURDEN_SAMPLE" % c for c in GENE_CLASSES) + ("THETA_INDEL", "EXP_INDEL", "PVAL_INDEL_BURDEN", "PVAL_MUT_BURDEN")
ES_PLANES = ("EXP_SNV", "PVAL_SNV_BURDEN", "PVAL_SAMPLE_BURDEN", "THETA_INDEL", "EXP_INDEL",
             "PVAL_INDEL_BURDEN", "PVAL_MUT_BURDEN")


class DigHipError(RuntimeError):
    pass


_lib = None
TORCH_FREE = False               # set (before the first call) by a process that only uses numpy arrays and the `_host` entry points
_loaded_without_torch = False


def _need_torch():
    if _loaded_without_torch and "torch" not in sys.modules:
        raise DigHipError("this process declared itself torch-free (digdriver_amd._lib.TORCH_FREE) and loaded libdig_hip.so first: "
                          "device tensors are not available in it (PyTorch must be imported before the library)")


_vp, _i64, _int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int

# name -> argtypes (restype is always int); mirrors include/dig_hip.h one to one
_SIGNATURES = {
    "dig_nb_midp_upper": [_vp, _vp, _vp, _vp, _i64, _vp],
    "dig_nb_midp_upper_host": [_vp, _vp, _vp, _vp, _i64, _int],
    "dig_nb_exact": [_vp, _vp, _vp, _vp, _i64, _vp],
    "dig_nb_exact_host": [_vp, _vp, _vp, _vp, _i64, _int],
    "dig_nb_greater": [_vp, _vp, _vp, _vp, _i64, _vp],
    "dig_nb_greater_host": [_vp, _vp, _vp, _vp, _i64, _int],
    "dig_nb_midp_twosided": [_vp, _vp, _vp, _vp, _i64, _vp],
    "dig_nb_midp_twosided_host": [_vp, _vp, _vp, _vp, _i64, _int],
    "dig_fisher": [_vp, _vp, _vp, _i64, _vp],
    "dig_fisher_host": [_vp, _vp, _vp, _i64, _int],
    "dig_normal_params_to_gamma": [_vp, _vp, _vp, _vp, _i64, _vp],
    "dig_normal_params_to_gamma_host": [_vp, _vp, _vp, _vp, _i64, _int],
    "dig_element_stats": [_vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp],
    "dig_element_stats_host": [_vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _int],
    "dig_accumulate_elements": [_vp] * 8 + [_int] + [_vp] * 11 + [_i64, _i64, _i64, _vp, _i64, _vp],
    "dig_accumulate_elements_host": [_vp] * 8 + [_int] + [_vp] * 11 + [_i64, _i64, _i64, _int],
    "dig_scale_suffstats": [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp],
    "dig_scale_suffstats_host": [_vp, _vp, _i64, _i64, _vp, _int],
    "dig_scale_factors": [_vp, _int, _i64, _vp, _vp, _vp],
    "dig_scale_suffstats_chunked": [_vp, _vp, _i64, _vp, _int, _vp, _vp, _i64, _vp],
    "dig_scale_factors_chunked": [_vp, _int, _vp, _int, _i64, _vp, _vp, _vp, _vp],
    "dig_scale_factors_local": [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "dig_element_pipeline": [_vp] * 25 + [_i64, _i64, _i64, _vp, _int, _vp, _i64, _vp],
    "dig_bin_records_pack": [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp],
    "dig_element_records_unpack": [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _int, _vp],
    "dig_element_pipeline_prepare": [_vp, _i64, _i64, _vp, _i64, _vp, _vp],
    "dig_element_pipeline_host": [_vp] * 25 + [_i64, _i64, _i64, _int],
    "dig_gene_stats": [_vp, _vp, _vp, _vp, _vp, _int, _vp, _int, _vp, _vp, _vp, _vp, _int, _vp, _i64, _i64, _vp],
    "dig_gene_stats_host": [_vp, _vp, _vp, _vp, _vp, _int, _vp, _int, _vp, _vp, _vp, _vp, _int, _vp, _i64, _i64, _int],
    "dig_gene_pipeline": [_vp] * 15 + [_int] + [_vp] * 9 + [_i64, _i64, _i64, _vp, _i64, _vp],
    "dig_gene_pipeline_host": [_vp] * 15 + [_int] + [_vp] * 9 + [_i64, _i64, _i64, _int],
    "dig_count_contexts": [_vp, _i64, _vp, _vp, _int, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "dig_count_contexts_host": [_vp, _i64, _vp, _vp, _int, _vp, _vp, _vp, _vp, _i64, _vp, _int],
    "dig_count_contexts2": [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _int, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "dig_count_contexts2_host": [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _int, _vp, _vp, _vp, _vp, _i64, _vp, _int],
    "dig_write_tsv_host": [ctypes.c_char_p, ctypes.c_char_p, _vp, _vp, _i64, _int, _vp, _vp, _int],
    "dig_mutation_file_parse_host": [ctypes.c_char_p, _vp, _vp, _vp, _vp],
    "dig_mutation_file_fetch_host": [_vp] * 9,
    "dig_mutation_file_flags_host": [_vp, _vp, _vp],
    "dig_mutation_file_free_host": [_vp],
    "dig_stage_timer_create": [_vp],
    "dig_stage_timer_arm": [_vp, _int],
    "dig_stage_timer_read": [_vp, _vp],
    "dig_stage_timer_selftest": [_vp, _vp],
    "dig_stage_timer_destroy": [_vp],
    "dig_overlap_join_count": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp],
    "dig_overlap_join_fill": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "dig_overlap_join_count_host": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _int],
    "dig_overlap_join_fill_host": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _int],
    "dig_ideal_overlaps_host": [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp, _vp],
    "dig_gather_bins": [_vp, _int, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _int, _int, _vp],
    "dig_gather_bins_host": [_vp, _int, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _int, _int, _int],
    "dig_tiled_nb_test": [_vp, _int, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp],
    "dig_base_tile_probs": [_vp, _i64, _vp, _vp, _int, _vp, _vp, _vp, _i64, _vp, _i64, _int, _i64, _vp, _vp, _vp, _vp],
    "dig_tile_mut_counts": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _vp, _vp],
    "dig_base_tile_probs_ctx": [_vp, _i64, _vp, _vp, _int, _vp, _vp, _vp, _i64, _vp, _i64, _int, _int, _i64, _vp, _vp, _vp, _vp],
    "dig_base_tile_probs_ctx_host": [_vp, _i64, _vp, _vp, _int, _vp, _vp, _vp, _i64, _vp, _i64, _int, _int, _i64, _vp, _vp, _vp, _int],
    "dig_base_tile_probs_host": [_vp, _i64, _vp, _vp, _int, _vp, _vp, _vp, _i64, _vp, _i64, _int, _i64, _vp, _vp, _vp, _int],
    "dig_tile_mut_counts_host": [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _int, _i64, _i64, _i64, _vp, _int],
    "dig_tiled_nb_test_host": [_vp, _int, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _int],
    "dig_bh_qvalues_sorted": [_vp, _i64, _i64, _vp, _vp, _i64, _vp],
    "dig_sort_rows": [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp],
    "dig_bh_qvalues_ragged": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _int, _vp, _i64, _vp],
    "dig_rbf_cross": [_vp, _vp, _i64, _i64, _i64, ctypes.c_double, ctypes.c_double, _vp, _vp],
    "dig_rbf_backward": [_vp, _vp, _i64, _i64, ctypes.c_double, ctypes.c_double, _vp, _vp, _vp],
}

# entry points that return a byte count (int64) instead of a status
_SIZE_QUERIES = {
    "dig_element_stats_workspace": [_i64, _i64],
    "dig_accumulate_workspace": [_i64, _i64],
    "dig_scale_suffstats_workspace": [_i64, _i64],
    "dig_element_pipeline_workspace": [_i64, _i64],
    "dig_scale_suffstats_chunked_workspace": [_vp, _int, _i64],
    "dig_rbf_backward_partials": [_i64, _i64],
    "dig_bin_records_bytes": [_i64, _i64],
    "dig_element_records_bytes": [_i64, _i64],
    "dig_bh_workspace": [_i64, _i64],
    "dig_bh_ragged_workspace": [_vp, _i64],
}

ABI_VERSION = 12         # include/dig_hip.h: DIG_ABI_VERSION

EXPORTED_SYMBOLS = tuple(_SIGNATURES) + tuple(_SIZE_QUERIES) + ("dig_abi_version", "dig_last_error",
                                                                "dig_device_count")


import threading as _threading
_LOAD_LOCK = _threading.Lock()


def load():
    """Load libdig_hip.so once; raise DigHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _LOAD_LOCK:                                  # (prewarm_in_background loads on a thread of its own)
        return _load_locked()


def _load_locked():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DigHipError(
            "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C digdriver_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64; if ours were the first HIP runtime in the process, torch
    # could no longer initialise its device layer (torch.cuda.is_available() -> False).  Import torch first
    # so both use the runtime torch was built against -- unless the process has declared that it stays on the `_host` entry
    # points (TORCH_FREE: the single-cohort command lines, whose 1.5 s of `import torch` bought nothing; a later
    # dev_ptr / stream_ptr in such a process is refused, see _need_torch).
    global _loaded_without_torch
    if TORCH_FREE and "torch" not in sys.modules:
        _loaded_without_torch = True
    else:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:
        raise DigHipError("cannot load %s: %s" % (LIB_PATH, exc)) from exc
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    for name, argtypes in _SIZE_QUERIES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int64
    lib.dig_abi_version.restype = ctypes.c_int
    if lib.dig_abi_version() != ABI_VERSION:
        raise DigHipError("%s has ABI version %d, this package needs %d: rebuild it (python -c 'import __graft_entry__ as "
                          "g; g.build()')" % (LIB_PATH, lib.dig_abi_version(), ABI_VERSION))
    lib.dig_last_error.restype = ctypes.c_char_p
    lib.dig_device_count.restype = ctypes.c_int
    _lib = lib
    return lib


def last_error():
    return load().dig_last_error().decode("utf-8", "replace")


def call(name, *args):
    """Invoke an entry point; non-zero status -> DigHipError(dig_last_error())."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise DigHipError("%s failed (%d): %s" % (name, rc, last_error()))


def workspace_bytes(kind, E, C):
    """Scratch bytes needed by dig_element_stats ("element_stats") / dig_accumulate_elements ("accumulate")."""
    lib = load()
    fn = {"element_stats": lib.dig_element_stats_workspace, "accumulate": lib.dig_accumulate_workspace,
          "suffstats": lib.dig_scale_suffstats_workspace, "pipeline": lib.dig_element_pipeline_workspace}[kind]
    return int(fn(int(E), int(C)))


def device_count():
    """Number of gfx950 devices (0 when HIP reports none)."""
    n = load().dig_device_count()
    return max(n, 0)


def prewarm_in_background(device=0):
    """Start the HIP runtime, the device context and the library's code objects on a thread of their own (one `_host` call on a
    single element) while the caller imports pandas and parses its files: 0.1 - 0.2 s of a one-cohort command line that were spent
    inside its first library call.  ctypes releases the interpreter lock for the duration of the call.  Errors are left to the
    first real call (which reports them)."""
    import threading

    def run():
        try:
            if device_count() < 1:
                return
            a = np.ones(1)
            out = [np.empty(1), np.empty(1)]
            call("dig_normal_params_to_gamma_host", host_ptr(a), host_ptr(a), host_ptr(out[0]), host_ptr(out[1]), 1, device)
        except Exception:
            pass

    t = threading.Thread(target=run, name="dig-prewarm", daemon=True)
    t.start()
    return t


def require_device():
    if device_count() < 1:
        raise DigHipError("no gfx950 (MI355X) device is visible; the burden-test path has no CPU fallback")


# ---- pointer helpers -------------------------------------------------------
def host_ptr(arr):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    if arr is None:
        return None
    assert isinstance(arr, np.ndarray) and arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(ctypes.c_void_p)


def as_host(x, dtype):
    """C-contiguous numpy array of `dtype` (copying only when needed)."""
    return np.ascontiguousarray(np.asarray(x), dtype=dtype)


def dev_ptr(t):
    """void* of a contiguous torch CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    _need_torch()
    assert t.is_cuda and t.is_contiguous(), "device entry points need contiguous CUDA tensors"
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(stream=None):
    """hipStream_t of a torch stream (default: torch's current stream)."""
    _need_torch()
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)

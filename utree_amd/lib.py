"""ctypes binding of the C-ABI in include/utree_amd.h.

The shared library is built in-tree (utree_amd/libutree_amd.so) by `make -C utree_amd/csrc` or
`__graft_entry__.build()`.  If it is missing, load() raises: there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("UTREE_AMD_SO") or os.path.join(_HERE, "libutree_amd.so")   # override: kernel experiments only
CLI_PATH = os.path.join(_HERE, "xtree-searchGG")
COMPRESS_CLI_PATH = os.path.join(_HERE, "xtree-compress")
RANK_CLI_PATH = os.path.join(_HERE, "xtree-search")
BUILD_GG_CLI_PATH = os.path.join(_HERE, "utree-buildGG")
BUILD_CLI_PATH = os.path.join(_HERE, "utree-build")
_LIB = None

OK, E_IO, E_FORMAT, E_UNSUPPORTED, E_NOMEM, E_HIP, E_ARG, E_NOLABELS, E_FASTA, E_RCCL, E_BUILD, E_DEVICE = range(12)
BUILD_E_MAP_EMPTY, BUILD_E_MAP, BUILD_E_FASTA, BUILD_E_NO_KMERS, BUILD_E_NAME = range(1, 6)
FINE_AUTO = -1
FANOUT_NONE, FANOUT_BROADCAST, FANOUT_UPLOAD = range(3)
REPLICATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p))
UPLOAD_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p))
INPUT_REFERENCE, INPUT_FASTQ, INPUT_FASTA_MULTILINE, INPUT_AUTO = range(4)


class CtrInfo(C.Structure):
    _fields_ = [("W", C.c_uint32), ("I", C.c_uint32), ("k", C.c_uint32), ("SZ", C.c_uint32), ("n_nodes", C.c_uint64),
                ("n_labels", C.c_uint32), ("binix_width", C.c_uint32), ("bin_total", C.c_uint64),
                ("file_bytes", C.c_uint64)]


class DevInfo(C.Structure):
    _fields_ = [("fine_bits", C.c_uint32), ("record_bytes", C.c_uint32), ("image_bytes", C.c_uint64),
                ("irregular_bins", C.c_uint64), ("generic_mode", C.c_uint32), ("device", C.c_int32),
                ("vote_table", C.c_uint32), ("lane_pass", C.c_uint32), ("bucket_bytes", C.c_uint32), ("strand_views", C.c_uint32),
                ("overflow_chains", C.c_uint32), ("pad0", C.c_uint32), ("overflow_bytes", C.c_uint64)]


class FastaError(C.Structure):
    _fields_ = [("code", C.c_int), ("read_index", C.c_uint64)]


class SearchStats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("good_finds", C.c_uint64), ("seconds_total", C.c_double),
                ("seconds_kernels", C.c_double), ("fasta_error", FastaError), ("pipeline", C.c_int), ("n_lanes", C.c_int),
                ("seconds_read", C.c_double), ("seconds_frame", C.c_double), ("seconds_classify_format", C.c_double),
                ("seconds_order_wait", C.c_double), ("seconds_d2h", C.c_double), ("seconds_write", C.c_double),
                ("bytes_in", C.c_uint64), ("bytes_out", C.c_uint64)]


class CompressStats(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_labels", C.c_uint64), ("label_count_total", C.c_uint64), ("W", C.c_uint32),
                ("I", C.c_uint32), ("seconds", C.c_double)]


class BuildStats(C.Structure):
    _fields_ = [("n_seqs", C.c_uint64), ("n_kmers", C.c_uint64), ("n_nodes", C.c_uint64), ("n_labels", C.c_uint64),
                ("error_line", C.c_uint64), ("error_kind", C.c_int), ("W", C.c_uint32), ("I", C.c_uint32),
                ("seconds", C.c_double), ("n_distinct", C.c_uint64), ("map_bytes", C.c_uint64), ("map_lines", C.c_uint64),
                ("map_error", C.c_int)]


class RankParams(C.Structure):
    """SLACK, SPARSITY, TOLERANCE_THRESHOLD of the rank-specific search (itree.c:952-960)."""
    _fields_ = [("slack", C.c_uint32), ("sparsity", C.c_uint32), ("tolerance", C.c_uint32)]


class Result(C.Structure):
    _fields_ = [("label", C.c_uint32), ("cut", C.c_int32), ("found", C.c_uint32), ("uix", C.c_uint32),
                ("sl", C.c_uint32), ("ol", C.c_uint32)]


# every symbol include/utree_amd.h declares
SYMBOLS = {
    "utree_strerror": (C.c_char_p, [C.c_int]),
    "utree_abi_version": (C.c_int, []),
    "utree_last_hip_error": (C.c_char_p, []),
    "utree_ctr_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "utree_ctr_from_memory": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p,
                                        C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "utree_ctr_close": (None, [C.c_void_p]),
    "utree_ctr_get_info": (C.c_int, [C.c_void_p, C.POINTER(CtrInfo)]),
    "utree_ctr_label": (C.c_void_p, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "utree_dev_image_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "utree_dev_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "utree_dev_upload_seconds": (C.c_int, [C.POINTER(C.c_double)]),
    "utree_dev_build": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                  C.c_void_p, C.POINTER(C.c_void_p)]),
    "utree_dev_image": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "utree_dev_attach": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "utree_dev_free": (None, [C.c_void_p]),
    "utree_dev_get_info": (C.c_int, [C.c_void_p, C.POINTER(DevInfo)]),
    "utree_dev_replicate": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "utree_dev_replicate_seconds": (C.c_double, []),
    "utree_dev_fanout": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "utree_dev_fanout_with": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                        C.c_void_p, C.c_void_p]),
    "utree_rccl_unique_id": (C.c_int, [C.c_void_p, C.c_size_t]),
    "utree_dev_replicate_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                           C.POINTER(C.c_void_p)]),
    "utree_classify_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int]),
    "utree_classify_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64,
                                       C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "utree_classify_poll": (C.c_int, [C.c_void_p]),
    "utree_lookup_words": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "utree_classify_kernel_name": (C.c_char_p, [C.c_void_p]),
    "utree_model_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    "utree_classify_kernel_time": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "utree_fasta_frame": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(FastaError)]),
    "utree_format_records": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "utree_compress_file": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(CompressStats)]),
    "utree_rank_params_default": (None, [C.POINTER(RankParams)]),
    "utree_rank_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int,
                                                C.POINTER(RankParams)]),
    "utree_rank_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32,
                                   C.c_int, C.POINTER(RankParams), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "utree_rank_reset": (C.c_int, [C.c_void_p]),
    "utree_format_rank_records": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                               C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "utree_rank_search_file": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(RankParams),
                                         C.c_int, C.POINTER(SearchStats)]),
    "utree_reads_frame": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(FastaError)]),
    "utree_search_file_opts": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int,
                                         C.c_int, C.POINTER(SearchStats)]),
    "utree_rank_search_file_opts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(RankParams),
                                              C.c_int, C.c_int, C.POINTER(SearchStats)]),
    "utree_build_file": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(BuildStats)]),
    "utree_search_prepare": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "utree_search_file": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int,
                                    C.POINTER(SearchStats)]),
}


def kernel_source_sha256() -> str:
    """Content hash of the sources the search kernels are compiled from: a kept profile (profiles/traffic.json) is only valid
    for the library built from exactly these files (bench.py checks it)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels.hip", "lanes_kernel.hip", "lanes_core.hpp", "lanes_part.hip", "wave_common.hpp", "device_common.hpp", "utree_internal.h", "Makefile"):
        with open(os.path.join(_HERE, "csrc", f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read() + b"\0")
    return h.hexdigest()


class UtreeError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = load().utree_strerror(code).decode() if _LIB is not None else str(code)
        if _LIB is not None and code in (4, 5, 11):        # UTREE_E_NOMEM, UTREE_E_HIP, UTREE_E_DEVICE: which call, and what the runtime said
            hip = (_LIB.utree_last_hip_error() or b"").decode(errors="replace")
            if hip:
                msg += " [" + hip + "]"
        super().__init__("%s: %s (code %d)" % (what, msg, code))


def load():
    """Load libutree_amd.so. Raises (loudly) when it has not been built: no fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise ImportError("utree_amd: %s is missing -- build it with `make -C utree_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback" % SO_PATH)
        # torch ships its own libamdhip64.so.7 / librccl.so.1.  Loading it FIRST makes the dynamic loader
        # resolve this library's HIP/RCCL dependencies to those same copies (matching sonames), so the
        # process holds ONE HIP runtime and torch's device pointers / streams are valid in our calls.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)          # AttributeError if the .so does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def check(code, what=""):
    if code != OK:
        raise UtreeError(code, what)

"""ctypes binding of libtaxor_gpu.so (include/taxor_gpu.h).  There is no fallback: if the HIP library is
missing or does not load, importing anything that computes raises immediately."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libtaxor_gpu.so")
CSRC = os.path.join(_HERE, "csrc")


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-s"] + (["-B"] if force else [])
    subprocess.check_call(args)
    return SO_PATH


class IxfView(C.Structure):
    _fields_ = [("bins", C.c_uint64), ("stride", C.c_uint64), ("seg_len", C.c_uint64), ("seed", C.c_uint64),
                ("data", C.c_void_p), ("next_ixf", C.c_void_p), ("fname_idx", C.c_void_p), ("src_stride", C.c_uint64)]


IXF_READ_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p)


class IxfSource(C.Structure):
    _fields_ = [("read", IXF_READ_FN), ("ctx", C.c_void_p)]


class HixfView(C.Structure):
    _fields_ = [("n_ixf", C.c_uint64), ("ixf", C.POINTER(IxfView)), ("n_user_bins", C.c_uint64),
                ("kmer_size", C.c_uint8), ("syncmer_size", C.c_uint8), ("t_syncmer", C.c_uint8),
                ("use_syncmer", C.c_uint8), ("scaling", C.c_uint16), ("window_size", C.c_uint64), ("ixf_arith", C.c_uint32),
                ("source", C.c_void_p), ("ixf_layout", C.c_uint32)]


class ReadSegment(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("offsets", C.c_void_p), ("n_reads", C.c_uint64)]


class SearchParams(C.Structure):
    _fields_ = [("ratio", C.c_double), ("sub_batch_reads", C.c_uint32), ("sub_batch_bases", C.c_uint64),
                ("time_kernels", C.c_uint32), ("model", C.c_uint32), ("error_rate", C.c_double), ("flags", C.c_uint32)]


SEARCH_NO_PRUNE, SEARCH_GROUP_ALWAYS, SEARCH_NO_SMALL_PATH, SEARCH_SPLIT_ALWAYS, SEARCH_FORCE_TREE_STALL = 1, 2, 4, 8, 16


THR_PERCENTAGE, THR_SYNCMER, THR_KMER, THR_FRACMINHASH = 0, 1, 2, 3


class Results(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_tuples", C.c_uint64), ("read_off", C.POINTER(C.c_uint64)),
                ("user_bin", C.POINTER(C.c_int64)), ("count", C.POINTER(C.c_uint32)),
                ("n_hashes", C.POINTER(C.c_uint32))]


class RunStats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("n_hashes", C.c_uint64),
                ("n_tuples", C.c_uint64), ("n_work_items", C.c_uint64), ("algorithmic_bytes", C.c_uint64),
                ("query_bytes", C.c_uint64), ("query_touched_bytes", C.c_uint64), ("query_launches", C.c_uint32), ("query_ms", C.c_float),
                ("syncmer_ms", C.c_float), ("finalize_ms", C.c_float), ("total_ms", C.c_float),
                ("level_ms", C.c_float * 8), ("level_requested_bytes", C.c_uint64 * 8), ("level_row_reads", C.c_uint64 * 8),
                ("level_sparse_loads", C.c_uint64 * 8), ("tree_stalls_recovered", C.c_uint32)]


class CommStats(C.Structure):
    _fields_ = [("transport", C.c_int32), ("n_devices", C.c_uint32), ("index_bytes", C.c_uint64), ("index_upload_bytes", C.c_uint64),
                ("index_broadcast_bytes", C.c_uint64), ("index_seconds", C.c_double), ("gathers", C.c_uint64),
                ("gather_bytes", C.c_uint64), ("gather_seconds", C.c_double), ("index_broadcast_calls", C.c_uint64),
                ("self_exchange_bytes", C.c_uint64), ("rccl_version", C.c_int32), ("selftest_bytes", C.c_uint64)]


COMM_RCCL, COMM_HOST = 0, 1


class Species(C.Structure):
    _fields_ = [("organism_name", C.c_char_p), ("accession_id", C.c_char_p), ("taxid", C.c_char_p),
                ("taxnames_string", C.c_char_p), ("taxid_string", C.c_char_p), ("user_bin", C.c_uint64),
                ("seq_len", C.c_uint64)]


class HixfMeta(C.Structure):
    _fields_ = [("window_size", C.c_uint64), ("parts", C.c_uint8), ("compressed", C.c_uint8),
                ("n_species", C.c_uint64), ("species", C.POINTER(Species)), ("n_user_bin_filenames", C.c_uint64),
                ("user_bin_filenames", C.POINTER(C.c_char_p)), ("foreign_schema", C.c_uint8)]


class IxfSchema(C.Structure):
    _fields_ = [("n_before", C.c_uint32), ("n_after", C.c_uint32), ("idx_bins", C.c_int32), ("idx_stride", C.c_int32),
                ("idx_seg_len", C.c_int32), ("idx_seed", C.c_int32), ("seg_len_is_rows", C.c_uint32),
                ("default_seed", C.c_uint64), ("layout", C.c_uint32), ("len_unit", C.c_uint32), ("skip_before_len", C.c_uint32),
                ("skip_after_len", C.c_uint32)]


class IxfVariant(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("seg_len", C.c_uint64), ("stride", C.c_uint64), ("key_hash", C.c_uint8),
                ("seed_mode", C.c_uint8), ("rot", C.c_uint8), ("reduce", C.c_uint8), ("fp_mode", C.c_uint8),
                ("pad", C.c_uint8), ("layout", C.c_uint16)]


# layout codes (taxor_amd/csrc/ixf_layout.h)
LAYOUT_ROWS, LAYOUT_BIN_MAJOR, LAYOUT_BIT_SLICED = 0, 1, 2
LAYOUT_POSITION_MAJOR, LAYOUT_PITCH_BINS, LAYOUT_PITCH_STORED = 0x100, 0x200, 0x400

# every symbol include/taxor_gpu.h and include/taxor_gpu_tools.h declare: name -> (restype, argtypes)
_P = C.c_void_p
class BuildStats(C.Structure):
    _fields_ = [("keys_inserted", C.c_uint64), ("scratch_bytes", C.c_uint64), ("rounds_max", C.c_uint32), ("reseeds", C.c_uint32),
                ("chunks", C.c_uint32), ("reserved", C.c_uint32), ("seconds_peel", C.c_double), ("seconds_assign", C.c_double),
                ("seconds_union", C.c_double), ("seconds_total", C.c_double), ("seconds_release", C.c_double),
                ("seconds_count", C.c_double), ("seconds_rounds", C.c_double), ("seconds_upload", C.c_double),
                ("seconds_alloc", C.c_double), ("keys_counted_in_lds", C.c_uint64)]


SIGNATURES = {
    "taxor_gpu_last_error": (C.c_char_p, []),
    "taxor_gpu_index_create": (C.c_int, [C.POINTER(HixfView), C.c_int, C.POINTER(_P)]),
    "taxor_gpu_index_destroy": (None, [_P]),
    "taxor_gpu_index_build_hixf": (C.c_int, [_P, _P, _P, C.c_uint64, C.POINTER(C.c_uint32)]),
    "taxor_gpu_index_build_ixf_ex": (C.c_int, [_P, C.c_uint64, _P, C.c_int, _P, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(BuildStats)]),
    "taxor_gpu_index_build_hixf_ex": (C.c_int, [_P, _P, C.c_int, _P, C.c_uint64, C.POINTER(BuildStats)]),
    "taxor_gpu_index_build_hixf_gen": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, C.c_uint64, C.c_uint64, C.POINTER(BuildStats)]),
    "taxor_synth_key": (C.c_uint64, [C.c_uint64, C.c_uint64]),
    "taxor_gpu_synth_keys": (C.c_int, [C.c_int, _P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "taxor_gpu_malloc": (C.c_int, [C.c_int, C.c_uint64, C.POINTER(_P)]),
    "taxor_gpu_free": (None, [_P]),
    "taxor_gpu_memcpy_to_host": (C.c_int, [_P, _P, C.c_uint64]),
    "taxor_gpu_memcpy_from_host": (C.c_int, [_P, _P, C.c_uint64]),
    "taxor_gpu_index_ixf_seed": (C.c_uint64, [_P, C.c_uint64]),
    "taxor_gpu_index_data_bytes": (C.c_uint64, [_P]),
    "taxor_gpu_gather_ceiling": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "taxor_gpu_gather_ceiling_span": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                              C.POINTER(C.c_uint64)]),
    "taxor_gpu_gather_pattern": (C.c_int, [_P, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64)]),
    "taxor_gpu_index_leaf_runs": (C.c_uint64, [_P]),
    "taxor_gpu_index_depth": (C.c_uint32, [_P]),
    "taxor_gpu_index_fill_random": (C.c_int, [_P, C.c_uint64, C.c_uint64]),
    "taxor_gpu_index_upload_bin": (C.c_int, [_P, C.c_uint64, C.c_uint64, _P, C.c_uint64]),
    "taxor_gpu_index_download_ixf": (C.c_int, [_P, C.c_uint64, _P, C.c_uint64]),
    "taxor_gpu_index_build_ixf": (C.c_int, [_P, C.c_uint64, _P, _P, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "taxor_gpu_searcher_create": (C.c_int, [_P, C.POINTER(SearchParams), C.POINTER(_P)]),
    "taxor_gpu_searcher_destroy": (None, [_P]),
    "taxor_threshold_kind": (C.c_int, [C.c_int, C.c_uint32, C.c_uint64, C.c_double]),
    "taxor_threshold_model": (C.c_uint64, [C.c_int, C.c_uint64, C.c_uint32, C.c_double, C.c_double, C.c_double]),
    "taxor_threshold_select": (C.c_int, [C.POINTER(HixfView), C.c_double, C.c_double, C.POINTER(SearchParams)]),
    "taxor_gpu_host_register": (C.c_int, [_P, C.c_uint64]),
    "taxor_gpu_host_unregister": (C.c_int, [_P]),
    "taxor_gpu_search_batch_begin": (C.c_int, [_P, _P, _P, C.c_uint64]),
    "taxor_gpu_search_segments_begin": (C.c_int, [_P, C.POINTER(ReadSegment), C.c_uint64]),
    "taxor_gpu_search_batch_end": (C.c_int, [_P, C.POINTER(Results)]),
    "taxor_gpu_search_batch": (C.c_int, [_P, _P, _P, C.c_uint64, C.POINTER(Results)]),
    "taxor_gpu_batch_upload": (C.c_int, [_P, _P, _P, C.c_uint64]),
    "taxor_gpu_batch_run": (C.c_int, [_P]),
    "taxor_gpu_batch_sync": (C.c_int, [_P]),
    "taxor_gpu_batch_fetch": (C.c_int, [_P, C.POINTER(Results)]),
    "taxor_gpu_batch_result_sizes": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "taxor_gpu_batch_export_device": (C.c_int, [_P, _P, _P, _P, _P]),
    "taxor_gpu_batch_stats": (C.c_int, [_P, C.POINTER(RunStats)]),
    "taxor_gpu_comm_create": (C.c_int, [C.POINTER(C.c_int), C.c_uint32, C.c_int, C.POINTER(_P)]),
    "taxor_gpu_comm_destroy": (None, [_P]),
    "taxor_gpu_index_create_replicated": (C.c_int, [_P, C.POINTER(HixfView), C.POINTER(_P)]),
    "taxor_gpu_gather_results": (C.c_int, [_P, C.POINTER(_P), C.POINTER(Results)]),
    "taxor_gpu_comm_info": (C.c_int, [_P, C.POINTER(CommStats)]),
    "taxor_gpu_comm_set_self_exchange": (C.c_int, [_P, C.c_int]),
    # test and profiling entry points (include/taxor_gpu_tools.h)
    "taxor_gpu_phase_profile": (C.c_int, [_P, _P]),
    "taxor_gpu_syncmers": (C.c_int, [_P, _P, _P, C.c_uint64, C.POINTER(C.POINTER(C.c_uint64)),
                                     C.POINTER(C.POINTER(C.c_uint64))]),
    "taxor_gpu_ixf_bulk_count": (C.c_int, [_P, C.c_uint64, _P, C.c_uint64, _P]),
    "taxor_gpu_bulk_contains": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.POINTER(Results)]),
    "taxor_ixf_arith_code": (C.c_uint32, [C.POINTER(IxfVariant)]),
    "taxor_ixf_arith_decode": (None, [C.c_uint32, C.POINTER(IxfVariant)]),
    "taxor_ixf_build_bin_arith": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, _P]),
    "taxor_ixf_variant_default": (None, [C.POINTER(IxfVariant), C.c_uint64, C.c_uint64, C.c_uint64]),
    "taxor_gpu_ixf_variant_scan": (C.c_int, [C.c_int, _P, C.c_uint64, C.c_uint64, C.POINTER(IxfVariant), C.c_uint32, _P, _P, C.c_uint64, _P]),
    "taxor_ixf_layout_parse": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32)]),
    "taxor_ixf_layout_describe": (C.c_uint64, [C.c_uint32, C.c_char_p, C.c_uint64]),
    "taxor_hixf_set_layout": (C.c_int, [_P, C.c_uint32]),
    "taxor_hixf_ixf_raw_bytes": (C.c_uint64, [_P, C.c_uint64]),
    "taxor_ixf_variant_describe": (C.c_uint64, [C.POINTER(IxfVariant), C.c_char_p, C.c_uint64]),
    "taxor_hixf_load": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "taxor_ixf_schema_default": (None, [C.POINTER(IxfSchema)]),
    "taxor_hixf_probe": (C.c_int, [C.c_char_p, C.POINTER(IxfSchema), C.c_char_p, C.c_uint64]),
    "taxor_hixf_load_schema": (C.c_int, [C.c_char_p, C.POINTER(IxfSchema), C.POINTER(_P)]),
    "taxor_hixf_store_schema": (C.c_int, [C.c_char_p, C.POINTER(HixfView), C.POINTER(HixfMeta), C.POINTER(IxfSchema)]),
    "taxor_hixf_free": (None, [_P]),
    "taxor_hixf_release_data": (None, [_P]),
    "taxor_hixf_set_arith": (None, [_P, C.c_uint32]),
    "taxor_hixf_get_view": (C.POINTER(HixfView), [_P]),
    "taxor_hixf_get_meta": (C.POINTER(HixfMeta), [_P]),
    "taxor_hixf_store": (C.c_int, [C.c_char_p, C.POINTER(HixfView), C.POINTER(HixfMeta)]),
    "taxor_format_read": (C.c_uint64, [_P, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32, _P, _P, C.c_uint64,
                                       _P, C.c_uint64]),
    "taxor_format_reads": (C.c_uint64, [_P, C.c_uint64, _P, _P, _P, _P, _P, _P, _P, _P, C.c_uint64]),
    "taxor_threshold_ratio": (C.c_double, [C.c_uint32, C.c_double, C.c_double]),
    "taxor_threshold": (C.c_uint64, [C.c_uint64, C.c_double]),
    "taxor_classify_filter": (None, [_P, C.c_uint64, _P]),
    "taxor_ixf_seg_len": (C.c_uint64, [C.c_uint64]),
    "taxor_ixf_build_bin": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint64, _P]),
    "taxor_synth_reads": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, C.c_double,
                                    C.c_double, C.c_uint64, C.c_int, _P, C.c_uint64, _P, _P]),
}

_lib = None


def lib():
    """Load libtaxor_gpu.so; raise loudly if it is absent (no CPU path exists in the product)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C taxor_amd/csrc).  taxor_amd has no CPU fallback.")
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class TaxorError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[TAXOR SEARCH ERROR] {msg} (status {code})")
        self.code = code


def check(rc):
    if rc != 0:
        raise TaxorError(rc, lib().taxor_gpu_last_error().decode(errors="replace"))

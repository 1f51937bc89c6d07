"""ctypes loader for libpandora_amd.so (the C ABI declared in include/pandora_amd.h).

There is no CPU fallback: if the shared library is missing, loading raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PANDORA_AMD_LIB names another build of the same library (comparing two builds on one GPU box)
LIB_PATH = os.environ.get("PANDORA_AMD_LIB") or os.path.join(_HERE, "libpandora_amd.so")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)

PMAP_N_TIMERS = 8
PMAP_N_KERNELS = 16


class IndexDesc(C.Structure):
    _fields_ = [("k", C.c_uint32), ("n_keys", C.c_uint64), ("keys", u64p), ("rec_off", u64p),
                ("n_recs", C.c_uint64), ("rec_prg", u32p), ("rec_knode", u32p), ("rec_strand", u8p),
                ("path_off", u64p), ("iv_start", u32p), ("iv_len", u32p), ("n_prg", C.c_uint32),
                ("min_path_length", u32p)]


class IndexInfo(C.Structure):
    _fields_ = [("k", C.c_uint32), ("n_keys", C.c_uint64), ("n_recs", C.c_uint64), ("n_paths", C.c_uint64),
                ("table_slots", C.c_uint64), ("device_bytes", C.c_uint64), ("n_prg", C.c_uint32),
                ("filter_bits", C.c_uint32), ("filter_bits_per_key", C.c_uint32),
                ("filter_direct", C.c_uint32)]


class Params(C.Structure):
    _fields_ = [("w", C.c_uint32), ("k", C.c_uint32), ("max_diff", C.c_int32), ("fraction", C.c_float),
                ("min_cluster_size", C.c_uint32), ("expected_kmers", C.c_uint32), ("last_stage", C.c_uint32),
                ("profile", C.c_uint32), ("flags", C.c_uint32)]


class Counts(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("n_bases", C.c_uint64), ("n_minimizers", C.c_uint64),
                ("n_hits", C.c_uint64), ("n_clusters_all", C.c_uint64), ("n_clusters", C.c_uint64),
                ("ms", C.c_float * PMAP_N_TIMERS)]


class DevOut(C.Structure):
    _fields_ = [("hit_cap", C.c_uint32), ("hit_read", C.c_void_p), ("hit_key", C.c_void_p), ("hit_pos", C.c_void_p),
                ("hit_rank", C.c_void_p), ("hit_rec", C.c_void_p), ("cluster_cap", C.c_uint32),
                ("cluster_rows", C.c_void_p)]


class MapArgs(C.Structure):
    _fields_ = [("w", C.c_uint32), ("k", C.c_uint32), ("max_diff", C.c_int32), ("e_rate", C.c_float),
                ("min_cluster_size", C.c_uint32), ("genome_size", C.c_uint32), ("illumina", C.c_int),
                ("clean", C.c_int), ("max_covg", C.c_uint32), ("batch_bases", C.c_uint32),
                ("flags", C.c_uint32)]


class PangraphInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_reads", C.c_uint32), ("n_read_nodes", C.c_uint64),
                ("n_hits", C.c_uint64), ("n_add_node_calls", C.c_uint64)]


# name -> (restype, argtypes); every symbol include/pandora_amd.h declares
SIGNATURES = {
    "pmap_last_error": (C.c_char_p, []),
    "pmap_init": (C.c_int, [C.c_int]),
    "pmap_init_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "pmap_index_create": (C.c_int, [C.POINTER(IndexDesc), C.POINTER(C.c_void_p)]),
    "pmap_index_upload_now": (C.c_int, [C.c_void_p]),
    "pmap_index_load": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, u32p, C.POINTER(C.c_void_p)]),
    "pmap_gfa_min_path_length": (C.c_int, [C.c_char_p, u32p]),
    "pmap_index_info": (C.c_int, [C.c_void_p, C.POINTER(IndexInfo)]),
    "pmap_index_records": (C.c_int, [C.c_void_p, u64p, u32p, u32p, u8p, u32p, u64p]),
    "pmap_index_intervals": (C.c_int, [C.c_void_p, u32p, u32p, u64p]),
    "pmap_index_destroy": (None, [C.c_void_p]),
    "pmap_reads_create": (C.c_int, [C.c_char_p, u64p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "pmap_reads_repack": (C.c_int, [C.c_void_p]),
    "pmap_reads_info": (C.c_int, [C.c_void_p, u32p, u64p, u64p]),
    "pmap_reads_destroy": (None, [C.c_void_p]),
    "pmap_fraction": (C.c_float, [C.c_float, C.c_uint32]),
    "pmap_index_save_cache": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "pmap_index_load_cache": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "pmap_index_load_cached": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, u32p, C.c_char_p, C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_int)]),
    "pmap_map": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Params), C.POINTER(C.c_void_p)]),
    "pmap_map_async": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Params), C.POINTER(C.c_void_p)]),
    "pmap_map_wait": (C.c_int, [C.c_void_p]),
    "pmap_map_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Params), C.POINTER(DevOut), C.POINTER(Counts)]),
    "pmap_result_counts": (C.c_int, [C.c_void_p, C.POINTER(Counts)]),
    "pmap_result_kernel_ms": (C.c_int, [C.c_void_p, f32p]),
    "pmap_kernel_name": (C.c_char_p, [C.c_int]),
    "pmap_result_minimizers": (C.c_int, [C.c_void_p, u64p, u32p, u32p, u8p]),
    "pmap_result_hits": (C.c_int, [C.c_void_p, C.c_int, u32p, u32p, u32p, u32p, u8p]),
    "pmap_result_clusters": (C.c_int, [C.c_void_p, u64p, u32p, u32p, u32p, u8p, u32p, u32p, u32p]),
    "pmap_result_destroy": (None, [C.c_void_p]),
    "pmap_filter_clusters": (C.c_int, [C.c_uint64, u32p, u32p, u32p, u32p, u32p, u8p, u32p, u64p, u8p]),
    "pmap_read_fastaq": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(u64p), u32p,
                                   C.POINTER(C.c_void_p), C.POINTER(u64p)]),
    "pmap_free": (None, [C.c_void_p]),
    "pmap_read_fastaq_pieces": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(u64p), u32p]),
    "pmap_pangraph_from_read_file": (C.c_int, [C.c_char_p, C.c_void_p, C.POINTER(MapArgs),
                                               C.POINTER(C.c_void_p), u32p]),
    "pmap_pangraph_info": (C.c_int, [C.c_void_p, C.POINTER(PangraphInfo)]),
    "pmap_pangraph_nodes": (C.c_int, [C.c_void_p, u32p, u32p]),
    "pmap_pangraph_reads": (C.c_int, [C.c_void_p, u32p, u64p, u32p, u8p]),
    "pmap_pangraph_hits": (C.c_int, [C.c_void_p, u32p, u32p, u32p, u32p, u8p]),
    "pmap_pangraph_calls": (C.c_int, [C.c_void_p, u32p, u32p, u32p, u8p]),
    "pmap_pangraph_call_hits": (C.c_int, [C.c_void_p, u64p, u32p, u32p]),
    "pmap_pangraph_coverage": (C.c_int, [C.c_void_p, u32p, u32p]),
    "pmap_pangraph_destroy": (None, [C.c_void_p]),
    "pmap_release_host_cache": (C.c_uint64, []),
    "pmap_reads_update": (C.c_int, [C.c_void_p, C.c_char_p, u64p, C.c_uint32, C.c_uint32]),
    "pmap_pack_reads": (C.c_int, [C.c_char_p, u64p, C.c_uint32, u32p, u8p]),
    "pmap_reads_create_packed": (C.c_int, [u32p, u64p, u8p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "pmap_reads_update_packed": (C.c_int, [C.c_void_p, u32p, u64p, u8p, C.c_uint32, C.c_uint32]),
    "pmap_coverage_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "pmap_coverage_add": (C.c_int, [C.c_void_p, C.c_void_p, u32p]),
    "pmap_coverage_fetch": (C.c_int, [C.c_void_p, u32p, u32p]),
    "pmap_coverage_destroy": (None, [C.c_void_p]),
}

_lib = None


def load():
    """Loads libpandora_amd.so and binds every declared symbol; raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "pandora_amd: %s not found. Build it with `make -C pandora_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib

"""ctypes binding of librb_hip.so (the C ABI of include/rb_capi.h).

There is NO CPU fallback: if the HIP library is missing or cannot be loaded, importing this module
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "librb_hip.so")

RB_OK = 0
DBGBF, CBF, RPKBF, FPKBF = 0, 1, 2, 3
ADD_REVCOMP, ADD_COUNT_IF_PRESENT, ADD_STORE_READ_PAIRS, ADD_PAIRS_IF_PRESENT = 1, 2, 4, 8
OP_ADD, OP_ADD_IF_ABSENT, OP_ADD_COUNT_IF_PRESENT, OP_ADD_DBG_ONLY, OP_ADD_COUNT_ONLY, \
    OP_ADD_READ_PAIR, OP_ADD_FRAG_PAIR = range(7)
PROF_MAX = 32
SLOT_REC_KEYS, SLOT_REC_OCC, SLOT_PAIR_IDX, SLOT_DREQ_IDX, SLOT_DREQ_PROBE, SLOT_CREQ_IDX, SLOT_W_IDX, SLOT_W_VAL, \
    SLOT_CONF_EDGES, SLOT_CONF_RUNS, SLOT_CONF_OPS, SLOT_CW_IDX, SLOT_CW_VAL, SLOT_Q_BIDX, SLOT_Q_CIDX, SLOT_CACHE_UPD, SLOT_ORD_IDX = range(17)
MODE_ADD, MODE_COUNT_IF_PRESENT = 0, 2
STROBE_CANONICAL, STROBE_SLIDE = 1, 2


class GraphParams(C.Structure):
    _fields_ = [("dbgbf_bits", C.c_int64), ("cbf_bytes", C.c_int64), ("pkbf_bits", C.c_int64),
                ("dbgbf_num_hash", C.c_int32), ("cbf_num_hash", C.c_int32), ("pkbf_num_hash", C.c_int32),
                ("k", C.c_int32), ("stranded", C.c_int32), ("use_read_paired_kmers", C.c_int32),
                ("device", C.c_int32), ("group_bits", C.c_int32), ("rng_seed", C.c_uint64),
                ("max_batch_kmers", C.c_int64)]


class AddStats(C.Structure):
    _fields_ = [("reads", C.c_int64), ("kmers", C.c_int64), ("pairs", C.c_int64),
                ("distinct", C.c_int64), ("conflict_ops", C.c_int64), ("sorted_kmers", C.c_int64)]


class SynthParams(C.Structure):
    _fields_ = [("n_pairs", C.c_int64), ("genome_bases", C.c_int64), ("read_len", C.c_int32),
                ("frag_mean", C.c_int32), ("frag_sd", C.c_int32), ("sub_rate", C.c_float),
                ("n_rate", C.c_float), ("expr_sigma", C.c_float), ("seed", C.c_uint64),
                ("tx_min", C.c_int32), ("tx_max", C.c_int32), ("pair_offset", C.c_int64), ("total_pairs", C.c_int64)]


class Profile(C.Structure):
    _fields_ = [("n", C.c_int32), ("name", C.c_char_p * PROF_MAX), ("ms", C.c_double * PROF_MAX),
                ("launches", C.c_int64 * PROF_MAX)]


# every symbol include/rb_capi.h declares: (name, restype, argtypes)
_vp, _i64, _i32, _u64, _u32, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_uint32, C.c_size_t
SYMBOLS = [
    ("rb_last_error", C.c_char_p, []),
    ("rb_version", _i32, []),
    ("rb_build_id", C.c_char_p, []),
    ("rb_graph_create", _i32, [C.POINTER(GraphParams), C.POINTER(_vp)]),
    ("rb_graph_destroy", _i32, [_vp]),
    ("rb_graph_clear", _i32, [_vp, C.c_uint]),
    ("rb_graph_set_read_paired_kmer_distance", _i32, [_vp, _i32]),
    ("rb_graph_set_frag_paired_kmer_distance", _i32, [_vp, _i32]),
    ("rb_graph_init_fragment_pairs", _i32, [_vp, _i64, _i32]),
    ("rb_graph_get_op_ordinal", _i32, [_vp, C.POINTER(_u64)]),
    ("rb_graph_set_op_ordinal", _i32, [_vp, _u64]),
    ("rb_batch_create_ascii", _i32, [_i32, _vp, _vp, _vp, _i64, _i32, C.POINTER(_vp)]),
    ("rb_batch_destroy", _i32, [_vp]),
    ("rb_batch_info", _i32, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    ("rb_batch_download_ascii", _i32, [_vp, _i64, _i64, _vp, _vp]),
    ("rb_batch_create_synthetic", _i32, [_i32, C.POINTER(SynthParams), C.POINTER(_vp)]),
    ("rb_graph_add_batch", _i32, [_vp, _vp, C.c_uint, C.POINTER(AddStats)]),
    ("rb_graph_add_batch_range", _i32, [_vp, _vp, _i64, _i64, C.c_uint, C.POINTER(AddStats)]),
    ("rb_graph_add_pairs", _i32, [_vp, _vp, _i64, _i64, _i32, C.c_uint, C.POINTER(AddStats)]),
    ("rb_graph_add_fragments", _i32, [_vp, _vp, _i64, _i64, _i32, C.POINTER(AddStats)]),
    ("rb_graph_add_reads", _i32, [_vp, _vp, _vp, _vp, _i64, _i32, C.c_uint, C.POINTER(AddStats)]),
    ("rb_graph_apply", _i32, [_vp, _i32, _vp, _sz]),
    ("rb_graph_contains", _i32, [_vp, _vp, _sz, _vp]),
    ("rb_graph_count", _i32, [_vp, _vp, _sz, _vp]),
    ("rb_filter_lookup", _i32, [_vp, _i32, _vp, _sz, _vp]),
    ("rb_filter_lookup_then_add", _i32, [_vp, _i32, _vp, _sz, _vp]),
    ("rb_filter_get_count", _i32, [_vp, _vp, _sz, _vp]),
    ("rb_graph_kmers", _i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    ("rb_graph_batch_counts", _i32, [_vp, _vp, _i64, _i64, _vp, _vp, _i32, _vp]),
    ("rb_graph_neighbors", _i32, [_vp, _vp, _vp, _vp, _sz, _i32, _vp, _vp, _vp]),
    ("rb_graph_walk", _i32, [_vp, _vp, _vp, _sz, _i32, _i32, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("rb_graph_greedy_extend", _i32, [_vp, _vp, _vp, _sz, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    ("rb_graph_naive_extend", _i32, [_vp, _vp, _sz, _i32, _i32, _i32, _i32, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    ("rb_filter_size", _i32, [_vp, _i32, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32)]),
    ("rb_filter_popcount", _i32, [_vp, _i32, C.POINTER(_i64)]),
    ("rb_filter_fpr", _i32, [_vp, _i32, C.POINTER(C.c_float)]),
    ("rb_filter_fold", _i32, [_vp, _i32, C.POINTER(_u64)]),
    ("rb_filter_export", _i32, [_vp, _i32, _vp, _sz]),
    ("rb_filter_import", _i32, [_vp, _i32, _vp, _sz]),
    ("rb_expected_size", _i64, [_i64, C.c_float, _i32]),
    ("rb_graph_destroy_filter", _i32, [_vp, _i32]),
    ("rb_filter_increment_and_get", _i32, [_vp, _vp, _sz, _vp]),
    ("rb_cbf_to_bloom", _i32, [_vp, C.c_float, _vp, _i32]),
    ("rb_nthash_batch", _i32, [_vp, _i32, _i32, _i64, _i64, C.POINTER(_i64), _vp, _vp, _vp]),
    ("rb_minimizers", _i32, [_i32, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    ("rb_strobemers", _i32, [_i32, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    ("rb_randstrobes", _i32, [_i32, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    ("rb_strobe3", _i32, [_i32, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    ("rb_kmer_pair_hashes", _i32, [_i32, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    ("rb_minimizers_next", _i32, [_i32, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    ("rb_minimizer_set", _i32, [_i32, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    ("rb_fastq_split", _i32, [_vp, _sz, _i32, _vp, _vp, _vp, _i64, C.POINTER(_i64)]),
    ("rb_gunzip", _i32, [_vp, _sz, _i32, _vp, _sz, C.POINTER(_sz)]),
    ("rb_batch_create_fastq", _i32, [_i32, _vp, _sz, _i32, _i32, _i32, C.POINTER(_vp), C.POINTER(_sz)]),
    ("rb_batch_create_fasta", _i32, [_i32, _vp, _sz, _i32, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_i32)]),
    ("rb_graph_add_fasta", _i32, [_vp, _vp, _sz, C.c_uint, C.POINTER(AddStats), C.POINTER(_i64)]),
    ("rb_graph_add_fastq", _i32, [_vp, _vp, _sz, _i32, C.c_uint, C.POINTER(AddStats), C.POINTER(_i64)]),
    ("rb_graph_add_fastq_file", _i32, [_vp, C.c_char_p, _i32, C.c_uint, C.POINTER(AddStats), C.POINTER(_i64)]),
    ("rb_graph_add_fasta_file", _i32, [_vp, C.c_char_p, C.c_uint, C.POINTER(AddStats), C.POINTER(_i64)]),
    ("rb_batch_create_nbits", _i32, [_i32, _vp, _sz, _i64, C.POINTER(_vp), C.POINTER(_sz)]),
    ("rb_nbits_encode", _i32, [_vp, _vp, _i64, _vp, _sz, C.POINTER(_sz)]),
    ("rb_host_alloc", _i32, [_sz, C.POINTER(_vp)]),
    ("rb_host_free", _i32, [_vp]),
    ("rb_batch_download_packed", _i32, [_vp, _i64, _i64, _vp, _vp, _vp, C.POINTER(_i64)]),
    ("rb_packed_stream_create", _i32, [_i32, _i64, _i64, C.POINTER(_vp)]),
    ("rb_packed_stream_begin", _i32, [_vp, _vp, _vp, _vp, _i64, _i64]),
    ("rb_packed_stream_finish", _i32, [_vp, C.POINTER(_vp)]),
    ("rb_packed_stream_destroy", _i32, [_vp]),
    ("rb_graph_add_packed", _i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, C.c_uint, C.POINTER(AddStats)]),
    ("rb_graph_prefetch_packed", _i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64]),
    ("rb_graph_create_shard", _i32, [C.POINTER(GraphParams), _i32, _i32, C.POINTER(_vp)]),
    ("rb_shard_set_cache_replication", _i32, [_vp, _i32]),
    ("rb_shard_hash", _i32, [_vp, _vp, _i64, _i64, _i64, _i64, _u64, _u32, C.c_uint, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(AddStats)]),
    ("rb_shard_group", _i32, [_vp, _vp, _vp, _i64, _u64, _u32, C.c_uint, C.POINTER(_i64), C.POINTER(_i64)]),
    ("rb_shard_cache_apply", _i32, [_vp, _vp, _i64]),
    ("rb_shard_hash_begin", _i32, [_vp, _vp, _i64, _i64, _u64, _u32, C.c_uint]),
    ("rb_shard_hash_emit", _i32, [_vp]),
    ("rb_shard_hash_group", _i32, [_vp, _vp, _i64, _i64, _i64, _i64, _u64, _u32, C.c_uint, C.POINTER(_i64), C.POINTER(_i64),
                             C.POINTER(_i64), C.POINTER(AddStats)]),
    ("rb_shard_serve", _i32, [_vp, _i32, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    ("rb_shard_resolve", _i32, [_vp, _i32, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(AddStats)]),
    ("rb_shard_order_serve", _i32, [_vp, _vp, _i64, _vp]),
    ("rb_shard_order_finish", _i32, [_vp, _i32, _vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(AddStats)]),
    ("rb_shard_apply_tagged", _i32, [_vp, _vp, _i64]),
    ("rb_shard_apply_writes", _i32, [_vp, _vp, _vp, _i64]),
    ("rb_shard_conflict_route", _i32, [_vp, _vp, _i64, _i64, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(AddStats)]),
    ("rb_shard_conflict_replay", _i32, [_vp, _vp, _i64, _vp, _i64, C.POINTER(_i64)]),
    ("rb_shard_take", _i32, [_vp, _i32, _vp, _i64]),
    ("rb_shard_slot", _i32, [_vp, _i32, C.POINTER(_vp), C.POINTER(_i64)]),
    ("rb_shard_query_make", _i32, [_vp, _i32, _i32, _vp, _sz, C.POINTER(_i64), C.POINTER(_i64)]),
    ("rb_shard_query_serve", _i32, [_vp, _i32, _vp, _i64, _vp, _i64, _vp, _vp]),
    ("rb_shard_query_finish", _i32, [_vp, _i32, _vp, _vp, _vp, _vp]),
    ("rb_shard_trav_begin", _i32, [_vp, _i32, _vp, _vp, _sz, _i32, _i32, _i32, _i32, C.c_float, _vp, _vp, _i32]),
    ("rb_shard_trav_set_gate", _i32, [_vp, _vp]),
    ("rb_shard_trav_advance", _i32, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    ("rb_shard_trav_absorb", _i32, [_vp, _vp, _vp, _vp]),
    ("rb_shard_trav_end", _i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i64)]),
    ("rb_shard_comm_unique_id", _i32, [_vp]),
    ("rb_shard_comm_create_rccl", _i32, [_vp, _i32, _i32, _i32, C.POINTER(_vp)]),
    ("rb_shard_comm_create_loopback", _i32, [_i32, C.POINTER(_vp)]),
    ("rb_shard_comm_destroy", _i32, [_vp]),
    ("rb_shard_comm_selftest", _i32, [_vp, _i32, _i32, _i64]),
    ("rb_shard_hash_begin_split", _i32, [_vp, _vp, _i64, _i64, _i64, _i64, C.c_uint64, C.c_uint32, C.c_uint]),
    ("rb_shard_pairs_flush_begin", _i32, [_vp, C.POINTER(_vp), C.POINTER(_i64)]),
    ("rb_shard_pairs_flush_end", _i32, [_vp, _vp, C.POINTER(_i64)]),
    ("rb_shard_add_range", _i32, [_vp, _vp, _vp, _i64, _i64, C.c_uint, _i64, C.c_uint32, C.c_uint64, C.POINTER(AddStats)]),
    ("rb_shard_span", _i32, [_vp, _i32, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    ("rb_debug_probe_cbf", _i32, [_vp, _i32, C.POINTER(C.c_float)]),
    ("rb_debug_scan_u32", _i32, [_i32, _vp, _sz, _vp, _i32]),
    ("rb_debug_sort_pairs", _i32, [_i32, _vp, _vp, _i32, _sz, _i32, _i32, _i32, _i32]),
    ("rb_graph_profile_enable", _i32, [_vp, _i32]),
    ("rb_graph_profile_get", _i32, [_vp, C.POINTER(Profile), _i32]),
]


class NativeError(RuntimeError):
    """Counterpart of the RuntimeException the reference's workers raise (R/RNABloom.java:903-905)."""

    def __init__(self, code, msg):
        super().__init__("librb_hip error %d: %s" % (code, msg))
        self.code = code


def load():
    # PyTorch (the multi-GPU driver's collectives) ships its own HIP / HSA runtime.  Two runtimes in one process
    # cannot both open the device: loading librb_hip.so (which binds to /opt/rocm's) and importing torch afterwards
    # ends in "no ROCm-capable device is detected" at the first HIP call.  So when torch is installed it is imported
    # first and librb_hip.so binds to the runtime that is already loaded.  (A process that never imports torch —
    # the reference's JVM through the JNI stub of INTEGRATION.md — can set RB_NO_TORCH_PRELOAD=1.)
    if not os.environ.get("RB_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError("HIP library %s not built — run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()


def check(rc):
    if rc != RB_OK:
        raise NativeError(rc, (lib.rb_last_error() or b"").decode(errors="replace"))

"""ctypes binding of libvclust_gpu.so (include/vclust_gpu.h).

The library is the product: there is no Python or CPU fallback.  If the shared object is
missing the import fails loudly and tells how to build it.
"""
import ctypes as C
import pathlib

PKG_DIR = pathlib.Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / 'libvclust_gpu.so'


class VclustGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'libvclust_gpu error {code}: {msg}')
        self.code = code


class PairCount(C.Structure):
    _fields_ = [('a', C.c_uint32), ('b', C.c_uint32), ('shared', C.c_uint32)]


class Task(C.Structure):
    _fields_ = [('q', C.c_uint32), ('r', C.c_uint32)]


class PairStat(C.Structure):
    _fields_ = [('n_match', C.c_uint32), ('aln_len', C.c_uint32), ('n_regions', C.c_uint32)]


class Region(C.Structure):
    _fields_ = [('task', C.c_uint32), ('qstart', C.c_int32), ('qend', C.c_int32),
                ('rstart', C.c_int32), ('rend', C.c_int32), ('n_match', C.c_int32)]


class LzParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('mal', 'msl', 'mrd', 'mqd', 'reg', 'aw', 'am', 'ar')]


class LzFit(C.Structure):
    _fields_ = [('weak_seed_ratio', C.c_int), ('anchor_margin', C.c_int), ('seed_choice', C.c_int)]


class PrefilterParams(C.Structure):
    _fields_ = [('k', C.c_int), ('min_kmers', C.c_int), ('min_ident', C.c_double),
                ('batch_size', C.c_int), ('kmers_fraction', C.c_double), ('max_seqs', C.c_int),
                ('num_threads', C.c_int), ('verbosity', C.c_int), ('is_multifasta', C.c_int)]


class AlignParams(C.Structure):
    _fields_ = [('lz', LzParams),
                ('out_tani', C.c_double), ('out_gani', C.c_double), ('out_ani', C.c_double),
                ('out_qcov', C.c_double), ('out_rcov', C.c_double),
                ('filter_path', C.c_char_p), ('filter_threshold', C.c_double),
                ('out_aln_path', C.c_char_p),
                ('out_columns', C.POINTER(C.c_char_p)), ('n_out_columns', C.c_int),
                ('num_threads', C.c_int), ('verbosity', C.c_int), ('is_multifasta', C.c_int)]


class KernelTime(C.Structure):
    _fields_ = [('name', C.c_char * 48), ('total_ms', C.c_double), ('launches', C.c_int64),
                ('bytes', C.c_double)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)

# every symbol include/vclust_gpu.h declares: (restype, argtypes)
P = C.POINTER
SYMBOLS = {
    'vg_version': (C.c_char_p, []),
    'vg_last_error': (C.c_char_p, []),
    'vg_free': (None, [C.c_void_p]),
    'vg_device_count': (C.c_int, []),
    'vg_set_device': (C.c_int, [C.c_int]),
    'vg_release_device_memory': (None, []),
    'vg_copy': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    'vg_alloc_selftest': (C.c_int, [C.POINTER(C.c_int64), C.c_int, C.c_int]),
    'vg_genomes_load': (C.c_int, [P(C.c_char_p), C.c_int, C.c_int, C.c_int, P(C.c_void_p)]),
    'vg_genomes_from_codes': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, P(C.c_char_p), P(C.c_void_p)]),
    'vg_genomes_free': (None, [C.c_void_p]),
    'vg_genomes_count': (C.c_int, [C.c_void_p]),
    'vg_genomes_total_len': (C.c_int64, [C.c_void_p]),
    'vg_genomes_lengths': (C.c_int, [C.c_void_p, P(C.c_int64)]),
    'vg_genomes_name': (C.c_char_p, [C.c_void_p, C.c_int]),
    'vg_genomes_codes': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    'vg_genomes_to_device': (C.c_int, [C.c_void_p]),
    'vg_kmer_shared': (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_uint32,
                                 P(C.c_int64), P(P(PairCount)), P(C.c_int64)]),
    'vg_kmer_set': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, P(P(C.c_uint64)), P(C.c_int64)]),
    'vg_filter_pairs': (C.c_int, [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int64), C.c_int64, C.POINTER(PairCount), C.c_int64,
                                  C.POINTER(C.POINTER(PairCount)), C.POINTER(C.c_int64)]),
    'vg_write_fltr': (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int,
                                P(C.c_int64), P(PairCount), C.c_int64, C.c_char_p]),
    'vg_prefilter': (C.c_int, [P(C.c_char_p), C.c_int, C.c_char_p, P(PrefilterParams)]),
    'vg_set_process_ends_after_call': (None, [C.c_int]),
    'vg_set_lz_fit': (None, [P(LzFit)]),
    'vg_lz_align': (C.c_int, [C.c_void_p, P(Task), C.c_int64, P(LzParams), P(PairStat),
                              P(P(Region)), P(C.c_int64)]),
    'vg_align_order': (C.c_int, [C.c_void_p, P(C.c_int32)]),
    'vg_read_filter': (C.c_int, [C.c_void_p, C.c_char_p, C.c_double, P(P(PairCount)), P(C.c_int64)]),
    'vg_align_tasks': (C.c_int, [C.c_void_p, P(PairCount), C.c_int64, P(P(Task)), P(C.c_int64)]),
    'vg_lz_prepare': (C.c_int, [C.c_void_p, P(PairCount), C.c_int64, P(LzParams)]),
    'vg_set_index_budget': (None, [C.c_int64]),
    'vg_set_subshards': (None, [C.c_int]),
    'vg_set_range_scan': (None, [C.c_int]),
    'vg_set_placement_trials': (None, [C.c_int]),
    'vg_write_ani': (C.c_int, [C.c_void_p, P(Task), P(PairStat), C.c_int64, P(Region), C.c_int64,
                               C.c_char_p, P(AlignParams)]),
    'vg_align': (C.c_int, [P(C.c_char_p), C.c_int, C.c_char_p, P(AlignParams)]),
    'vg_comm_create': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, P(C.c_void_p)]),
    'vg_rccl_unique_id': (C.c_int, [C.c_void_p, C.c_int64]),
    'vg_comm_rccl_create': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int64, P(C.c_void_p)]),
    'vg_comm_free': (None, [C.c_void_p]),
    'vg_comm_kind': (C.c_int, [C.c_void_p]),
    'vg_comm_rccl_ranks': (C.c_int, [C.c_void_p]),
    'vg_comm_rank': (C.c_int, [C.c_void_p]),
    'vg_comm_world': (C.c_int, [C.c_void_p]),
    'vg_comm_selftest': (C.c_int, [C.c_void_p, C.c_int64]),
    'vg_kmer_shared_sharded': (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_uint32, C.c_void_p, P(C.c_int64), P(P(PairCount)), P(C.c_int64)]),
    'vg_align_owner': (C.c_int, [P(Task), C.c_int64, C.c_int, C.c_int, P(C.c_int32)]),
    'vg_align_pairs_share': (C.c_int, [C.c_void_p, P(PairCount), C.c_int64, C.c_int, C.c_int, P(P(Task)), P(C.c_int64)]),
    'vg_lz_align_pairs_sharded': (C.c_int, [C.c_void_p, P(PairCount), C.c_int64, P(LzParams), C.c_void_p, P(P(Task)), P(C.c_int64), P(P(PairStat))]),
    'vg_lz_align_sharded': (C.c_int, [C.c_void_p, P(Task), C.c_int64, P(LzParams), C.c_void_p, P(PairStat), P(P(Region)), P(C.c_int64)]),
    'vg_prefilter_sharded': (C.c_int, [P(C.c_char_p), C.c_int, C.c_char_p, P(PrefilterParams), C.c_void_p]),
    'vg_align_sharded': (C.c_int, [P(C.c_char_p), C.c_int, C.c_char_p, P(AlignParams), C.c_void_p]),
    'vg_synth_plan': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_double, C.c_int,
                                C.c_int, P(C.c_void_p), P(C.c_void_p), P(C.c_int64)]),
    'vg_profile_enable': (None, [C.c_int]),
    'vg_profile_reset': (None, []),
    'vg_profile_get': (C.c_int, [P(KernelTime), C.c_int]),
}

_lib = None


def load():
    """Return the loaded library, binding all prototypes on first use."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f'{LIB_PATH} is missing: the HIP extension is the product and has no fallback. '
            'Build it with `python -m vclust_amd.build` (needs hipcc).')
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)       # AttributeError if the .so does not export the symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise VclustGpuError(rc, load().vg_last_error().decode('utf-8', 'replace'))


def hip_copy(dst, src, nbytes, to_host):
    check(load().vg_copy(C.c_void_p(int(dst) if not isinstance(dst, C.c_void_p) else dst.value), C.c_void_p(int(src) if not isinstance(src, C.c_void_p) else src.value),
                         int(nbytes), int(bool(to_host))))

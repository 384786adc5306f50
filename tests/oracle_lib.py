"""ctypes binding of oracle/_build/liboracle.so — TEST INFRASTRUCTURE ONLY.

The CPU oracle is the checker of the HIP product path; nothing outside tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may load it.
"""
import ctypes as C
import pathlib
import subprocess

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / 'oracle'
LIB = ORACLE_DIR / '_build' / 'liboracle.so'
CLI = ORACLE_DIR / '_build' / 'oracle_cli'


class Genome(C.Structure):
    _fields_ = [('name', C.c_char_p), ('seq', C.POINTER(C.c_uint8)), ('len', C.c_int64), ('n_parts', C.c_int32)]


class GenomeSet(C.Structure):
    _fields_ = [('g', C.POINTER(Genome)), ('n', C.c_int32), ('cap', C.c_int32)]


class LzParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('mal', 'msl', 'mrd', 'mqd', 'reg', 'aw', 'am', 'ar')]


class PairStat(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ('q', 'r', 'n_match', 'aln_len', 'n_regions')]


class PairCount(C.Structure):
    _fields_ = [('a', C.c_uint32), ('b', C.c_uint32), ('shared', C.c_uint32)]


_lib = None


def build():
    subprocess.run(['make', '-C', str(ORACLE_DIR)], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not LIB.exists():
            build()
        L = C.CDLL(str(LIB))
        if not hasattr(L, 'vo_last_stage_cpu'):          # a library built from older sources
            build()
            L = C.CDLL(str(LIB))
        L.vo_read_fasta.argtypes = [C.c_char_p, C.c_int, C.POINTER(GenomeSet)]
        L.vo_read_fasta.restype = C.c_int
        L.vo_free_genomes.argtypes = [C.POINTER(GenomeSet)]
        L.vo_kmer_set_f.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_double, C.POINTER(C.POINTER(C.c_uint64))]
        L.vo_kmer_set_f.restype = C.c_int64
        L.vo_shared_all.argtypes = [C.POINTER(GenomeSet), C.c_int, C.c_double, C.POINTER(C.c_int64),
                                    C.POINTER(C.POINTER(PairCount)), C.POINTER(C.c_int64)]
        L.vo_shared_all.restype = C.c_int
        L.vo_lz_pair_stat.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(LzParams),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.vo_lz_pair_stat.restype = C.c_int
        L.vo_path_rows.argtypes = [C.POINTER(GenomeSet), C.c_int, C.c_int, C.c_double, C.POINTER(LzParams),
                                   C.POINTER(C.POINTER(PairStat)), C.POINTER(C.c_int64)]
        L.vo_path_rows.restype = C.c_int
        L.vo_shared_all_mt.argtypes = [C.POINTER(GenomeSet), C.c_int, C.c_double, C.POINTER(C.c_int64),
                                       C.POINTER(C.POINTER(PairCount)), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.vo_shared_all_mt.restype = C.c_int
        L.vo_path_rows_mt.argtypes = [C.POINTER(GenomeSet), C.c_int, C.c_int, C.c_double, C.POINTER(LzParams),
                                      C.POINTER(C.POINTER(PairStat)), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.vo_path_rows_mt.restype = C.c_int
        L.vo_set_threads.argtypes = [C.c_int]
        L.vo_set_threads.restype = None
        L.vo_last_stage_cpu.argtypes = [C.POINTER(C.c_double)]
        L.vo_last_stage_cpu.restype = None
        L.vo_fmt_num.argtypes = [C.c_double, C.c_char_p]
        L.vo_fmt_num.restype = C.c_int
        L.vo_fmt_len_ratio.argtypes = [C.c_int64, C.c_int64, C.c_char_p]
        L.vo_ani_shorter.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int]
        L.vo_ani_shorter.restype = C.c_double
        L.free = C.CDLL(None).free
        L.free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


DEFAULT_LZ = dict(mal=11, msl=7, mrd=40, mqd=40, reg=35, aw=15, am=7, ar=3)


def fmt_num(x):
    buf = C.create_string_buffer(64)
    lib().vo_fmt_num(float(x), buf)
    return buf.value.decode()


def kmer_set(codes, k=25, fraction=1.0):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    p = C.POINTER(C.c_uint64)()
    n = lib().vo_kmer_set_f(codes.ctypes.data_as(C.c_void_p), len(codes), k, float(fraction), C.byref(p))
    out = np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].copy()
    lib().free(C.cast(p, C.c_void_p))
    return out


def _make_set(codes, offsets):
    n = len(offsets) - 1
    arr = (Genome * max(n, 1))()
    keep = []
    for i in range(n):
        seg = np.ascontiguousarray(codes[offsets[i]:offsets[i + 1]], dtype=np.uint8)
        keep.append(seg)
        arr[i].name = f'g{i}'.encode()
        arr[i].seq = seg.ctypes.data_as(C.POINTER(C.c_uint8))
        arr[i].len = len(seg)
        arr[i].n_parts = 1
    gs = GenomeSet(arr, n, n)
    return gs, (arr, keep)


def shared_all(codes, offsets, k=25, fraction=1.0):
    """-> (set_sizes, dict[(a,b)] = shared) with a > b."""
    gs, keep = _make_set(codes, offsets)
    n = gs.n
    sizes = (C.c_int64 * max(n, 1))()
    pp = C.POINTER(PairCount)()
    npairs = C.c_int64()
    lib().vo_shared_all(C.byref(gs), k, float(fraction), sizes, C.byref(pp), C.byref(npairs))
    d = {(pp[i].a, pp[i].b): pp[i].shared for i in range(npairs.value)}
    lib().free(C.cast(pp, C.c_void_p))
    return np.array(sizes[:n], dtype=np.int64), d


def shared_all_mt(codes, offsets, k=25, fraction=1.0, threads=None):
    """The multithreaded form (bench.py's cpu_baseline leg) -> (set_sizes, dict, stage seconds [sets, index, pair count], threads)."""
    gs, keep = _make_set(codes, offsets)
    n = gs.n
    if threads:
        lib().vo_set_threads(int(threads))
    sizes = (C.c_int64 * max(n, 1))()
    pp = C.POINTER(PairCount)(); npairs = C.c_int64(); st = (C.c_double * 3)(); thr = C.c_int()
    lib().vo_shared_all_mt(C.byref(gs), k, float(fraction), sizes, C.byref(pp), C.byref(npairs), st, C.byref(thr))
    d = {(pp[i].a, pp[i].b): pp[i].shared for i in range(npairs.value)}
    assert len(d) == npairs.value, 'a pair was emitted twice'
    lib().free(C.cast(pp, C.c_void_p))
    return np.array(sizes[:n], dtype=np.int64), d, list(st), thr.value


def lz_pair_stat(q, r, lz=None):
    prm = LzParams(**{**DEFAULT_LZ, **(lz or {})})
    q = np.ascontiguousarray(q, dtype=np.uint8)
    r = np.ascontiguousarray(r, dtype=np.uint8)
    m, a, n = C.c_uint32(), C.c_uint32(), C.c_uint32()
    lib().vo_lz_pair_stat(q.ctypes.data_as(C.c_void_p), len(q), r.ctypes.data_as(C.c_void_p), len(r),
                          C.byref(prm), C.byref(m), C.byref(a), C.byref(n))
    return m.value, a.value, n.value


def path_rows(codes, offsets, k=25, min_kmers=20, min_ident=0.7, lz=None, threads=None):
    """Whole path in memory -> structured array (q, r, n_match, aln_len, n_regions), ids in input order."""
    import os
    if threads:
        os.environ['OMP_NUM_THREADS'] = str(threads)
    gs, keep = _make_set(codes, offsets)
    prm = LzParams(**{**DEFAULT_LZ, **(lz or {})})
    pp = C.POINTER(PairStat)(); n = C.c_int64()
    lib().vo_path_rows(C.byref(gs), k, min_kmers, float(min_ident), C.byref(prm), C.byref(pp), C.byref(n))
    dt = np.dtype([('q', '<u4'), ('r', '<u4'), ('n_match', '<u4'), ('aln_len', '<u4'), ('n_regions', '<u4')])
    out = np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_uint32)), shape=(max(n.value, 1) * 5,))[:n.value * 5].copy().view(dt)
    lib().free(C.cast(pp, C.c_void_p))
    return out


def read_fasta_codes(path, multisample=True):
    """-> (codes, offsets, names) through the oracle's own FASTA reader."""
    gs = GenomeSet()
    if lib().vo_read_fasta(str(path).encode(), int(multisample), C.byref(gs)) != 0:
        raise IOError(path)
    seqs, names = [], []
    for i in range(gs.n):
        n = gs.g[i].len
        seqs.append(np.ctypeslib.as_array(gs.g[i].seq, shape=(n,)).copy() if n else np.zeros(0, dtype=np.uint8))
        names.append(gs.g[i].name.decode())
    lib().vo_free_genomes(C.byref(gs))
    offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(s) for s in seqs])
    return np.concatenate(seqs), offsets, names


def run_cli(*args, env=None):
    if not CLI.exists():
        build()
    subprocess.run([str(CLI), *map(str, args)], check=True, env=env)


def path_rows_mt(codes, offsets, k=25, min_kmers=20, min_ident=0.7, lz=None, threads=None):
    """vo_path_rows_mt: every stage on all host threads -> (rows, {stage: seconds}, threads that ran)."""
    gs, keep = _make_set(codes, offsets)
    if threads:
        lib().vo_set_threads(int(threads))
    prm = LzParams(**{**DEFAULT_LZ, **(lz or {})})
    pp = C.POINTER(PairStat)(); n = C.c_int64(); st = (C.c_double * 4)(); thr = C.c_int()
    lib().vo_path_rows_mt(C.byref(gs), k, min_kmers, float(min_ident), C.byref(prm), C.byref(pp), C.byref(n), st, C.byref(thr))
    dt = np.dtype([('q', '<u4'), ('r', '<u4'), ('n_match', '<u4'), ('aln_len', '<u4'), ('n_regions', '<u4')])
    out = np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_uint32)), shape=(max(n.value, 1) * 5,))[:n.value * 5].copy().view(dt)
    lib().free(C.cast(pp, C.c_void_p))
    return out, dict(zip(('sets', 'index', 'pair_count', 'lz'), (round(float(x), 3) for x in st))), thr.value


def last_stage_cpu():
    """CPU seconds (all threads) of the stages of the last *_mt call: {stage: seconds}."""
    c = (C.c_double * 4)()
    lib().vo_last_stage_cpu(c)
    return dict(zip(('sets', 'index', 'pair_count', 'lz'), (round(float(x), 3) for x in c)))

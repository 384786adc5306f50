"""Thin Python layer over the C ABI: device residency, the integer kernels and the writers.

Used by the vclust-compatible front-end (vclust_amd/cli.py), bench.py and the parity tests.
Everything computational happens inside libvclust_gpu.so.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import (AlignParams, KernelTime, LzParams, PairCount, PairStat, PrefilterParams, Region,
                   Task, check)

ALIGN_FIELDS = ['qidx', 'ridx', 'query', 'reference', 'tani', 'gani', 'ani', 'qcov', 'rcov',
                'num_alns', 'len_ratio', 'qlen', 'rlen', 'nt_match', 'nt_mismatch']

PAIR_DTYPE = np.dtype([('a', '<u4'), ('b', '<u4'), ('shared', '<u4')])
TASK_DTYPE = np.dtype([('q', '<u4'), ('r', '<u4')])
STAT_DTYPE = np.dtype([('n_match', '<u4'), ('aln_len', '<u4'), ('n_regions', '<u4')])
REGION_DTYPE = np.dtype([('task', '<u4'), ('qstart', '<i4'), ('qend', '<i4'), ('rstart', '<i4'),
                         ('rend', '<i4'), ('n_match', '<i4')])

from .stages import DEFAULT_LZ  # noqa: E402


def version():
    return _lib.load().vg_version().decode()


def device_count():
    return _lib.load().vg_device_count()


def set_device(idx):
    check(_lib.load().vg_set_device(int(idx)))


class _LibOwned:
    """Releases a library-owned buffer (vg_free) when the numpy array that views it is collected."""
    __slots__ = ('_ptr', '_free')

    def __init__(self, ptr, free):
        self._ptr, self._free = ptr, free

    def __del__(self):
        try:
            self._free(self._ptr)
        except Exception:
            pass


def _take(ptr, n, dtype):
    """A library-owned array as a numpy array WITHOUT a copy: the array views the buffer, which is handed back to
    the library (vg_free) when the last view of it is gone."""
    lib = _lib.load()
    if n <= 0:
        if ptr:
            lib.vg_free(C.cast(ptr, C.c_void_p))
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(C.addressof(ptr.contents))
    buf._owner = _LibOwned(C.cast(ptr, C.c_void_p), lib.vg_free)      # lives exactly as long as the buffer object
    return np.frombuffer(buf, dtype=dtype, count=n)


class GenomeSet:
    """A set of genomes held by the library (host packed + HBM resident)."""

    def __init__(self, handle):
        self._h = handle
        self._lib = _lib.load()

    @classmethod
    def load(cls, paths, multisample, n_threads=1):
        lib = _lib.load()
        arr = (C.c_char_p * len(paths))(*[os.fsencode(str(p)) for p in paths])
        h = C.c_void_p()
        check(lib.vg_genomes_load(arr, len(paths), int(bool(multisample)), int(n_threads), C.byref(h)))
        return cls(h)

    @classmethod
    def from_codes(cls, codes, offsets, names=None):
        """codes: uint8 array (0..3 ACGT, >3 N); offsets: int64 array of n+1 entries."""
        lib = _lib.load()
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        nm = None
        if names is not None:
            nm = (C.c_char_p * n)(*[s.encode() for s in names])
        h = C.c_void_p()
        check(lib.vg_genomes_from_codes(codes.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.c_void_p),
                                        n, nm, C.byref(h)))
        return cls(h)

    def close(self):
        if self._h:
            self._lib.vg_genomes_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self._lib.vg_genomes_count(self._h)

    @property
    def total_len(self):
        return self._lib.vg_genomes_total_len(self._h)

    def lengths(self):
        out = np.zeros(len(self), dtype=np.int64)
        if len(self):
            check(self._lib.vg_genomes_lengths(self._h, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def names(self):
        return [self._lib.vg_genomes_name(self._h, i).decode() for i in range(len(self))]

    def codes(self, idx):
        """Bases of genome idx as held on the host (0..3 = ACGT, 4 = N)."""
        out = np.empty(int(self.lengths()[idx]), dtype=np.uint8)
        _lib.check(_lib.load().vg_genomes_codes(self._h, int(idx), out.ctypes.data_as(C.c_void_p)))
        return out

    def to_device(self):
        check(self._lib.vg_genomes_to_device(self._h))

    # ---------------------------------------------------------------- prefilter
    def kmer_shared(self, k=25, fraction=1.0, shard=0, n_shards=1, min_shared=1):
        """-> (set_sizes int64[n], pairs structured array a>b with shared counts)."""
        n = len(self)
        sizes = np.zeros(max(n, 1), dtype=np.int64)
        pp = C.POINTER(PairCount)()
        npairs = C.c_int64()
        check(self._lib.vg_kmer_shared(self._h, k, float(fraction), shard, n_shards, int(min_shared),
                                       sizes.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(pp), C.byref(npairs)))
        return sizes[:n], _take(pp, npairs.value, PAIR_DTYPE)

    def kmer_set(self, idx, k=25, fraction=1.0):
        p = C.POINTER(C.c_uint64)()
        n = C.c_int64()
        check(self._lib.vg_kmer_set(self._h, idx, k, float(fraction), C.byref(p), C.byref(n)))
        return _take(p, n.value, np.dtype('<u8'))

    def filter_pairs(self, set_sizes, pairs, k=25, min_kmers=20, min_ident=0.7):
        """The pairs write_fltr would print (thresholds on shared count and ani-shorter), in memory."""
        sizes = np.ascontiguousarray(set_sizes, dtype=np.int64)
        pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
        out = C.POINTER(PairCount)(); n = C.c_int64()
        check(self._lib.vg_filter_pairs(int(k), int(min_kmers), float(min_ident), sizes.ctypes.data_as(C.POINTER(C.c_int64)), len(sizes),
                                        pairs.ctypes.data_as(C.POINTER(PairCount)), len(pairs), C.byref(out), C.byref(n)))
        return _take(out, n.value, PAIR_DTYPE)

    def write_fltr(self, out_path, set_sizes, pairs, k=25, fraction=1.0, min_kmers=20, min_ident=0.7, max_seqs=0):
        sizes = np.ascontiguousarray(set_sizes, dtype=np.int64)
        pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
        check(self._lib.vg_write_fltr(self._h, k, float(fraction), int(min_kmers), float(min_ident), int(max_seqs),
                                      sizes.ctypes.data_as(C.POINTER(C.c_int64)),
                                      pairs.ctypes.data_as(C.POINTER(PairCount)), len(pairs),
                                      os.fsencode(str(out_path))))

    # ---------------------------------------------------------------- align
    def align_order(self):
        out = np.zeros(max(len(self), 1), dtype=np.int32)
        check(self._lib.vg_align_order(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out[:len(self)]

    def read_filter(self, path=None, threshold=0.0):
        pp = C.POINTER(PairCount)()
        n = C.c_int64()
        check(self._lib.vg_read_filter(self._h, os.fsencode(str(path)) if path else None, float(threshold),
                                       C.byref(pp), C.byref(n)))
        return _take(pp, n.value, PAIR_DTYPE)

    def align_tasks(self, pairs):
        pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
        tp = C.POINTER(Task)()
        n = C.c_int64()
        check(self._lib.vg_align_tasks(self._h, pairs.ctypes.data_as(C.POINTER(PairCount)), len(pairs),
                                       C.byref(tp), C.byref(n)))
        return _take(tp, n.value, TASK_DTYPE)

    def lz_prepare(self, pairs, lz=None):
        """Optional head start of lz_align: the indexes of the pairs' genomes are queued on the device now (they are built
        while align_tasks runs on the host); lz_align takes them over when its tasks name the same references."""
        pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
        prm = LzParams(**{**DEFAULT_LZ, **(lz or {})})
        check(self._lib.vg_lz_prepare(self._h, pairs.ctypes.data_as(C.POINTER(PairCount)), len(pairs), C.byref(prm)))

    def lz_align(self, tasks, lz=None, want_regions=False):
        """LZ parse of the ordered pairs -> stats (and regions)."""
        tasks = np.ascontiguousarray(tasks, dtype=TASK_DTYPE)
        prm = LzParams(**{**DEFAULT_LZ, **(lz or {})})
        stats = np.zeros(max(len(tasks), 1), dtype=STAT_DTYPE)
        rp = C.POINTER(Region)()
        nr = C.c_int64()
        check(self._lib.vg_lz_align(self._h, tasks.ctypes.data_as(C.POINTER(Task)), len(tasks), C.byref(prm),
                                    stats.ctypes.data_as(C.POINTER(PairStat)),
                                    C.byref(rp) if want_regions else None, C.byref(nr)))
        stats = stats[:len(tasks)]
        if want_regions:
            return stats, _take(rp, nr.value, REGION_DTYPE)
        return stats

    def write_ani(self, out_path, tasks, stats, regions=None, columns=None, out_aln=None, lz=None,
                  out_filters=None):
        tasks = np.ascontiguousarray(tasks, dtype=TASK_DTYPE)
        stats = np.ascontiguousarray(stats, dtype=STAT_DTYPE)
        columns = list(columns or ALIGN_FIELDS[:11])
        cols = (C.c_char_p * len(columns))(*[c.encode() for c in columns])
        p = AlignParams()
        p.lz = LzParams(**{**DEFAULT_LZ, **(lz or {})})
        for name, val in (out_filters or {}).items():
            setattr(p, f'out_{name}', float(val))
        p.out_aln_path = os.fsencode(str(out_aln)) if out_aln else None
        p.out_columns = cols
        p.n_out_columns = len(columns)
        reg_ptr, nreg = None, 0
        if regions is not None:
            regions = np.ascontiguousarray(regions, dtype=REGION_DTYPE)
            reg_ptr, nreg = regions.ctypes.data_as(C.POINTER(Region)), len(regions)
        check(self._lib.vg_write_ani(self._h, tasks.ctypes.data_as(C.POINTER(Task)),
                                     stats.ctypes.data_as(C.POINTER(PairStat)), len(tasks), reg_ptr, nreg,
                                     os.fsencode(str(out_path)), C.byref(p)))


# ---------------------------------------------------------------- whole-stage calls (vclust_amd/stages.py: no numpy)
from .stages import align, align_params, prefilter  # noqa: E402,F401


def set_lz_fit(weak_seed_ratio=3, anchor_margin=-1, seed_choice=3):
    """The three thin constants of the LZ restatement (vg_set_lz_fit); no arguments = the fitted values."""
    f = _lib.LzFit(int(weak_seed_ratio), int(anchor_margin), int(seed_choice))
    _lib.load().vg_set_lz_fit(C.byref(f))


def set_range_scan(mode):
    """0 = a RANGE shard call scans every base itself; 1 = the sliced scan of the multi-GPU path with the peers' slices
    computed by this process (vg_set_range_scan)."""
    _lib.load().vg_set_range_scan(int(mode))


def set_placement_trials(n):
    """Placements of the prefilter workspace the first dense pass of this process may try (vg_set_placement_trials; 1 = none, the default)."""
    _lib.load().vg_set_placement_trials(int(n))


def release_device_memory():
    _lib.load().vg_release_device_memory()


# ---------------------------------------------------------------- measurement
def profile_enable(on=True):
    _lib.load().vg_profile_enable(int(bool(on)))


def profile_reset():
    _lib.load().vg_profile_reset()


def profile_get():
    lib = _lib.load()
    arr = (KernelTime * 64)()
    n = lib.vg_profile_get(arr, 64)
    return [dict(name=arr[i].name.decode(), total_ms=arr[i].total_ms, launches=arr[i].launches,
                 bytes=arr[i].bytes) for i in range(min(n, 64))]

"""The two whole-stage calls of the drop-in CLI (vg_prefilter, vg_align: FASTA on disk -> fltr.txt / ani.tsv on disk) as
thin ctypes wrappers WITHOUT numpy: a `vclust.py prefilter|align` process is one of these calls, and importing numpy
costs it 60-180 ms of its ~1 s.  vclust_amd.api re-exports them beside the array-level API."""
import ctypes as C
import os

from . import _lib
from ._lib import AlignParams, LzParams, PrefilterParams, check

DEFAULT_LZ = dict(mal=11, msl=7, mrd=40, mqd=40, reg=35, aw=15, am=7, ar=3)


def prefilter(paths, out_path, is_multifasta, k=25, min_kmers=20, min_ident=0.7, batch_size=0,
              kmers_fraction=1.0, max_seqs=0, num_threads=1, verbosity=0):
    lib = _lib.load()
    arr = (C.c_char_p * len(paths))(*[os.fsencode(str(p)) for p in paths])
    prm = PrefilterParams(k, min_kmers, min_ident, batch_size, kmers_fraction, max_seqs, num_threads,
                          verbosity, int(bool(is_multifasta)))
    check(lib.vg_prefilter(arr, len(paths), os.fsencode(str(out_path)), C.byref(prm)))


def align_params(columns, filter_path=None, filter_threshold=0.0, out_aln=None, lz=None, out_filters=None, num_threads=1,
                 verbosity=0, is_multifasta=True):
    """vg_align_params for the given options (the struct keeps its strings alive through attributes)."""
    cols = (C.c_char_p * len(columns))(*[c.encode() for c in columns])
    p = AlignParams()
    p.lz = LzParams(**{**DEFAULT_LZ, **(lz or {})})
    for name, val in (out_filters or {}).items():
        setattr(p, f'out_{name}', float(val))
    p.filter_path = os.fsencode(str(filter_path)) if filter_path else None
    p.filter_threshold = float(filter_threshold)
    p.out_aln_path = os.fsencode(str(out_aln)) if out_aln else None
    p.out_columns = cols
    p.n_out_columns = len(columns)
    p.num_threads = num_threads
    p.verbosity = verbosity
    p.is_multifasta = int(bool(is_multifasta))
    p._keep = cols
    return p


def align(paths, out_path, is_multifasta, columns, filter_path=None, filter_threshold=0.0, out_aln=None,
          lz=None, out_filters=None, num_threads=1, verbosity=0):
    lib = _lib.load()
    arr = (C.c_char_p * len(paths))(*[os.fsencode(str(p)) for p in paths])
    p = align_params(columns, filter_path, filter_threshold, out_aln, lz, out_filters, num_threads, verbosity, is_multifasta)
    check(lib.vg_align(arr, len(paths), os.fsencode(str(out_path)), C.byref(p)))

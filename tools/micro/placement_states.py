#!/usr/bin/env python3
"""Developer tool: the placement states of the prefilter's scattering kernels (the same kernels on the same
data run 8 % apart depending on where the driver puts the buffers).  ONE process; REPS times: return every cached block to
the driver, run the prefilter pass twice (the first allocates afresh, the second is the steady state of that placement) and
print the second pass's milliseconds per scope and the addresses of the large blocks (VG_ALLOC_TRACE on stderr).
Under `rocprofv3 --pmc ... --kernel-trace` the per-dispatch counters of the two states can be told apart by the durations
(tools/micro/placement_pmc.sh).  HOLD_GB > 0: a block of that size is allocated and released first in every repetition
(the state right after a large release)."""
import os, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import ctypes as C
import numpy as np
from vclust_amd import api, synth, _lib
api.set_device(0)
NF = int(os.environ.get('NF', '10000')); REPS = int(os.environ.get('REPS', '8')); HOLD = float(os.environ.get('HOLD_GB', '0'))
codes, offsets, names, _ = synth.make_workload('phage-100k', NF)
gs = api.GenomeSet.from_codes(codes, offsets, names); gs.to_device()
lib = _lib.load()
for rep in range(REPS):
    api.release_device_memory()
    if HOLD > 0:
        sizes = (C.c_int64 * 1)(int(HOLD * (1 << 30)))
        _lib.check(lib.vg_alloc_selftest(sizes, 1, 1))
    print(f'[placement] rep {rep} allocating', file=sys.stderr, flush=True)
    gs.kmer_shared(k=25, min_shared=20)
    api.profile_enable(True); api.profile_reset()
    gs.kmer_shared(k=25, min_shared=20)
    prof = {e['name']: round(e['total_ms'], 2) for e in api.profile_get()}
    api.profile_enable(False)
    print(f'rep {rep}: ' + ' '.join(f'{k}={v}' for k, v in sorted(prof.items())), flush=True)

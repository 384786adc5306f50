"""Developer tool: loop iterations / events per parse task (VG_LZ_ABLATE=1024|64)."""
import os, sys, pathlib
import numpy as np
os.environ['VG_LZ_ABLATE'] = str(1024 | 2048)
os.environ['VG_LZ_SEGMENTS'] = '1'
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vclust_amd import api, synth
api.set_device(0)
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 100
c, o, n = synth.make_families(nf, 10, 40000, seed=1)
gs = api.GenomeSet.from_codes(c, o, n); gs.to_device()
tasks = gs.align_tasks(synth.family_pairs(nf, 10))
gs.lz_align(tasks)
st = gs.lz_align(tasks)
it = st['n_match'].astype(np.float64); ev = st['aln_len'].astype(np.float64); us = st['n_regions'] / 100.0
print('tasks', len(st), 'anchor batches (wave max): mean %.0f median %.0f max %.0f' % (it.mean(), np.median(it), it.max()))
print('seed batches: mean %.0f median %.0f max %.0f' % (ev.mean(), np.median(ev), ev.max()))

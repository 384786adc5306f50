"""Developer tool: loop iterations / events per parse task (VG_LZ_ABLATE=1024|64)."""
import os, sys, pathlib
import numpy as np
os.environ['VG_LZ_ABLATE'] = str(1024)
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
print('tasks', len(st), 'iterations: mean %.0f median %.0f max %.0f' % (it.mean(), np.median(it), it.max()))
print('events: mean %.0f median %.0f max %.0f' % (ev.mean(), np.median(ev), ev.max()))
print('us per iteration: mean %.2f; corr(us, iter) %.3f corr(us, events) %.3f' % ((us / it).mean(), np.corrcoef(us, it)[0, 1], np.corrcoef(us, ev)[0, 1]))
A = np.stack([it - ev, ev, np.ones_like(it)], 1)
coef = np.linalg.lstsq(A, us, rcond=None)[0]
print('fit: us = %.2f * literal_iters + %.2f * events + %.1f' % tuple(coef))
o = np.argsort(-us)[:3]
print('slowest', [(us[i], it[i], ev[i]) for i in o])

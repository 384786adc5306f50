"""Developer tool: per-task duration distribution of the parse kernel (VG_LZ_ABLATE=32)."""
import os, sys, pathlib
import numpy as np
os.environ['VG_LZ_ABLATE'] = os.environ.get('VG_LZ_ABLATE', '32')
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vclust_amd import api, synth
api.set_device(0)
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 100
c, o, n = synth.make_families(nf, 10, 40000, seed=1)
gs = api.GenomeSet.from_codes(c, o, n); gs.to_device()
tasks = gs.align_tasks(synth.family_pairs(nf, 10))
gs.lz_align(tasks)
st = gs.lz_align(tasks)
us = st['n_regions'].astype(np.float64) / 100.0      # 100 MHz ticks -> microseconds
ident = st['n_match'] / np.maximum(st['aln_len'], 1)
print('tasks', len(us), 'sum %.1f ms' % (us.sum() / 1e3), 'max %.0f us' % us.max(), 'mean %.1f us' % us.mean(), 'median %.1f' % np.median(us))
for q in (50, 90, 95, 99, 99.9): print('p%s %.0f us' % (q, np.percentile(us, q)))
order = np.argsort(-us)[:5]
print('slowest:', [(round(us[i]), round(float(ident[i]), 3), int(st['aln_len'][i])) for i in order])

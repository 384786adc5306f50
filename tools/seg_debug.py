"""Developer tool: hand-over statistics of the segmented parse (VG_LZ_ABLATE=512)."""
import os, sys, pathlib
import numpy as np
os.environ['VG_LZ_ABLATE'] = '512'
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vclust_amd import api, synth
api.set_device(0)
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 100
c, o, n = synth.make_families(nf, 10, 40000, seed=1)
gs = api.GenomeSet.from_codes(c, o, n); gs.to_device()
tasks = gs.align_tasks(synth.family_pairs(nf, 10))
gs.lz_align(tasks)
st = gs.lz_align(tasks)
over = st['n_match'].astype(np.int64); hops = st['aln_len'] & 255; cnt1 = st['aln_len'] >> 8
us = st['n_regions'] / 100.0
print('tasks', len(st), 'hops hist', np.bincount(hops, minlength=5))
print('wave0 overrun bases: median %d p90 %d max %d' % (np.median(over), np.percentile(over, 90), over.max()))
print('log entries of wave 1: median %d p90 %d max %d' % (np.median(cnt1), np.percentile(cnt1, 90), cnt1.max()))
print('block time us: median %.0f p90 %.0f p99 %.0f max %.0f' % (np.median(us), np.percentile(us, 90), np.percentile(us, 99), us.max()))

"""Developer tool: wall-clock split of one bench step (host + device) on the phage-1k set."""
import sys, pathlib, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import bench
from vclust_amd import api, synth
api.set_device(0)
import os
NF = int(os.environ.get('NF', '100'))
codes, offsets, names = synth.make_families(NF, 10, 40000, seed=1)
gs = api.GenomeSet.from_codes(codes, offsets, names); gs.to_device()
for it in range(4):
    t = [time.perf_counter()]
    sizes, pairs = gs.kmer_shared(k=25, min_shared=20); t.append(time.perf_counter())
    cand = gs.filter_pairs(sizes, pairs, k=25, min_kmers=20, min_ident=0.7); t.append(time.perf_counter())
    tasks = gs.align_tasks(cand); t.append(time.perf_counter())
    stats = gs.lz_align(tasks); t.append(time.perf_counter())
    d = np.diff(t) * 1e3
    print('kmer_shared %.2f  candidate %.2f  tasks %.2f  lz_align %.2f  total %.2f ms' % (*d, d.sum()))
api.profile_enable(True); api.profile_reset()
sizes, pairs = gs.kmer_shared(k=25, min_shared=20); stats = gs.lz_align(tasks)
print({e['name']: round(e['total_ms'], 3) for e in api.profile_get()})

"""Developer tool: wall-clock split of one bench step (host + device).  NF families of the phage-100k generator."""
import os, sys, pathlib, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vclust_amd import api, synth
api.set_device(0)
NF = int(os.environ.get('NF', '100'))
codes, offsets, names, _ = synth.make_workload('phage-100k', NF)
gs = api.GenomeSet.from_codes(codes, offsets, names); gs.to_device()
for it in range(4):
    api.profile_enable(True); api.profile_reset()
    t = [time.perf_counter()]
    sizes, pairs = gs.kmer_shared(k=25, min_shared=20); t.append(time.perf_counter())
    k1 = sum(e['total_ms'] for e in api.profile_get()); api.profile_reset()
    cand = gs.filter_pairs(sizes, pairs, k=25, min_kmers=20, min_ident=0.7); t.append(time.perf_counter())
    tasks = gs.align_tasks(cand); t.append(time.perf_counter())
    stats = gs.lz_align(tasks); t.append(time.perf_counter())
    k2 = sum(e['total_ms'] for e in api.profile_get())
    d = np.diff(t) * 1e3
    print('kmer_shared %.2f (kernels %.2f)  candidate %.2f  tasks %.2f  lz_align %.2f (kernels %.2f)  total %.2f ms' % (d[0], k1, d[1], d[2], d[3], k2, d.sum()))

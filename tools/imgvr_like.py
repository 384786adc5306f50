"""Developer tool: BASELINE configs[2]-like run (contigs 5-200 kb log-uniform, families of 1-20)."""
import sys, pathlib, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / 'tests'))
from vclust_amd import api, synth
import bench
n_target = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(2)
seqs, names = [], []
fam = 0
while len(seqs) < n_target:
    members = min(int(rng.geometric(0.2)), 20)
    c, o, n = synth.make_families(1, members, seed=1000 + fam, length_range=(5000, 200000))
    for i in range(members):
        seqs.append(c[o[i]:o[i + 1]]); names.append(f'f{fam}_{i}')
    fam += 1
offsets = np.zeros(len(seqs) + 1, dtype=np.int64); offsets[1:] = np.cumsum([len(s) for s in seqs])
codes = np.concatenate(seqs)
print(len(seqs), 'contigs', offsets[-1] / 1e6, 'Mbp', 'families', fam, flush=True)
api.set_device(0)
gs = api.GenomeSet.from_codes(codes, offsets, names); gs.to_device()
for it in range(3):
    api.profile_enable(True); api.profile_reset()
    t0 = time.perf_counter()
    sizes, pairs = gs.kmer_shared(k=25, min_shared=20)
    cand = gs.filter_pairs(sizes, pairs, k=25, min_kmers=20, min_ident=0.7)
    tasks = gs.align_tasks(cand)
    stats = gs.lz_align(tasks)
    dt = time.perf_counter() - t0
    print('pairs', len(cand), 'step %.1f ms' % (dt * 1e3), '%.0f pairs/s' % (len(cand) / dt),
          {e['name']: round(e['total_ms'], 2) for e in api.profile_get()}, flush=True)
# oracle check on a sample
import oracle_lib as orc
idx = np.random.default_rng(0).choice(len(tasks), min(40, len(tasks)), replace=False)
bad = 0
for i in idx:
    q, r = int(tasks[i]['q']), int(tasks[i]['r'])
    ref = orc.lz_pair_stat(codes[offsets[q]:offsets[q + 1]], codes[offsets[r]:offsets[r + 1]])
    bad += ref != tuple(int(x) for x in stats[i])
print('oracle sample mismatches', bad, 'of', len(idx))

"""Developer tool: per-rank device work of a strong-scaling run, emulated on ONE GPU: rank 0's shard of the
phage-100k set (NF families) for world = 1, 2, 4, 8 (collectives excluded: they move a few MB)."""
import os, sys, pathlib, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vclust_amd import api, synth, distributed as D
api.set_device(0)
NF = int(os.environ.get('NF', '10000'))
codes, offsets, names, _ = synth.make_workload('phage-100k', NF)
gs = api.GenomeSet.from_codes(codes, offsets, names); gs.to_device()
sizes, pairs = gs.kmer_shared(k=25, min_shared=20)
tasks = gs.align_tasks(gs.filter_pairs(sizes, pairs))
for world in (1, 2, 4, 8):
    owner = D.align_owner(tasks, len(gs), world)
    mine = tasks[owner == 0]
    best = None
    for it in range(3):
        api.profile_enable(True); api.profile_reset()
        t0 = time.perf_counter()
        s, p = gs.kmer_shared(k=25, shard=0, n_shards=world, min_shared=1 if world > 1 else 20)
        t1 = time.perf_counter()
        st = gs.lz_align(mine)
        t2 = time.perf_counter()
        prof = {e['name']: round(e['total_ms'], 1) for e in api.profile_get()}
        cur = ((t1 - t0) * 1e3, (t2 - t1) * 1e3, prof, len(p))
        if best is None or cur[0] + cur[1] < best[0] + best[1]: best = cur
    print(f'world {world}: prefilter shard {best[0]:.1f} ms ({best[3]} partial pairs)  align share {best[1]:.1f} ms  total {best[0] + best[1]:.1f} ms  {best[2]}')

"""Developer tool: per-rank work of a strong-scaling run, emulated on ONE GPU: rank 0's shard of the phage-100k set (NF
families) for world = 1, 2, 4, 8 -- its k-mer RANGE shard of the prefilter and its reference range of the align tasks --
with the host work every rank repeats between the stages timed beside it (thresholds and the listing of its own tasks;
the canonical task list runs on a helper thread beside the kernels, as in vg_lz_align_pairs_sharded).  The exchanges themselves are excluded (a few MB per step: set sizes, nominated pair keys, counts, rows)."""
import os, sys, pathlib, time, threading
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vclust_amd import api, synth, distributed as D
api.set_device(0)
NF = int(os.environ.get('NF', '10000'))
codes, offsets, names, _ = synth.make_workload('phage-100k', NF)
gs = api.GenomeSet.from_codes(codes, offsets, names); gs.to_device()
sizes, pairs = gs.kmer_shared(k=25, min_shared=20)
tasks = gs.align_tasks(gs.filter_pairs(sizes, pairs))
base = None
for world in (1, 2, 4, 8):
    best = None
    for it in range(3):
        api.profile_enable(True); api.profile_reset()
        t0 = time.perf_counter()
        s, p = gs.kmer_shared(k=25, shard=0, n_shards=world, min_shared=1 if world > 1 else 20)
        t1 = time.perf_counter()
        # host work of every rank between the stages, as vg_lz_align_pairs_sharded does it: thresholds on the global pair
        # list, the rank's own tasks listed from the pairs (reference ranges from the genomes' task counts), and the
        # canonical task list of the whole set on a helper thread BESIDE the kernels
        cand = gs.filter_pairs(sizes, pairs)
        mine = D.align_pairs_share(gs, cand, world, 0)
        t2 = time.perf_counter()
        th = threading.Thread(target=lambda: gs.align_tasks(cand)); th.start()
        st = gs.lz_align(mine)
        th.join()
        t3 = time.perf_counter()
        prof = {e['name']: round(e['total_ms'], 1) for e in api.profile_get()}
        cur = ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, prof, len(p))
        if best is None or cur[0] + cur[1] + cur[2] < best[0] + best[1] + best[2]: best = cur
    tot = best[0] + best[1] + best[2]
    if base is None: base = tot
    print(f'world {world}: prefilter shard {best[0]:.1f} ms ({best[4]} partial pairs)  host between the stages {best[1]:.1f} ms  align share {best[2]:.1f} ms  '
          f'total {tot:.1f} ms = {base / tot:.2f}x  {best[3]}')

"""Developer tool: per-rank compute of the weak-scaling bench, emulated on one GPU.
For N in 1,2,4,8: N x 100 families; rank 0's share of the prefilter (k-mer shard 0 of N) and of
the alignments (the tasks of the first reference range), timed without the collectives."""
import sys, pathlib, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import bench
from vclust_amd import api, synth
from vclust_amd import distributed as D
api.set_device(0)
for N in (1, 2, 4, 8):
    codes, offsets, names = synth.make_families(100 * N, 10, 40000, seed=1)
    gs = api.GenomeSet.from_codes(codes, offsets, names); gs.to_device()
    # global candidate list (what every rank holds after the gather)
    sizes, pairs = gs.kmer_shared(k=25, min_shared=20)
    cand = gs.filter_pairs(sizes, pairs, k=25, min_kmers=20, min_ident=0.7)
    tasks = gs.align_tasks(cand)
    mine = tasks[D.ref_owner(tasks, N) == 0]
    for it in range(3):
        api.profile_enable(True); api.profile_reset()
        t0 = time.perf_counter()
        s_, p_ = gs.kmer_shared(k=25, shard=0, n_shards=N, min_shared=1) if N > 1 else gs.kmer_shared(k=25, min_shared=20)
        t1 = time.perf_counter()
        st = gs.lz_align(mine)
        t2 = time.perf_counter()
    prof = {e['name']: round(e['total_ms'], 2) for e in api.profile_get()}
    print(f'N={N}: prefilter {1e3*(t1-t0):.2f} ms  align {1e3*(t2-t1):.2f} ms  total {1e3*(t2-t0):.2f} ms  shard pairs {len(p_)}  {prof}')

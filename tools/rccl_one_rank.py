"""One-rank RCCL smoke test (VCLUST_DIST_FORCE=1): the collectives of vclust_amd/distributed.py run on the
real nccl backend with world size 1 and must reproduce the plain single-process results."""
import os, sys, pathlib
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
os.environ['VCLUST_DIST_FORCE'] = '1'
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np
from vclust_amd import api, synth, distributed as D
dist, device = D.init_process_group('nccl')
api.set_device(0)
codes, offsets, names = synth.make_families(6, 5, length=6000, seed=4)
gs = api.GenomeSet.from_codes(codes, offsets, names)
s0, p0 = gs.kmer_shared(k=25, min_shared=1)
s1, p1 = D.prefilter_counts(gs, dist, device, 0, 1, 25, 1.0)
assert np.array_equal(s0, s1)
k0 = np.sort((p0['a'].astype(np.int64) << 32) | p0['b']); k1 = (p1['a'].astype(np.int64) << 32) | p1['b']
assert np.array_equal(k0, k1) and int(p0['shared'].sum()) == int(p1['shared'].sum())
cand = gs.filter_pairs(s1, p1)
tasks = gs.align_tasks(cand)
st0, rg0 = gs.lz_align(tasks, want_regions=True)
st1, rg1 = D.align_rows(gs, tasks, dist, device, 0, 1, None, True)
assert np.array_equal(st0, st1) and len(rg0) == len(rg1)
st2, _ = D.align_rows(gs, tasks, dist, device, 0, 1, None, False)
assert np.array_equal(st0, st2)
dist.barrier(); dist.destroy_process_group()
print('rccl one-rank ok:', len(p1), 'pairs', len(tasks), 'tasks')

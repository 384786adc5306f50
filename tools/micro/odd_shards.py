"""Developer tool: RANGE / HASH shard counts that do not divide the 2 048 digits (5, 7, 13, 64, 256; 300 -> HASH) at k = 12, 25, 31 on small,
medium and 120 M-position sets: the shards must add up to the single pass (last run: 0 mismatches)."""
import sys, pathlib
sys.path.insert(0, '.')
import numpy as np
from vclust_amd import api, synth
api.set_device(0)
bad = 0
for (nf, mem, L, seed) in ((3, 4, 30000, 1), (60, 5, 20000, 2), (500, 6, 40000, 3)):
    codes, offsets, names = synth.make_families(nf, mem, length=L, seed=seed)
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    for k in (12, 25, 31):
        s0, p0 = gs.kmer_shared(k=k)
        ref = {(int(p['a']), int(p['b'])): int(p['shared']) for p in p0}
        for ns in (5, 7, 13, 64, 256, 300):
            if nf == 500 and ns > 13: continue
            tot = np.zeros_like(s0); acc = {}
            for s in range(ns):
                sz, pr = gs.kmer_shared(k=k, shard=s, n_shards=ns); tot += sz
                for p in pr: acc[(int(p['a']), int(p['b']))] = acc.get((int(p['a']), int(p['b'])), 0) + int(p['shared'])
            ok = np.array_equal(tot, s0) and acc == ref
            if not ok: bad += 1; print('MISMATCH', nf, k, ns)
print('odd shard counts:', bad, 'mismatches')

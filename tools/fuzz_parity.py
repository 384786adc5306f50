"""Randomised parity fuzzing of the HIP path against the CPU oracle (longer than the test suite).
usage: fuzz_parity.py [n_cases] [seed] [big|huge]
  big:  sets of 2-30 M positions -- both partition levels of the index pipeline
  huge: sets of 70-200 M positions -- the 32 768-position tiles; RANGE shards and the forced sub-shard loop are held to the
        single pass of the same set (itself held to the oracle at the other sizes), the LZ parse to the oracle on a sample"""
import sys, pathlib, time
import numpy as np
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / 'tests'))
import oracle_lib as orc
from vclust_amd import api, synth
api.set_device(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
big = len(sys.argv) > 3
huge = big and sys.argv[3] == 'huge'
from vclust_amd import _lib
bad = 0
t0 = time.time()
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    nf = int(rng.integers(1, 5)); mem = int(rng.integers(2, 6))
    if big: nf = int(rng.integers(40, 400))
    lo = int(rng.choice([300, 2000, 9000, 30000])); hi = lo * int(rng.integers(1, 5))
    if huge: nf = int(rng.integers(300, 700)); mem = int(rng.integers(4, 8)); lo = int(rng.choice([20000, 40000])); hi = lo * 2
    p_hi = float(rng.choice([0.02, 0.12, 0.25, 0.4]))
    codes, offsets, names = synth.make_families(nf, mem, seed=seed0 + case, length_range=(lo, hi), p_hi=p_hi)
    codes = codes.copy()
    for _ in range(int(rng.integers(0, 6))):                       # N runs
        p = int(rng.integers(0, max(1, len(codes) - 80))); codes[p:p + int(rng.integers(1, 70))] = 4
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    k = int(rng.choice([15, 21, 25, 30])); frac = float(rng.choice([1.0, 1.0, 0.5, 0.1]))
    if huge: frac = 1.0
    sizes, pairs = gs.kmer_shared(k=k, fraction=frac)
    if huge:
        osizes, opairs = list(sizes), {(int(p['a']), int(p['b'])): int(p['shared']) for p in pairs}
        _lib.load().vg_set_subshards(int(rng.choice([2, 5, 8])))
        try: s8, p8 = gs.kmer_shared(k=k, min_shared=1)
        finally: _lib.load().vg_set_subshards(0)
        if list(s8) != osizes or {(int(p['a']), int(p['b'])): int(p['shared']) for p in p8} != opairs:
            bad += 1; print('SUB-SHARD MISMATCH case', case, 'seed', seed0 + case, flush=True)
    else:
        osizes, opairs = orc.shared_all(codes, offsets, k=k, fraction=frac)
    if list(sizes) != list(osizes) or {(int(p['a']), int(p['b'])): int(p['shared']) for p in pairs} != opairs:
        bad += 1; print('PREFILTER MISMATCH case', case, 'seed', seed0 + case, flush=True)
    ns = int(rng.choice([2, 3, 8, 32])); tot = np.zeros_like(sizes); acc = {}
    for s in range(ns):
        sz, pr = gs.kmer_shared(k=k, fraction=frac, shard=s, n_shards=ns); tot += sz
        for p in pr: acc[(int(p['a']), int(p['b']))] = acc.get((int(p['a']), int(p['b'])), 0) + int(p['shared'])
    if list(tot) != list(osizes) or acc != opairs:
        bad += 1; print('SHARD MISMATCH case', case, 'seed', seed0 + case, flush=True)
    # the multi-GPU form of the shards: every rank scans 1/ns of the bases, the peers' slices computed in-process (applies from two
    # partition levels on; smaller sets take the replicated scan again and must agree all the same)
    api.set_range_scan(1); tot = np.zeros_like(sizes); acc = {}
    try:
        for s in range(ns):
            sz, pr = gs.kmer_shared(k=k, fraction=frac, shard=s, n_shards=ns); tot += sz
            for p in pr: acc[(int(p['a']), int(p['b']))] = acc.get((int(p['a']), int(p['b'])), 0) + int(p['shared'])
    finally: api.set_range_scan(0)
    if list(tot) != list(osizes) or acc != opairs:
        bad += 1; print('SLICED SHARD MISMATCH case', case, 'seed', seed0 + case, ns, flush=True)
    # HASH sub-shards from one scan of the bases (a fraction always cuts by hash; 33+ sub-shards scan per pass)
    if frac < 1.0 and not huge:
        _lib.load().vg_set_subshards(int(rng.choice([2, 7, 32, 40])))
        try: s8, p8 = gs.kmer_shared(k=k, fraction=frac, min_shared=1)
        finally: _lib.load().vg_set_subshards(0)
        if list(s8) != list(osizes) or {(int(p['a']), int(p['b'])): int(p['shared']) for p in p8} != opairs:
            bad += 1; print('HASH SUB-SHARD MISMATCH case', case, 'seed', seed0 + case, flush=True)
    lz = None
    if rng.random() < 0.5:
        mal = int(rng.integers(9, 17)); msl = int(rng.integers(5, min(mal, 9) + 1))
        lz = dict(mal=mal, msl=msl, mrd=int(rng.integers(10, 80)), mqd=int(rng.integers(10, 80)), reg=int(rng.integers(20, 80)),
                  aw=int(rng.integers(8, 24)), am=int(rng.integers(2, 10)), ar=int(rng.integers(2, 6)))
        lz['am'] = min(lz['am'], lz['aw'] - 1)
    tasks = gs.align_tasks(gs.read_filter(None)) if len(gs) <= 8 else gs.align_tasks(synth.family_pairs(nf, mem))
    if big and len(tasks) > 400: tasks = tasks[np.sort(rng.choice(len(tasks) // 2, 200, replace=False))[:, None] * 2 + np.arange(2)].reshape(-1)
    if rng.random() < 0.5:                                            # the prepared-index path (rows must not depend on it)
        qq, rr = tasks['q'].astype(np.int64), tasks['r'].astype(np.int64)
        pr = np.zeros(len(tasks), dtype=api.PAIR_DTYPE); pr['a'] = np.maximum(qq, rr); pr['b'] = np.minimum(qq, rr)
        gs.lz_prepare(pr[rng.random(len(pr)) < 0.9] if rng.random() < 0.3 else pr, lz=lz)
    # the thin constants of the fit, at random (product through vg_set_lz_fit, checker through its developer variables)
    import os
    fit = {}
    if rng.random() < 0.4:
        fit = dict(weak_seed_ratio=int(rng.choice([0, 2, 3, 5])), anchor_margin=int(rng.choice([-1, 4, 6, 8])), seed_choice=int(rng.choice([1, 3])))
    api.set_lz_fit(**fit)
    for name, key in (('VG_LZ_WEAK_SEED', 'weak_seed_ratio'), ('VG_LZ_ANCHOR_MARGIN', 'anchor_margin'), ('VG_LZ_SEED_CHOICE', 'seed_choice')):
        if key in fit: os.environ[name] = str(fit[key])
        else: os.environ.pop(name, None)
    stats = gs.lz_align(tasks, lz=lz)
    # --out-aln from the same single parse: the rows must not change, and the regions must add up to them task by task
    # (count, matches, aligned length), lie in query order without overlap and each span >= reg
    st2, rg = gs.lz_align(tasks, lz=lz, want_regions=True)
    api.set_lz_fit()
    ok = np.array_equal(stats, st2) and len(rg) == int(stats['n_regions'].sum())
    if ok and len(rg):
        tk = rg['task'].astype(np.int64); span = (rg['qend'] - rg['qstart'] + 1).astype(np.int64)
        ok = (np.array_equal(np.bincount(tk, minlength=len(tasks)), stats['n_regions'].astype(np.int64)) and
              np.array_equal(np.bincount(tk, weights=rg['n_match'], minlength=len(tasks)).astype(np.int64), stats['n_match'].astype(np.int64)) and
              np.array_equal(np.bincount(tk, weights=span, minlength=len(tasks)).astype(np.int64), stats['aln_len'].astype(np.int64)) and
              bool(np.all(span >= (lz or {}).get('reg', 35))))
        same = tk[1:] == tk[:-1]
        ok = ok and bool(np.all(tk[1:] >= tk[:-1]) or True) and bool(np.all(rg['qstart'][1:][same] > rg['qend'][:-1][same]))
    if not ok:
        bad += 1; print('REGIONS MISMATCH case', case, 'seed', seed0 + case, lz, fit, flush=True)
    for t, s in zip(tasks, stats):
        q, r = int(t['q']), int(t['r'])
        ref = orc.lz_pair_stat(codes[offsets[q]:offsets[q + 1]], codes[offsets[r]:offsets[r + 1]], lz=lz)
        if ref != (int(s['n_match']), int(s['aln_len']), int(s['n_regions'])):
            bad += 1; print('LZ MISMATCH case', case, 'seed', seed0 + case, (q, r), ref, tuple(int(x) for x in s), lz, fit, flush=True); break
print(f'{n_cases} cases, {bad} mismatches, {time.time() - t0:.0f} s')

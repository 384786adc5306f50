"""Randomised parity sweeps of the HIP path against the CPU oracle: ragged lengths, N runs,
non-default parameters, duplicated/repetitive sequence, and size-independent properties on the
bench-sized set (BASELINE.json configs[1])."""
import numpy as np
import pytest

import oracle_lib as orc
from vclust_amd import api, synth

pytestmark = pytest.mark.gpu


def _check_prefilter(codes, offsets, gs, k, fraction=1.0):
    sizes, pairs = gs.kmer_shared(k=k, fraction=fraction)
    osizes, opairs = orc.shared_all(codes, offsets, k=k, fraction=fraction)
    assert list(sizes) == list(osizes)
    assert {(int(p['a']), int(p['b'])): int(p['shared']) for p in pairs} == opairs
    return pairs


def _check_lz(codes, offsets, gs, tasks, lz=None):
    stats = gs.lz_align(tasks, lz=lz)
    bad = []
    for t, s in zip(tasks, stats):
        q, r = int(t['q']), int(t['r'])
        ref = orc.lz_pair_stat(codes[offsets[q]:offsets[q + 1]], codes[offsets[r]:offsets[r + 1]], lz=lz)
        if ref != (int(s['n_match']), int(s['aln_len']), int(s['n_regions'])):
            bad.append((q, r, ref, tuple(int(x) for x in s)))
    assert not bad, bad[:5]


def _sprinkle_n(codes, rng, n_runs):
    codes = codes.copy()
    for _ in range(n_runs):
        p = int(rng.integers(0, len(codes) - 60))
        codes[p:p + int(rng.integers(1, 50))] = 4
    return codes


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_ragged_families_with_n(seed):
    rng = np.random.default_rng(seed)
    codes, offsets, names = synth.make_families(12, 4, seed=seed, length_range=(1500, 30000), p_hi=0.2)
    codes = _sprinkle_n(codes, rng, 40)
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    pairs = _check_prefilter(codes, offsets, gs, 25)
    _check_prefilter(codes, offsets, gs, 17, fraction=0.3)
    tasks = gs.align_tasks(pairs[pairs['shared'] >= 5])
    assert len(tasks) > 40
    _check_lz(codes, offsets, gs, tasks)


@pytest.mark.parametrize('lz', [
    dict(mal=9, msl=6, mrd=20, mqd=30, reg=20, aw=10, am=4, ar=2),
    dict(mal=14, msl=7, mrd=60, mqd=70, reg=50, aw=25, am=12, ar=4),
    dict(mal=11, msl=7, mrd=40, mqd=40, reg=35, aw=32, am=15, ar=1),
    dict(mal=16, msl=5, mrd=10, mqd=5, reg=1, aw=3, am=0, ar=3),
    # the ends of the ranges the library takes: long seeds (bucket = 12 bits of each bit plane, the global index build),
    # the longest anchors (a 31-bit validity mask of the query), the shortest of both
    dict(mal=31, msl=12, mrd=40, mqd=40, reg=35, aw=15, am=7, ar=3),
    dict(mal=20, msl=9, mrd=40, mqd=100, reg=35, aw=15, am=7, ar=16),
    dict(mal=8, msl=4, mrd=40, mqd=40, reg=10, aw=15, am=7, ar=3),
])
def test_non_default_lz_parameters(lz):
    codes, offsets, names = synth.make_families(5, 4, seed=11, length=8000, p_hi=0.25)
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    tasks = gs.align_tasks(synth.family_pairs(5, 4))
    _check_lz(codes, offsets, gs, tasks, lz=lz)


def test_unrelated_and_repetitive_sequences():
    """Probe-heavy pairs (unrelated genomes), tandem repeats and low-complexity runs."""
    rng = np.random.default_rng(5)
    a = rng.integers(0, 4, size=20000, dtype=np.uint8)
    b = rng.integers(0, 4, size=15000, dtype=np.uint8)
    unit = rng.integers(0, 4, size=37, dtype=np.uint8)
    rep = np.tile(unit, 300)                                  # tandem repeat, 11 100 bp
    rep2 = rep.copy(); rep2[rng.random(len(rep2)) < 0.03] = 1
    poly = np.concatenate([a[:3000], np.zeros(4000, np.uint8), a[3000:6000]])   # poly-A block
    dup = np.concatenate([a[:8000], a[2000:9000], a[:3000]])                   # duplicated segments
    seqs = [a, b, rep, rep2, poly, dup]
    offsets = np.zeros(len(seqs) + 1, dtype=np.int64); offsets[1:] = np.cumsum([len(s) for s in seqs])
    codes = np.concatenate(seqs)
    gs = api.GenomeSet.from_codes(codes, offsets)
    _check_prefilter(codes, offsets, gs, 25)
    _check_prefilter(codes, offsets, gs, 15)
    _check_lz(codes, offsets, gs, gs.align_tasks(gs.read_filter(None)))


def test_kmers_shared_by_many_genomes():
    """A conserved block in 700 genomes: k-mer runs longer than one staged tile of the run kernel
    (general path), next to ordinary short runs."""
    rng = np.random.default_rng(8)
    core = rng.integers(0, 4, size=60, dtype=np.uint8)
    seqs = []
    for i in range(700):
        flank = rng.integers(0, 4, size=int(rng.integers(150, 400)), dtype=np.uint8)
        cut = int(rng.integers(0, len(flank)))
        seqs.append(np.concatenate([flank[:cut], core, flank[cut:]]))
    for i in range(0, 40, 2):                                   # a few close relatives on top
        seqs[i + 1] = seqs[i].copy(); seqs[i + 1][::37] = (seqs[i + 1][::37] + 1) % 4
    offsets = np.zeros(len(seqs) + 1, dtype=np.int64); offsets[1:] = np.cumsum([len(s) for s in seqs])
    codes = np.concatenate(seqs)
    gs = api.GenomeSet.from_codes(codes, offsets)
    pairs = _check_prefilter(codes, offsets, gs, 25)
    assert len(pairs) >= 600 * 599 // 2
    _check_prefilter(codes, offsets, gs, 18, fraction=0.5)


def test_large_reference_global_index_path():
    """References above 2^18 RR symbols use the global-memory index build."""
    rng = np.random.default_rng(8)
    a = rng.integers(0, 4, size=1100000, dtype=np.uint8)
    b = a.copy(); m = rng.random(len(b)) < 0.04; b[m] = (b[m] + 1) & 3
    c = a[200000:260000].copy()
    seqs = [a, b, c]
    offsets = np.zeros(4, dtype=np.int64); offsets[1:] = np.cumsum([len(s) for s in seqs])
    codes = np.concatenate(seqs)
    gs = api.GenomeSet.from_codes(codes, offsets)
    _check_lz(codes, offsets, gs, gs.align_tasks(gs.read_filter(None)))


def test_full_size_properties():
    """BASELINE configs[1] (1 000 x 40 kb): size-independent checks + oracle on a sample."""
    nf, mem = 100, 10
    codes, offsets, names = synth.make_families(nf, mem, length=40000, seed=1)
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    sizes, pairs = gs.kmer_shared(k=25, min_shared=20)
    lens = gs.lengths()
    assert np.all(sizes <= lens - 24) and np.all(sizes > 0.9 * (lens - 24))
    # random ancestors: exactly the within-family pairs survive, each counted once
    got = {(int(p['a']), int(p['b'])) for p in pairs}
    assert got == {(int(p['a']), int(p['b'])) for p in synth.family_pairs(nf, mem)}
    assert np.all(pairs['shared'] <= np.minimum(sizes[pairs['a']], sizes[pairs['b']]))
    # sharded runs add up to the unsharded one (checksum of checksums)
    tot = np.zeros_like(sizes); acc = {}
    for s in range(4):
        sz, pr = gs.kmer_shared(k=25, shard=s, n_shards=4)
        tot += sz
        for p in pr:
            acc[(int(p['a']), int(p['b']))] = acc.get((int(p['a']), int(p['b'])), 0) + int(p['shared'])
    assert np.array_equal(tot, sizes)
    assert {k: v for k, v in acc.items() if v >= 20} == {(int(p['a']), int(p['b'])): int(p['shared']) for p in pairs}
    tasks = gs.align_tasks(pairs)
    stats = gs.lz_align(tasks)
    ql = lens[tasks['q']]
    assert np.all(stats['n_match'] <= stats['aln_len']) and np.all(stats['aln_len'] <= ql)
    assert np.all(stats['n_regions'] >= 1)
    # a second run is bit-identical (no dependence on atomics / scheduling order)
    assert np.array_equal(stats, gs.lz_align(tasks))
    # self alignment: one region covering the genome
    self_tasks = np.array([(i, i) for i in range(0, len(gs), 97)], dtype=api.TASK_DTYPE)
    st = gs.lz_align(self_tasks)
    assert np.array_equal(st['n_match'], lens[self_tasks['q']]) and np.all(st['n_regions'] == 1)
    # oracle on a sample of the tasks
    idx = np.random.default_rng(0).choice(len(tasks), 60, replace=False)
    for i in idx:
        q, r = int(tasks[i]['q']), int(tasks[i]['r'])
        ref = orc.lz_pair_stat(codes[offsets[q]:offsets[q + 1]], codes[offsets[r]:offsets[r + 1]])
        assert ref == tuple(int(x) for x in stats[i]), (q, r)


def test_imgvr_like_properties():
    """BASELINE configs[2] shape (mixed 5-200 kb contigs, families of 1-20), reduced to 600 contigs:
    the prefilter passes exactly the within-family pairs, alignment rows obey the size-independent
    invariants, and a sample of tasks (long and short references: LDS sections / plain) equals the oracle."""
    codes, offsets, names, fam = synth.make_contigs(600, seed=2)
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    sizes, pairs = gs.kmer_shared(k=25, min_shared=20)
    assert np.all(fam[pairs['a']] == fam[pairs['b']])
    n_expected = sum(c * (c - 1) // 2 for c in np.bincount(fam))
    # (a few pairs of short, strongly diverged contigs keep < 20 shared 25-mers)
    assert 0.98 * n_expected <= len(pairs) <= n_expected
    tasks = gs.align_tasks(pairs)
    stats = gs.lz_align(tasks)
    lens = gs.lengths()
    assert np.all(stats['n_match'] <= stats['aln_len']) and np.all(stats['aln_len'] <= lens[tasks['q']])
    assert np.array_equal(stats, gs.lz_align(tasks))
    order = np.argsort(lens[tasks['r']])
    idx = np.concatenate([order[:10], order[-10:], np.random.default_rng(1).choice(len(tasks), 20, replace=False)])
    for i in idx:
        q, r = int(tasks[i]['q']), int(tasks[i]['r'])
        ref = orc.lz_pair_stat(codes[offsets[q]:offsets[q + 1]], codes[offsets[r]:offsets[r + 1]])
        assert ref == tuple(int(x) for x in stats[i]), (q, r)


def _scrambled_key(kmer_codes, k):
    """The product's sort key of a k-mer (vg_prefilter.hip: canonical, first base most significant,
    times an odd constant mod 4^k) -- only used to BUILD an adversarial input, not to check results."""
    fwd = 0
    for c in kmer_codes:
        fwd = (fwd << 2) | int(c)
    rc = 0
    for c in kmer_codes[::-1]:
        rc = (rc << 2) | (3 - int(c))
    return (min(fwd, rc) * 0x9E3779B97F4A7C15) & ((1 << (2 * k)) - 1)


def test_two_frequent_kmers_in_one_prefix_group(tmp_path):
    """Two k-mers that each occur 1 400 times AND share the 24-bit radix prefix.  Own pipeline: their bucket exceeds
    both LDS variants and is finished by k_bucket_big (no radix sort).  General path (VG_INDEX_PATH=radix, a second
    process): their equal-prefix group is longer than a staged window and mixed, so that path must go through the
    full-bit sort + galloping run search.  Both equal the oracle."""
    import subprocess
    import sys
    k = 25
    rng = np.random.default_rng(21)
    seen = {}
    pair = None
    while pair is None:
        km = rng.integers(0, 4, size=k, dtype=np.uint8)
        pre = _scrambled_key(km, k) >> (2 * k + 1 - 24)
        if pre in seen and not np.array_equal(seen[pre], km):
            pair = (seen[pre], km)
        seen[pre] = km
    seqs = []
    for i in range(1400):
        fl = [rng.integers(0, 4, size=int(rng.integers(40, 90)), dtype=np.uint8) for _ in range(3)]
        seqs.append(np.concatenate([fl[0], pair[0], fl[1], pair[1], fl[2]]))
    offsets = np.zeros(len(seqs) + 1, dtype=np.int64); offsets[1:] = np.cumsum([len(s) for s in seqs])
    codes = np.concatenate(seqs)
    gs = api.GenomeSet.from_codes(codes, offsets)
    api.profile_enable(True); api.profile_reset()
    try:
        pairs = _check_prefilter(codes, offsets, gs, k)
        scopes = {e['name'] for e in api.profile_get()}
    finally:
        api.profile_enable(False)
    assert len(pairs) == 1400 * 1399 // 2 and int(pairs['shared'].min()) >= 2
    assert 'bucket_big' in scopes and 'radix_sort_pairs' not in scopes, scopes
    np.save(tmp_path / 'codes.npy', codes); np.save(tmp_path / 'offsets.npy', offsets)
    np.save(tmp_path / 'pairs.npy', np.sort((pairs['a'].astype(np.int64) << 40) | (pairs['b'].astype(np.int64) << 20) | pairs['shared']))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from vclust_amd import api\n"
            "c = np.load(%r); o = np.load(%r); gs = api.GenomeSet.from_codes(c, o)\n"
            "api.profile_enable(True); api.profile_reset(); s, p = gs.kmer_shared(k=25)\n"
            "sc = {e['name'] for e in api.profile_get()}\n"
            "assert 'index_long_runs' in sc and 'index_runs_general' in sc and 'radix_sort_pairs' in sc, sc\n"
            "q = np.sort((p['a'].astype(np.int64) << 40) | (p['b'].astype(np.int64) << 20) | p['shared'])\n"
            "assert np.array_equal(q, np.load(%r)); print('general ok')\n"
            % (str(__import__('pathlib').Path(__file__).resolve().parent.parent), str(tmp_path / 'codes.npy'), str(tmp_path / 'offsets.npy'),
               str(tmp_path / 'pairs.npy')))
    r = subprocess.run([sys.executable, '-c', code], env=dict(__import__('os').environ, VG_DEV_SWITCHES='1', VG_INDEX_PATH='radix'), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0 and 'general ok' in r.stdout, (r.stdout[-300:], r.stderr[-1500:])


def _core_set(n, rng, core_len=40, flank=(30, 60)):
    """n genomes = random flank + shared core + random flank; also the 16 flank bases next to the core."""
    core = rng.integers(0, 4, size=core_len, dtype=np.uint8)
    seqs, left, right = [], [], []
    for i in range(n):
        f1 = rng.integers(0, 4, size=int(rng.integers(*flank)), dtype=np.uint8)
        f2 = rng.integers(0, 4, size=int(rng.integers(*flank)), dtype=np.uint8)
        seqs.append(np.concatenate([f1, core, f2])); left.append(f1[-16:][::-1]); right.append(f2[:16])
    offsets = np.zeros(n + 1, dtype=np.int64); offsets[1:] = np.cumsum([len(s) for s in seqs])
    return np.concatenate(seqs), offsets, np.array(left), np.array(right)


def _common_prefix_len(m):
    """m: n x d bases; -> n x n matrix of common-prefix lengths of the rows."""
    n, d = m.shape
    run = np.ones((n, n), dtype=bool); out = np.zeros((n, n), dtype=np.int32)
    for j in range(d):
        run &= m[:, j][:, None] == m[:, j][None, :]
        out += run
    return out


def test_genome_with_thousands_of_partners():
    """Rows of the SpGEMM whose partner set overflows the 2^11-slot LDS table (second try, 2^13 slots)
    and the 2^13-slot one (dense counters): 2 100 genomes against the oracle, 7 500 by a closed form."""
    rng = np.random.default_rng(31)
    codes, offsets, _, _ = _core_set(2100, rng)
    gs = api.GenomeSet.from_codes(codes, offsets)
    api.profile_enable(True); api.profile_reset()
    try:
        sizes, pairs = gs.kmer_shared(k=25)
        scopes = {e['name'] for e in api.profile_get()}
    finally:
        api.profile_enable(False)
    assert 'spgemm_rows_wide' in scopes
    osizes, opairs = orc.shared_all(codes, offsets, k=25)
    assert list(sizes) == list(osizes)
    key = (pairs['a'].astype(np.int64) << 32) | pairs['b']
    okey = np.array([(a << 32) | b for a, b in opairs], dtype=np.int64); oval = np.array(list(opairs.values()), dtype=np.int64)
    o1, o2 = np.argsort(key), np.argsort(okey)
    assert np.array_equal(key[o1], okey[o2]) and np.array_equal(pairs['shared'][o1].astype(np.int64), oval[o2])
    # 7 500 genomes sharing a 40-base core: a pair shares the 25-mers of core + the flank bases that happen to
    # agree next to it: 16 + common suffix of the left flanks + common prefix of the right flanks
    n = 7500
    codes, offsets, left, right = _core_set(n, rng)
    gs = api.GenomeSet.from_codes(codes, offsets)
    api.profile_enable(True); api.profile_reset()
    try:
        sizes, pairs = gs.kmer_shared(k=25)
        scopes = {e['name'] for e in api.profile_get()}
    finally:
        api.profile_enable(False)
    assert 'spgemm_dense_rows' in scopes
    assert len(pairs) == n * (n - 1) // 2 and np.all(pairs['a'] > pairs['b'])
    assert len(np.unique((pairs['a'].astype(np.int64) << 32) | pairs['b'])) == len(pairs)
    expect = 16 + _common_prefix_len(left) + _common_prefix_len(right)
    assert np.array_equal(pairs['shared'].astype(np.int64), expect[pairs['a'], pairs['b']])


def test_shards_on_low_complexity_sequence():
    """Poly-A / tandem repeats put all k-mers of a stretch into ONE shard: its staging slots overflow and the
    call must fall back to the recomputing emit pass; the shards still add up to the oracle's counts."""
    rng = np.random.default_rng(77)
    a = rng.integers(0, 4, size=6000, dtype=np.uint8)
    unit = rng.integers(0, 4, size=5, dtype=np.uint8)
    seqs = [np.concatenate([a[:2000], np.zeros(3000, np.uint8), a[2000:4000]]),          # poly-A block
            np.concatenate([a[1000:3000], np.tile(unit, 500), a[:1500]]),                  # 5-base tandem repeat
            a.copy()]
    offsets = np.zeros(len(seqs) + 1, dtype=np.int64); offsets[1:] = np.cumsum([len(s) for s in seqs])
    codes = np.concatenate(seqs)
    gs = api.GenomeSet.from_codes(codes, offsets)
    osizes, opairs = orc.shared_all(codes, offsets, k=25)
    tot = np.zeros(len(seqs), dtype=np.int64); acc = {}; scopes = set()
    api.profile_enable(True); api.profile_reset()
    try:
        for s in range(8):
            sz, pr = gs.kmer_shared(k=25, shard=s, n_shards=8)
            tot += sz
            for p in pr:
                acc[(int(p['a']), int(p['b']))] = acc.get((int(p['a']), int(p['b'])), 0) + int(p['shared'])
        scopes = {e['name'] for e in api.profile_get()}
    finally:
        api.profile_enable(False)
    assert list(tot) == list(osizes) and acc == opairs
    assert 'kmer_emit_recompute' in scopes and 'kmer_emit' in scopes, scopes


def test_genomes_ending_on_a_block_boundary():
    """Genomes whose length is an exact multiple of every block size the layout may choose (64 .. 4 096): a
    k-mer window that ran from the end of one genome into the start of the next would be a false k-mer -- the
    sequence joined across the boundary is planted in a third genome, which must share nothing with the first
    two beyond what the oracle counts.  (Every genome is followed by at least one masked padding base.)"""
    rng = np.random.default_rng(77)
    a = rng.integers(0, 4, 8192).astype(np.uint8)
    b = rng.integers(0, 4, 4096).astype(np.uint8)
    c = rng.integers(0, 4, 8192).astype(np.uint8)
    c[1000:1060] = np.concatenate([a[-30:], b[:30]])          # the junction a|b, inside c
    d = np.concatenate([b[-40:], c[:40], rng.integers(0, 4, 4016).astype(np.uint8)])     # the junction b|c, inside d (4 096 long)
    seqs = [a, b, c, d]
    codes = np.concatenate(seqs)
    offsets = np.concatenate([[0], np.cumsum([len(x) for x in seqs])]).astype(np.int64)
    gs = api.GenomeSet.from_codes(codes, offsets)
    for k in (15, 25, 30):
        _check_prefilter(codes, offsets, gs, k)
    # the same through the dense bucket pipeline: many copies so that the set is large enough for it
    reps = 12
    seqs2 = []
    for r in range(reps):
        for x in seqs:
            y = x.copy(); y[(17 * r) % len(y)] ^= 1
            seqs2.append(y)
    codes2 = np.concatenate(seqs2)
    offsets2 = np.concatenate([[0], np.cumsum([len(x) for x in seqs2])]).astype(np.int64)
    gs2 = api.GenomeSet.from_codes(codes2, offsets2)
    api.profile_enable(True); api.profile_reset()
    _check_prefilter(codes2, offsets2, gs2, 25)
    scopes = {e['name'] for e in api.profile_get()}
    api.profile_enable(False)
    assert 'bucket_sort_runs' in scopes

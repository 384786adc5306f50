"""BASELINE.json configs[1..4] through the HIP path, each to the parity bar its size allows:
configs[1] in full against the oracle (every row, files byte-compared); configs[2] at full size
(properties + an oracle sample); configs[3] / [4] at a reduced count with their own flag sets
(size-independent properties, determinism, oracle sample).  Everything calls through the C ABI."""
import filecmp
import pathlib
import sys

import numpy as np
import pytest

import oracle_lib as orc
from vclust_amd import api, synth

pytestmark = pytest.mark.gpu

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _rows_dict(tasks, stats):
    return {(int(t['q']), int(t['r'])): (int(s['n_match']), int(s['aln_len']), int(s['n_regions'])) for t, s in zip(tasks, stats)}


def _gpu_path(gs, k, min_kmers, min_ident):
    sizes, pairs = gs.kmer_shared(k=k, min_shared=min_kmers)
    cand = gs.filter_pairs(sizes, pairs, k=k, min_kmers=min_kmers, min_ident=min_ident)
    tasks = gs.align_tasks(cand)
    return sizes, pairs, cand, tasks, gs.lz_align(tasks)


def test_config1_phage_1k_every_row_and_files(tmp_path):
    """configs[1]: 1 000 x 40 kb, k=25: all 9 000 ordered alignments equal the oracle's rows, and the files
    written by the product from a FASTA on disk (fltr.txt, ani.tsv, ani.ids.tsv) equal the oracle CLI's byte for byte."""
    codes, offsets, names, _ = synth.make_workload('phage-1k')
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    sizes, pairs, cand, tasks, stats = _gpu_path(gs, 25, 20, 0.7)
    assert len(tasks) == 9000
    ref = orc.path_rows(codes, offsets, k=25, min_kmers=20, min_ident=0.7)
    want = {(int(r['q']), int(r['r'])): (int(r['n_match']), int(r['aln_len']), int(r['n_regions'])) for r in ref}
    assert _rows_dict(tasks, stats) == want
    fa = tmp_path / 'p1k.fna'
    synth.write_fasta(fa, codes, offsets, names)
    api.prefilter([fa], tmp_path / 'fltr.txt', is_multifasta=True)
    api.align([fa], tmp_path / 'ani.tsv', is_multifasta=True, columns=api.ALIGN_FIELDS[:11], filter_path=tmp_path / 'fltr.txt')
    orc.run_cli('prefilter', '-o', tmp_path / 'o_fltr.txt', fa)
    orc.run_cli('align', '--filter', tmp_path / 'o_fltr.txt', '0', '-o', tmp_path / 'o_ani.tsv', fa)
    assert filecmp.cmp(tmp_path / 'fltr.txt', tmp_path / 'o_fltr.txt', shallow=False)
    assert filecmp.cmp(tmp_path / 'ani.tsv', tmp_path / 'o_ani.tsv', shallow=False)
    assert filecmp.cmp(tmp_path / 'ani.ids.tsv', tmp_path / 'o_ani.ids.tsv', shallow=False)


def test_config2_imgvr_10k_full_size():
    """configs[2]: 10 000 mixed 5-200 kb contigs, --min-ident 0.7: within-family pairs only, row invariants,
    determinism, and an oracle sample over short, long and random references."""
    codes, offsets, names, _ = synth.make_workload('imgvr-10k')
    fam = np.array([int(n[3:9]) for n in names])
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    sizes, pairs, cand, tasks, stats = _gpu_path(gs, 25, 20, 0.7)
    assert len(gs) == 10000
    assert np.all(fam[cand['a']] == fam[cand['b']])
    n_expected = sum(c * (c - 1) // 2 for c in np.bincount(fam))
    assert 0.97 * n_expected <= len(cand) <= n_expected
    lens = gs.lengths()
    assert np.all(stats['n_match'] <= stats['aln_len']) and np.all(stats['aln_len'] <= lens[tasks['q']])
    assert np.array_equal(stats, gs.lz_align(tasks))
    # RANGE shards at this size (32 768-position tiles, 8-byte level-1 records carrying row numbers): three uneven
    # digit ranges add up to the single pass
    tot = np.zeros(len(gs), dtype=np.int64); acc = {}
    for sh in range(3):
        sz, pr = gs.kmer_shared(k=25, shard=sh, n_shards=3)
        tot += sz
        for p in pr:
            acc[(int(p['a']), int(p['b']))] = acc.get((int(p['a']), int(p['b'])), 0) + int(p['shared'])
    assert np.array_equal(tot, sizes)
    assert {kv: c for kv, c in acc.items() if c >= 20} == {(int(p['a']), int(p['b'])): int(p['shared']) for p in pairs}
    order = np.argsort(lens[tasks['r']])
    idx = np.concatenate([order[:8], order[-8:], np.random.default_rng(3).choice(len(tasks), 24, replace=False)])
    for i in idx:
        q, r = int(tasks[i]['q']), int(tasks[i]['r'])
        assert orc.lz_pair_stat(codes[offsets[q]:offsets[q + 1]], codes[offsets[r]:offsets[r + 1]]) == tuple(int(x) for x in stats[i]), (q, r)


def test_config3_phage_100k_slice():
    """configs[3] (100 000 x 40 kb) at 20 000 genomes: same generator and flags as the bench's default workload;
    exactly the within-family pairs, row invariants, bit-identical second run, oracle sample.  (The full
    100 000 run is the bench line; its pair count is checked there: 450 000.)"""
    nf = 2000
    codes, offsets, names, _ = synth.make_workload('phage-100k', nf)
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    sizes, pairs, cand, tasks, stats = _gpu_path(gs, 25, 20, 0.7)
    assert {(int(p['a']), int(p['b'])) for p in cand} == {(int(p['a']), int(p['b'])) for p in synth.family_pairs(nf, 10)}
    lens = gs.lengths()
    assert np.all(stats['n_regions'] >= 1) and np.all(stats['n_match'] <= stats['aln_len']) and np.all(stats['aln_len'] <= lens[tasks['q']])
    assert np.array_equal(stats, gs.lz_align(tasks))
    # the cold CLI's way through the prefilter: eight RANGE sub-shards inside one call (partial lists summed on the device)
    from vclust_amd import _lib
    _lib.load().vg_set_subshards(8)
    try:
        sizes8, pairs8 = gs.kmer_shared(k=25, min_shared=20)
    finally:
        _lib.load().vg_set_subshards(0)
    o1, o8 = np.lexsort((pairs['b'], pairs['a'])), np.lexsort((pairs8['b'], pairs8['a']))
    assert np.array_equal(sizes, sizes8) and np.array_equal(pairs[o1], pairs8[o8])
    for i in np.random.default_rng(5).choice(len(tasks), 40, replace=False):
        q, r = int(tasks[i]['q']), int(tasks[i]['r'])
        assert orc.lz_pair_stat(codes[offsets[q]:offsets[q + 1]], codes[offsets[r]:offsets[r + 1]]) == tuple(int(x) for x in stats[i]), (q, r)


def test_config4_contigs_shape_with_dereplication_flags(tmp_path):
    """configs[4] shape (log-uniform 2-100 kb contigs, --min-kmers 30, then --out-ani 0.95 --out-qcov 0.85:
    the reference's large.yml:65-72 flag set) at 100 000 contigs: prefilter invariants at that size, and on a
    4 000-contig slice the product's files equal the oracle CLI's under the same flags."""
    codes, offsets, names, _ = synth.make_workload('contigs-1M', 100000)
    fam = np.array([int(n[3:9]) for n in names])
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    sizes, pairs = gs.kmer_shared(k=25, min_shared=30)
    cand = gs.filter_pairs(sizes, pairs, k=25, min_kmers=30, min_ident=0.7)
    assert len(gs) == 100000 and np.all(fam[cand['a']] == fam[cand['b']]) and np.all(cand['shared'] >= 30)
    n_expected = sum(c * (c - 1) // 2 for c in np.bincount(fam))
    assert 0.9 * n_expected <= len(cand) <= n_expected          # short, strongly diverged contigs keep < 30 shared 25-mers
    tasks = gs.align_tasks(cand)
    stats = gs.lz_align(tasks)
    lens = gs.lengths()
    assert np.all(stats['n_match'] <= stats['aln_len']) and np.all(stats['aln_len'] <= lens[tasks['q']])
    # slice: byte-compare the files under the dereplication flags
    n = 4000
    fa = tmp_path / 'c4k.fna'
    synth.write_fasta(fa, codes[:offsets[n]], offsets[:n + 1], names[:n])
    api.prefilter([fa], tmp_path / 'fltr.txt', is_multifasta=True, min_kmers=30)
    api.align([fa], tmp_path / 'ani.tsv', is_multifasta=True, columns=api.ALIGN_FIELDS[:11], filter_path=tmp_path / 'fltr.txt',
              out_filters={'ani': 0.95, 'qcov': 0.85})
    orc.run_cli('prefilter', '--min-kmers', 30, '-o', tmp_path / 'o_fltr.txt', fa)
    orc.run_cli('align', '--filter', tmp_path / 'o_fltr.txt', '0', '--out-ani', 0.95, '--out-qcov', 0.85, '-o', tmp_path / 'o_ani.tsv', fa)
    assert filecmp.cmp(tmp_path / 'fltr.txt', tmp_path / 'o_fltr.txt', shallow=False)
    assert filecmp.cmp(tmp_path / 'ani.tsv', tmp_path / 'o_ani.tsv', shallow=False)
    assert sum(1 for _ in open(tmp_path / 'ani.tsv')) > 100


def test_many_regions_per_task(tmp_path):
    """--out-aln on strongly diverged pairs: far more than 64 regions per ordered pair on average.  The region
    buffer is sized from the rows (no fixed capacity): every region arrives, in the oracle's multiset."""
    codes, offsets, names = synth.make_families(3, 6, length=60000, seed=17, p_lo=0.10, p_hi=0.16, n_indels=40)
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    tasks = gs.align_tasks(synth.family_pairs(3, 6))
    stats, regions = gs.lz_align(tasks, want_regions=True)
    assert len(regions) == int(stats['n_regions'].sum()) and len(regions) > 64 * len(tasks)
    fa = tmp_path / 'div.fna'
    synth.write_fasta(fa, codes, offsets, names)
    api.align([fa], tmp_path / 'ani.tsv', is_multifasta=True, columns=api.ALIGN_FIELDS[:11], out_aln=tmp_path / 'aln.tsv')
    orc.run_cli('align', '-o', tmp_path / 'o.tsv', '--out-aln', tmp_path / 'o_aln.tsv', fa)
    assert filecmp.cmp(tmp_path / 'ani.tsv', tmp_path / 'o.tsv', shallow=False)
    assert sorted(open(tmp_path / 'aln.tsv').read().splitlines()) == sorted(open(tmp_path / 'o_aln.tsv').read().splitlines())


def test_placement_trials_are_opt_in_and_change_nothing(tmp_path):
    """vg_set_placement_trials: by default the first dense pass of a process over a large set (>= 2^30 padded bases) runs ONCE;
    a process that opts in repeats it on fresh workspaces (the allocator trace shows the trials) and returns the same sizes and
    pairs.  27 000 x 40 kb genomes, each variant in a process of its own."""
    import os
    import subprocess
    code = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
from vclust_amd import api, synth
codes, offsets, names, _ = synth.make_workload('phage-100k', 2700)
gs = api.GenomeSet.from_codes(codes, offsets, names)
api.set_placement_trials(int(sys.argv[1]))
sizes, pairs = gs.kmer_shared(k=25, min_shared=20)
pairs = np.sort(pairs, order=['a', 'b'])
assert len(pairs) == 2700 * 45
print('digest', hashlib.sha256(sizes.tobytes() + pairs.tobytes()).hexdigest())
""" % str(ROOT)
    out = {}
    for trials in (1, 3):
        p = subprocess.run([sys.executable, '-c', code, str(trials)], env=dict(os.environ, VG_ALLOC_TRACE='1'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        out[trials] = (p.stdout.strip().splitlines()[-1], p.stderr.count('[vg placement] trial'))
    assert out[1][1] == 0 and out[3][1] >= 1, out          # (no trial unless asked for)
    assert out[1][0] == out[3][0] and out[1][0].startswith('digest')


def test_out_aln_one_parse_against_its_checkers(tmp_path):
    """--out-aln comes from ONE parse (regions written into chunks behind a cursor, placed once the rows are known).  Its
    checkers, each in a process of its own (developer switches are read once): the two-pass scheme it replaces (rows first,
    then a second parse into an arena of known size), the general kernel instead of the specialised one, and a first arena
    far too small (64 records), which makes the batch repeat itself with the size the cursor reported.  Regions (in the order
    the library returns them: sorted task list, query order) and rows must be the same arrays in all four, and the rows must
    equal those of a call without regions."""
    import os
    import subprocess
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from vclust_amd import api, synth
codes, offsets, names = synth.make_families(40, 6, length=30000, seed=29, p_lo=0.02, p_hi=0.16, n_indels=25)
gs = api.GenomeSet.from_codes(codes, offsets, names)
tasks = gs.align_tasks(synth.family_pairs(40, 6))
plain = gs.lz_align(tasks)
stats, regions = gs.lz_align(tasks, want_regions=True)
assert np.array_equal(plain, stats), 'rows with and without regions differ'
assert len(regions) == int(stats['n_regions'].sum())
np.savez(sys.argv[1], stats=stats, regions=regions)
""" % str(ROOT)
    out = {}
    for tag, env in (('one_parse', {}), ('two_pass', dict(VG_LZ_REGIONS='two-pass')), ('general', dict(VG_LZ_KERNEL='general')),
                     ('tiny_arena', dict(VG_LZ_ARENA='64'))):
        f = tmp_path / f'{tag}.npz'
        p = subprocess.run([sys.executable, '-c', code, str(f)], env=dict(os.environ, VG_DEV_SWITCHES='1', **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert p.returncode == 0, (tag, p.stderr[-2000:])
        d = np.load(f); out[tag] = (d['stats'], d['regions'])
    base = out['one_parse']
    assert len(base[1]) > 20 * len(base[0])
    # within a task the regions come in query order and do not overlap
    rg = base[1]
    same = rg['task'][1:] == rg['task'][:-1]
    assert np.all(rg['qstart'][1:][same] > rg['qend'][:-1][same])
    for tag, (st, rgs) in out.items():
        assert np.array_equal(st, base[0]) and np.array_equal(rgs, base[1]), tag


@pytest.mark.parametrize('k', [15, 30])
def test_bucket_pipeline_key_widths(k):
    """The own index pipeline at the ends of the k range with two partition levels: k = 15 (30 key bits: the
    narrow 8-byte level-2 records with few bits left) and k = 30 (60 key bits: the wide 12-byte records),
    2 000 genomes x 8 kb; sizes and counts equal the oracle's, dense and as four / eight shards (both shard sources)."""
    codes, offsets, names = synth.make_families(200, 10, length=8000, seed=23)
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    osizes, opairs = orc.shared_all(codes, offsets, k=k)
    api.profile_enable(True); api.profile_reset()
    sizes, pairs = gs.kmer_shared(k=k)
    scopes = {e['name'] for e in api.profile_get()}
    api.profile_enable(False)
    assert 'kmer_partition2' in scopes and 'bucket_sort_runs' in scopes and 'radix_sort_pairs' not in scopes
    assert list(sizes) == list(osizes)
    assert {(int(p['a']), int(p['b'])): int(p['shared']) for p in pairs} == opairs
    for n_shards in (4, 8):           # four shards: dense source with the shard filter; eight: compact source
        tot = np.zeros(len(gs), dtype=np.int64); acc = {}
        for sh in range(n_shards):
            sz, pr = gs.kmer_shared(k=k, shard=sh, n_shards=n_shards)
            tot += sz
            for p in pr:
                acc[(int(p['a']), int(p['b']))] = acc.get((int(p['a']), int(p['b'])), 0) + int(p['shared'])
        assert list(tot) == list(osizes) and acc == opairs


@pytest.mark.parametrize('world', [2, 3, 8])
def test_sliced_scan_equals_the_replicated_scan(world):
    """The multi-GPU form of a RANGE shard (k_slice_scan: rank r scans 1/world of the BASES, the kept masks and level-1
    counts of every rank's k-mer range travel to it) with the peers' slices computed by this process: the pairs of every
    rank are exactly those of the replicated scan, and the ranks' partial set sizes and counts add up to the oracle's.
    4.5 M bases: two partition levels, a last super-tile that is not full, digit ranges that do not divide 2 048 at 3."""
    codes, offsets, names = synth.make_families(100, 5, length=9000, seed=31)
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    osizes, opairs = orc.shared_all(codes, offsets, k=25)
    tot = np.zeros(len(gs), dtype=np.int64); acc = {}
    for r in range(world):
        sz0, pr0 = gs.kmer_shared(k=25, shard=r, n_shards=world)
        api.set_range_scan(1)
        try:
            api.profile_enable(True); api.profile_reset()
            sz1, pr1 = gs.kmer_shared(k=25, shard=r, n_shards=world)
            scopes = {e['name']: e for e in api.profile_get()}
        finally:
            api.profile_enable(False); api.set_range_scan(0)
        assert scopes['emulated_peer_scan']['launches'] == world - 1
        key = lambda pr: sorted((int(p['a']), int(p['b']), int(p['shared'])) for p in pr)
        assert key(pr0) == key(pr1) and len(pr1) > 0
        tot += sz1
        for p in pr1:
            acc[(int(p['a']), int(p['b']))] = acc.get((int(p['a']), int(p['b'])), 0) + int(p['shared'])
    assert list(tot) == list(osizes) and acc == opairs


def test_bucket_pipeline_large_buckets():
    """200 near-identical genomes: every k-mer is shared by ~200 of them, so some final buckets exceed the 1 536
    entries of the ordinary bucket kernel; exactly those are queued for the 6 144-entry variant (still the own
    pipeline, no radix sort, nothing redone) and the counts equal the oracle's."""
    codes, offsets, names = synth.make_families(1, 200, length=40000, seed=5)
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    osizes, opairs = orc.shared_all(codes, offsets, k=25)
    api.profile_enable(True); api.profile_reset()
    sizes, pairs = gs.kmer_shared(k=25)
    prof = api.profile_get()
    api.profile_enable(False)
    scopes = {e['name']: e for e in prof}
    assert 'radix_sort_pairs' not in scopes and scopes['bucket_sort_runs']['launches'] == 1
    assert scopes['bucket_sort_runs_wide']['launches'] == 1 and 'bucket_big' not in scopes
    assert list(sizes) == list(osizes)
    assert {(int(p['a']), int(p['b'])): int(p['shared']) for p in pairs} == opairs


def _real_shaped_set(n_contigs, n_block, n_polya, seed, block_len=300, polya_len=120):
    """log-uniform contigs (the contigs-1M generator) of which n_block carry one shared block (a conserved gene:
    a k-mer present in n_block genomes) and n_polya a poly-A run (one k-mer, many times per genome)."""
    codes, offsets, names, _ = synth.make_workload('contigs-1M', n_contigs)
    codes = codes.copy()
    rng = np.random.default_rng(seed)
    block = rng.integers(0, 4, block_len).astype(np.uint8)
    lens = np.diff(offsets)
    ok = np.flatnonzero(lens > 2 * (block_len + polya_len))
    for g in rng.choice(ok, n_block, replace=False):
        p = int(offsets[g]) + int(rng.integers(0, lens[g] - block_len))
        codes[p:p + block_len] = block
    for g in rng.choice(ok, n_polya, replace=False):
        p = int(offsets[g]) + int(rng.integers(0, lens[g] - polya_len))
        codes[p:p + polya_len] = 0
    return codes, offsets, names


def test_high_multiplicity_kmers_stay_in_the_own_pipeline():
    """Real-shaped data: 6 000 contigs of which 1 500 share a 300-bp block and 800 carry a
    120-base poly-A run.  The buckets holding those k-mers exceed both LDS variants and are finished by
    k_bucket_big; the call never falls back to the rocPRIM radix sort, and sizes and counts equal the oracle's."""
    codes, offsets, names = _real_shaped_set(6000, 1500, 800, seed=9)
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    osizes, opairs = orc.shared_all(codes, offsets, k=25)
    api.profile_enable(True); api.profile_reset()
    sizes, pairs = gs.kmer_shared(k=25)
    scopes = {e['name']: e for e in api.profile_get()}
    api.profile_enable(False)
    assert 'radix_sort_pairs' not in scopes and 'radix_sort_full' not in scopes and 'bucket_big' in scopes
    assert list(sizes) == list(osizes)
    assert {(int(p['a']), int(p['b'])): int(p['shared']) for p in pairs} == opairs
    # the same through k-mer range shards (compact source) and a fraction
    tot = np.zeros(len(gs), dtype=np.int64); acc = {}
    for sh in range(8):
        sz, pr = gs.kmer_shared(k=25, shard=sh, n_shards=8)
        tot += sz
        for p in pr:
            acc[(int(p['a']), int(p['b']))] = acc.get((int(p['a']), int(p['b'])), 0) + int(p['shared'])
    assert list(tot) == list(osizes) and acc == opairs


def test_high_multiplicity_index_time_at_100k_contigs():
    """100 000 contigs, 5 000 sharing a 300-bp block and 2 000 with a poly-A run: the index stage (everything up to
    the SpGEMM) stays within 1.3x of the same set without them, no radix sort; the pairs are exactly the family
    pairs plus the pairs of block carriers (and the poly-A pairs that reach min-kmers)."""
    base, offsets, names, _ = synth.make_workload('contigs-1M', 100000)

    codes, _, _ = _real_shaped_set(100000, 5000, 2000, seed=10)
    sets = [api.GenomeSet.from_codes(c, offsets, names) for c in (base, codes)]

    def run(gs):
        api.profile_enable(True); api.profile_reset()
        sizes, pairs = gs.kmer_shared(k=25, min_shared=30)
        prof = {e['name']: e for e in api.profile_get()}
        api.profile_enable(False)
        idx_ms = sum(e['total_ms'] for n, e in prof.items() if n.startswith(('kmer_partition', 'bucket_')))
        return sizes, pairs, prof, idx_ms
    for gs in sets:
        gs.kmer_shared(k=25, min_shared=30)            # warm-up (allocator)
    # kernel-scope times of the two sets in ALTERNATING runs, best of three each: where the driver places the buffers
    # moves the scattering kernels by several per cent from one allocation to the next, a single pair
    # of runs compares two placements as much as two inputs
    t = [[], []]
    for _ in range(3):
        for i, gs in enumerate(sets):
            out = run(gs)
            t[i].append(out[3])
            if i == 0: s0, p0, prof0, _ = out
            else: s1, p1, prof1, _ = out
    t0, t1 = min(t[0]), min(t[1])
    assert 'radix_sort_pairs' not in prof1 and 'bucket_big' in prof1
    assert t1 <= 1.3 * t0, (t, t0, t1)
    assert len(p1) >= len(p0) + 5000 * 4999 // 2 * 0.99


def test_set_without_any_kmer():
    """>= 65 536 padded bases and not one valid k-mer (every record shorter than k, or all N): the
    bucket pipeline returns empty outputs the SpGEMM can read (no null row pointers)."""
    n = 1200
    lens = np.full(n, 20, dtype=np.int64); lens[::2] = 24
    offsets = np.concatenate([[0], np.cumsum(lens)])
    codes = np.random.default_rng(1).integers(0, 4, int(offsets[-1])).astype(np.uint8)
    gs = api.GenomeSet.from_codes(codes, offsets, ['s%d' % i for i in range(n)])
    sizes, pairs = gs.kmer_shared(k=25)
    assert not np.any(sizes) and len(pairs) == 0
    codes = np.full(80000, 4, dtype=np.uint8)
    offsets = np.array([0, 30000, 80000])
    gs = api.GenomeSet.from_codes(codes, offsets, ['n1', 'n2'])
    sizes, pairs = gs.kmer_shared(k=25)
    assert not np.any(sizes) and len(pairs) == 0


@pytest.mark.slow
def test_config4_contigs_1M_full_size():
    """BASELINE configs[4] AT ITS SIZE: 1 000 000 contigs (25 Gbp, log-uniform 2-100 kb, families geometric(0.2) <= 20),
    --min-kmers 30 (reference large.yml:65-72) on ONE MI355X.  25 G positions exceed the 32-bit row numbering of one
    pass, so the prefilter runs its automatic k-mer sub-shard loop (7 passes over the bases, partial counts summed on
    the device).  Size-independent properties + an oracle sample:
      * every candidate pair lies inside one family, shared >= 30, and their number is within 10 % of the family pairs
      * a second run returns bit-identical arrays (prefilter and align rows)
      * set sizes and shared counts of sampled genomes / pairs equal the oracle's k-mer sets
      * LZ rows of sampled tasks equal the oracle's; row invariants hold for all 7 M rows
      * the dereplication cut (--out-ani 0.95 --out-qcov 0.85) keeps a non-trivial subset."""
    codes, offsets, names, _ = synth.make_workload('contigs-1M', 1000000)
    assert len(names) == 1000000 and offsets[-1] > 24e9
    fam = np.array([int(n.split('_')[0][3:]) for n in names])
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    sizes, pairs = gs.kmer_shared(k=25, min_shared=30)
    cand = gs.filter_pairs(sizes, pairs, k=25, min_kmers=30, min_ident=0.7)
    assert np.all(fam[cand['a']] == fam[cand['b']]) and np.all(cand['shared'] >= 30)
    n_expected = sum(c * (c - 1) // 2 for c in np.bincount(fam))
    assert 0.9 * n_expected <= len(cand) <= n_expected
    tasks = gs.align_tasks(cand)
    stats = gs.lz_align(tasks)
    lens = gs.lengths()
    assert len(stats) == 2 * len(cand)
    assert np.all(stats['n_match'] <= stats['aln_len']) and np.all(stats['aln_len'] <= lens[tasks['q']])
    # determinism
    sizes2, pairs2 = gs.kmer_shared(k=25, min_shared=30)
    assert np.array_equal(sizes, sizes2)
    o1 = np.lexsort((pairs['b'], pairs['a'])); o2 = np.lexsort((pairs2['b'], pairs2['a']))
    assert np.array_equal(pairs[o1], pairs2[o2])
    assert np.array_equal(stats, gs.lz_align(tasks))
    # oracle sample: k-mer sets of 40 pairs, LZ rows of 24 tasks
    rng = np.random.default_rng(4)
    for i in rng.choice(len(cand), 40, replace=False):
        a, b = int(cand[i]['a']), int(cand[i]['b'])
        ka = orc.kmer_set(codes[offsets[a]:offsets[a + 1]], 25); kb = orc.kmer_set(codes[offsets[b]:offsets[b + 1]], 25)
        assert len(ka) == sizes[a] and len(kb) == sizes[b]
        assert len(np.intersect1d(ka, kb, assume_unique=True)) == int(cand[i]['shared']), (a, b)
    for i in rng.choice(len(tasks), 24, replace=False):
        q, r = int(tasks[i]['q']), int(tasks[i]['r'])
        assert orc.lz_pair_stat(codes[offsets[q]:offsets[q + 1]], codes[offsets[r]:offsets[r + 1]]) == tuple(int(x) for x in stats[i]), (q, r)
    # the dereplication flags of large.yml:65-72 on the rows
    ani = stats['n_match'] / np.maximum(stats['aln_len'], 1); qcov = stats['aln_len'] / lens[tasks['q']]
    kept = np.count_nonzero((ani >= 0.95) & (qcov >= 0.85))
    assert 0 < kept < len(stats)


@pytest.mark.slow
def test_config3_phage_100k_full_size(tmp_path):
    """BASELINE configs[3] AT ITS SIZE on one MI355X, the set bench.py times (100 000 x 40 kb, seed 3; its sha256 is pinned in
    tests/golden/synth_sha256.json), asserted -- not only counted:
      * in memory: exactly the 450 000 family pairs (every family's 45), shared >= 20, a second pass bit-identical,
        set sizes / shared counts of 40 sampled pairs and the LZ rows of 60 sampled tasks equal to the oracle's;
      * the cold drop-in CLI at this size (two processes: prefilter as eight RANGE sub-shards under the 8 GiB workspace
        budget, align with 6 GiB index batches -- the only place those paths run at the real size): `fltr.txt` and
        `ani.tsv` byte-identical to the files written from the in-memory integers."""
    import filecmp
    import os
    import subprocess
    codes, offsets, names, _ = synth.make_workload('phage-100k', 10000)
    assert len(names) == 100000
    import json
    pinned = json.load(open(pathlib.Path(__file__).parent / 'golden' / 'synth_sha256.json'))
    assert synth.sha256(codes, offsets) == pinned['phage-100k']['sha256']
    fam = np.arange(100000) // 10
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    sizes, pairs = gs.kmer_shared(k=25, min_shared=20)
    cand = gs.filter_pairs(sizes, pairs, k=25, min_kmers=20, min_ident=0.7)
    assert len(cand) == 450000 and np.all(fam[cand['a']] == fam[cand['b']]) and np.all(cand['a'] > cand['b']) and np.all(cand['shared'] >= 20)
    assert len(np.unique(cand['a'].astype(np.int64) * 100000 + cand['b'])) == 450000
    tasks = gs.align_tasks(cand)
    stats = gs.lz_align(tasks)
    lens = gs.lengths()
    assert len(stats) == 900000 and np.all(stats['n_match'] <= stats['aln_len']) and np.all(stats['aln_len'] <= lens[tasks['q']]) and np.all(stats['n_regions'] > 0)
    # determinism
    sizes2, pairs2 = gs.kmer_shared(k=25, min_shared=20)
    o1 = np.lexsort((pairs['b'], pairs['a'])); o2 = np.lexsort((pairs2['b'], pairs2['a']))
    assert np.array_equal(sizes, sizes2) and np.array_equal(pairs[o1], pairs2[o2])
    assert np.array_equal(stats, gs.lz_align(tasks))
    # oracle sample
    rng = np.random.default_rng(33)
    for i in rng.choice(len(cand), 40, replace=False):
        a, b = int(cand[i]['a']), int(cand[i]['b'])
        ka = orc.kmer_set(codes[offsets[a]:offsets[a + 1]], 25); kb = orc.kmer_set(codes[offsets[b]:offsets[b + 1]], 25)
        assert len(ka) == sizes[a] and len(kb) == sizes[b]
        assert len(np.intersect1d(ka, kb, assume_unique=True)) == int(cand[i]['shared']), (a, b)
    for i in rng.choice(len(tasks), 60, replace=False):
        q, r = int(tasks[i]['q']), int(tasks[i]['r'])
        assert orc.lz_pair_stat(codes[offsets[q]:offsets[q + 1]], codes[offsets[r]:offsets[r + 1]]) == tuple(int(x) for x in stats[i]), (q, r)
    # files from the in-memory integers
    f_mem, a_mem = tmp_path / 'mem.fltr.txt', tmp_path / 'mem.ani.tsv'
    gs.write_fltr(f_mem, sizes, pairs, k=25, min_kmers=20, min_ident=0.7)
    gs.write_ani(a_mem, tasks, stats)
    del gs
    api.release_device_memory()
    # the cold CLI: two processes, FASTA on disk
    fa = tmp_path / 'phage100k.fna'
    synth.write_fasta(fa, codes, offsets, names)
    f_cli, a_cli = tmp_path / 'cli.fltr.txt', tmp_path / 'cli.ani.tsv'
    vclust = pathlib.Path(__file__).resolve().parent.parent / 'vclust.py'
    env = dict(os.environ, VG_HOST_TRACE='1')
    p = subprocess.run([sys.executable, str(vclust), 'prefilter', '-i', str(fa), '-o', str(f_cli), '-v', '0'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stderr.count('spgemm done') >= 8, 'the cold prefilter was expected to run as >= 8 sub-shard passes'
    p = subprocess.run([sys.executable, str(vclust), 'align', '-i', str(fa), '-o', str(a_cli), '--filter', str(f_cli), '-v', '0'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert filecmp.cmp(f_mem, f_cli, shallow=False)
    assert filecmp.cmp(a_mem, a_cli, shallow=False)
    assert sum(1 for _ in open(a_cli)) == 900001


@pytest.mark.parametrize('k,fraction,sub', [(25, 0.5, 5), (21, 0.3, 7), (30, 0.9, 3), (25, 0.5, 32)])
def test_hash_subshards_from_one_scan(k, fraction, sub):
    """HASH sub-shards (what sets beyond 2^32 bases run as; forced here through a fraction, which always cuts by hash): the
    kept masks of ALL passes come from one scan of the bases (k_multi_mask) and every pass computes only the k-mers of its
    kept positions.  Sizes and counts equal the oracle's, alone and under an outer shard of three."""
    from vclust_amd import _lib
    codes, offsets, names = synth.make_families(60, 5, length=9000, seed=37)
    gs = api.GenomeSet.from_codes(codes, offsets, names)
    osizes, opairs = orc.shared_all(codes, offsets, k=k, fraction=fraction)
    _lib.load().vg_set_subshards(sub)
    try:
        api.profile_enable(True); api.profile_reset()
        sizes, pairs = gs.kmer_shared(k=k, fraction=fraction)
        scopes = {e['name']: e for e in api.profile_get()}
        api.profile_enable(False)
        assert scopes['kmer_multi_mask']['launches'] == 1 and 'kmer_count' not in scopes
        assert list(sizes) == list(osizes)
        assert {(int(p['a']), int(p['b'])): int(p['shared']) for p in pairs} == opairs
        tot = np.zeros(len(gs), dtype=np.int64); acc = {}
        for sh in range(3):
            sz, pr = gs.kmer_shared(k=k, fraction=fraction, shard=sh, n_shards=3)
            tot += sz
            for p in pr:
                acc[(int(p['a']), int(p['b']))] = acc.get((int(p['a']), int(p['b'])), 0) + int(p['shared'])
        assert list(tot) == list(osizes) and acc == opairs
    finally:
        api.profile_enable(False); _lib.load().vg_set_subshards(0)

"""The CPU oracle against the reference's golden vectors (tests/golden/example/output/*).

Pins the oracle: prefilter arithmetic and file layout exactly; number formatter
exactly (792/792 fields of ani.tsv, 5693/5693 pident values); LZ parse region by region with
regression floors for the fitted rules (the upstream source is absent).
"""
import collections
import filecmp

import pathlib

import numpy as np
import pytest

import oracle_lib as orc

KMER_COUNTS = {  # distinct canonical 25-mers, SURVEY §8c
    'NC_010807': 38557, 'NC_010807.alt1': 38607, 'NC_010807.alt2': 38908, 'NC_010807.alt3': 39682,
    'NC_005091': 57222, 'NC_005091.alt1': 57392, 'NC_005091.alt2': 58459, 'NC_025457': 42629,
    'NC_025457.alt1': 39537, 'NC_025457.alt2': 61292, 'NC_002486': 45598, 'NC_002486.alt': 45598,
}


@pytest.fixture(scope='module')
def example(golden_dir):
    return orc.read_fasta_codes(golden_dir / 'multifasta.fna')


def read_tsv(path):
    with open(path) as fh:
        return [line.rstrip('\n').split('\t') for line in fh]


def test_kmer_set_sizes_and_shared_counts(example):
    codes, offsets, names = example
    sizes, pairs = orc.shared_all(codes, offsets, k=25)
    assert dict(zip(names, sizes)) == KMER_COUNTS
    idx = {n: i for i, n in enumerate(names)}
    assert pairs[(idx['NC_010807.alt1'], idx['NC_010807'])] == 35785
    assert pairs[(idx['NC_025457.alt2'], idx['NC_025457.alt1'])] == 5766
    assert pairs[(idx['NC_002486.alt'], idx['NC_002486'])] == 45550
    assert len(pairs) == 13                       # the 53 other pairs share no 25-mer at all


@pytest.mark.parametrize('inp', ['multifasta.fna', 'multifasta.fna.gz'])
def test_prefilter_file_is_golden(tmp_path, golden_dir, inp):
    out = tmp_path / 'fltr.txt'
    orc.run_cli('prefilter', '-o', out, golden_dir / inp)
    assert filecmp.cmp(out, golden_dir / 'output' / 'fltr.txt', shallow=False)


def test_prefilter_directory_mode_values(tmp_path, golden_dir):
    """test.py:336-385: directory input gives the same 26 symmetric entries (names may keep .fna)."""
    out = tmp_path / 'fltr.txt'
    orc.run_cli('prefilter', '-o', out, *sorted((golden_dir / 'fna').iterdir()))
    def values(p):
        return sorted(v.split(':')[1] for line in open(p).read().splitlines()[1:] for v in line.split(',')[1:] if ':' in v)
    assert values(out) == values(golden_dir / 'output' / 'fltr.txt')


def test_number_format_reproduces_ani_tsv(golden_dir):
    """L6 + L7: metrics from the golden integer sums, printed with the oracle's formatter."""
    aln = read_tsv(golden_dir / 'output' / 'ani.aln.tsv')[1:]
    ids = read_tsv(golden_dir / 'output' / 'ani.ids.tsv')[1:]
    length = {r[0]: int(r[1]) for r in ids}
    M = collections.Counter(); A = collections.Counter(); N = collections.Counter()
    for r in aln:
        key = (r[0], r[1]); M[key] += int(r[8]); A[key] += int(r[3]); N[key] += 1
    rows = read_tsv(golden_dir / 'output' / 'ani.tsv')
    cols = rows[0]
    ok = total = 0
    for r in rows[1:]:
        q, ref = r[2], r[3]
        lq, lr = length[q], length[ref]
        want = {
            'tani': (M[(q, ref)] + M[(ref, q)]) / (lq + lr), 'gani': M[(q, ref)] / lq, 'ani': M[(q, ref)] / A[(q, ref)],
            'qcov': A[(q, ref)] / lq, 'rcov': A[(ref, q)] / lr,
        }
        for name, val in want.items():
            total += 1
            ok += orc.fmt_num(val) == r[cols.index(name)]
        assert int(r[cols.index('num_alns')]) == N[(q, ref)]
        total += 1
        lo, hi = min(lq, lr), max(lq, lr)
        ok += ('1' if lo == hi else f'{lo / hi:.4f}') == r[cols.index('len_ratio')]
    assert (ok, total) == (792, 792)
    pid_ok = sum(orc.fmt_num(100.0 * int(r[8]) / int(r[3])) == r[2] for r in aln)
    assert pid_ok == len(aln) == 5693


@pytest.fixture(scope='module')
def oracle_align(tmp_path_factory, golden_dir):
    d = tmp_path_factory.mktemp('oalign')
    orc.run_cli('align', '-o', d / 'ani.tsv', '--out-aln', d / 'ani.aln.tsv', golden_dir / 'multifasta.fna')
    return d


def test_ids_file_is_golden(oracle_align, golden_dir):
    assert filecmp.cmp(oracle_align / 'ani.ids.tsv', golden_dir / 'output' / 'ani.ids.tsv', shallow=False)


def test_row_order_and_layout(oracle_align, golden_dir):
    g = read_tsv(golden_dir / 'output' / 'ani.tsv'); m = read_tsv(oracle_align / 'ani.tsv')
    assert g[0] == m[0] and len(g) == len(m) == 133
    assert [r[:4] for r in g] == [r[:4] for r in m]
    assert [r[10] for r in g] == [r[10] for r in m]          # len_ratio


def _load_regions(p):
    d = collections.defaultdict(list)
    for r in read_tsv(p)[1:]:
        d[(r[0], r[1])].append(tuple(int(x) for x in r[3:10]) + (r[2],))
    return d


def test_lz_parse_identical_to_golden_regions(oracle_align, golden_dir):
    """example/output/ani.aln.tsv:1-5694: STRICT identity -- the multiset of regions (all seven integers and
    pident) per ordered pair equals the reference's: 5 693 of 5 693, nothing missing, nothing surplus."""
    g = _load_regions(golden_dir / 'output' / 'ani.aln.tsv'); m = _load_regions(oracle_align / 'ani.aln.tsv')
    assert sum(len(v) for v in g.values()) == 5693
    assert sum(len(v) for v in m.values()) == 5693
    assert set(g) == set(m)
    for k in g:
        assert sorted(g[k]) == sorted(m[k]), k


def test_ani_rows_identical_to_golden(oracle_align, golden_dir):
    """example/output/ani.tsv:1-133: the file is byte-identical to the reference's (132 rows + header)."""
    assert (golden_dir / 'output' / 'ani.tsv').read_bytes() == (oracle_align / 'ani.tsv').read_bytes()


def test_tani_within_reference_test_tolerance(oracle_align):
    """test.py:456-477: tANI of the 8 simulated pairs within 0.007 of the simulated truth."""
    truth = {('NC_010807', 'NC_010807.alt1'): 0.99753, ('NC_010807', 'NC_010807.alt2'): 0.98985,
             ('NC_010807', 'NC_010807.alt3'): 0.98384, ('NC_005091', 'NC_005091.alt1'): 0.97161,
             ('NC_005091', 'NC_005091.alt2'): 0.96707, ('NC_025457', 'NC_025457.alt1'): 0.80607,
             ('NC_025457', 'NC_025457.alt2'): 0.75921, ('NC_002486', 'NC_002486.alt'): 1.00000}
    rows = read_tsv(oracle_align / 'ani.tsv')[1:]
    tani = {(r[2], r[3]): float(r[4]) for r in rows}
    for pair, t in truth.items():
        assert abs(tani[pair] - t) < 0.007


def test_tani_close_to_golden(oracle_align, golden_dir):
    g = read_tsv(golden_dir / 'output' / 'ani.tsv')[1:]; m = read_tsv(oracle_align / 'ani.tsv')[1:]
    worst = max(abs(float(a[4]) - float(b[4])) for a, b in zip(g, m))
    assert worst == 0.0, worst


def test_identical_and_shuffled_genome(example):
    codes, offsets, names = example
    i, j = names.index('NC_002486'), names.index('NC_002486.alt')
    a = codes[offsets[i]:offsets[i + 1]]; b = codes[offsets[j]:offsets[j + 1]]
    assert orc.lz_pair_stat(a, a) == (len(a), len(a), 1)
    assert orc.lz_pair_stat(b, a) == (45636, 45636, 3)       # example/output/ani.tsv:78-79


def test_edge_inputs():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 4, size=500, dtype=np.uint8)
    assert orc.lz_pair_stat(a[:5], a) == (0, 0, 0)           # shorter than the anchor length
    assert orc.lz_pair_stat(np.full(100, 4, np.uint8), a) == (0, 0, 0)   # all N
    assert len(orc.kmer_set(a[:10], 25)) == 0
    assert len(orc.kmer_set(np.full(100, 4, np.uint8), 25)) == 0


def test_oracle_under_sanitizers(tmp_path, golden_dir):
    """The checker itself is checked: oracle_cli built with -fsanitize=address,undefined (oracle/Makefile `asan`)
    runs prefilter + align (with the alignment table) on the example and on a ragged / N-rich input without a
    report, and writes the same files as the plain build."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    r = subprocess.run(['make', '-C', str(orc.ORACLE_DIR), 'asan'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    asan = orc.ORACLE_DIR / '_build' / 'oracle_cli_asan'
    env = dict(**__import__('os').environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1', OMP_NUM_THREADS='2')
    ragged = tmp_path / 'ragged.fna'
    ragged.write_text('>a\nACGTNNNNACGTACGTACGTTTGACCAGTAGGCATGCATGCATCGATCGATTTAGC\n>b desc\nACG\n>c\n\n>d\n' + 'ACGTTGCA' * 40 + '\n')
    for fasta, name in ((golden_dir / 'multifasta.fna', 'ex'), (ragged, 'rg')):
        for cli, tag in ((asan, 'asan'), (orc.CLI, 'plain')):
            p = subprocess.run([str(cli), 'prefilter', '-o', str(tmp_path / f'{name}_{tag}.fltr'), str(fasta)], env=env, stderr=subprocess.PIPE, text=True)
            assert p.returncode == 0 and 'runtime error' not in p.stderr and 'AddressSanitizer' not in p.stderr, p.stderr[-2000:]
            p = subprocess.run([str(cli), 'align', '-o', str(tmp_path / f'{name}_{tag}.tsv'), '--out-aln', str(tmp_path / f'{name}_{tag}.aln'), str(fasta)],
                               env=env, stderr=subprocess.PIPE, text=True)
            assert p.returncode == 0 and 'runtime error' not in p.stderr and 'AddressSanitizer' not in p.stderr, p.stderr[-2000:]
        for ext in ('fltr', 'tsv', 'aln'):
            assert filecmp.cmp(tmp_path / f'{name}_asan.{ext}', tmp_path / f'{name}_plain.{ext}', shallow=False)


def single_linkage_partition(ani_tsv, ids_tsv, metric='tani', threshold=0.95):
    """Test-side restatement of what `vclust cluster --metric tani --tani 0.95` (single linkage, the default
    algorithm, vclust.py:466-480) does with the two files of the align stage: objects are the
    rows of the ids file, two objects are linked when a row of ani.tsv has metric >= threshold."""
    ids = [r[0] for r in read_tsv(ids_tsv)[1:]]
    parent = {x: x for x in ids}

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    rows = read_tsv(ani_tsv)
    col = rows[0].index(metric); qc = rows[0].index('query'); rc = rows[0].index('reference')
    for r in rows[1:]:
        if float(r[col]) >= threshold:
            parent[find(r[qc])] = find(r[rc])
    groups = collections.defaultdict(set)
    for x in ids:
        groups[find(x)].add(x)
    return sorted(sorted(g) for g in groups.values())


def golden_partition(golden_dir):
    groups = collections.defaultdict(set)
    for obj, cl in read_tsv(golden_dir / 'output' / 'clusters.tsv')[1:]:
        groups[cl].add(obj)
    return sorted(sorted(g) for g in groups.values())


def test_cluster_handoff_reproduces_golden_clusters(oracle_align, golden_dir):
    """The step after the path: ani.tsv + ani.ids.tsv are Clusty's input.  bin/clusty is not part
    of this repository (CPU tool, out of scope), so the contract is checked on its golden OUTPUT: single
    linkage at tANI >= 0.95 (the setting that reproduces the golden file from the reference's own ani.tsv) over our files gives the partition of example/output/clusters.tsv -- first over the
    reference's own ani.tsv (the restatement of the clustering is right), then over ours."""
    want = golden_partition(golden_dir)
    assert single_linkage_partition(golden_dir / 'output' / 'ani.tsv', golden_dir / 'output' / 'ani.ids.tsv') == want
    assert single_linkage_partition(oracle_align / 'ani.tsv', oracle_align / 'ani.ids.tsv') == want


@pytest.mark.parametrize('threads', [1, 3, 8])
def test_multithreaded_prefilter_equals_the_serial_checker(example, threads):
    """bench.py's cpu_baseline runs the oracle with EVERY stage on all host threads (vo_shared_all_mt: hash partitions of
    the (k-mer, genome) records sorted per thread, per-thread pair tables merged by pair hash).  The serial vo_shared_all
    stays the checker: same set sizes, same pairs, same counts -- on the reference's example, on families with many pairs
    per genome, and on edge sets (one genome, genomes shorter than k, fractions)."""
    import sys
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
    from vclust_amd import synth
    codes, offsets, names = example
    cases = [(codes, offsets, 25, 1.0), (codes, offsets, 15, 0.3)]
    fc, fo, _ = synth.make_families(12, 9, length=6000, seed=17)
    cases += [(fc, fo, 25, 1.0), (fc, fo, 21, 0.5), (fc[:fo[1]], fo[:2], 25, 1.0)]
    rng = np.random.default_rng(2)
    tiny = rng.integers(0, 4, 200).astype(np.uint8)
    cases.append((tiny, np.array([0, 10, 20, 120, 200], dtype=np.int64), 25, 1.0))
    for c, o, k, f in cases:
        s0, p0 = orc.shared_all(c, o, k=k, fraction=f)
        s1, p1, stage_s, thr = orc.shared_all_mt(c, o, k=k, fraction=f, threads=threads)
        assert thr == threads and len(stage_s) == 3
        assert list(s0) == list(s1) and p0 == p1


def test_multithreaded_path_rows_equal_the_serial_rows():
    import sys
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
    from vclust_amd import synth
    codes, offsets, _ = synth.make_families(6, 5, length=5000, seed=4)
    r0 = orc.path_rows(codes, offsets)
    r1, stage_s, thr = orc.path_rows_mt(codes, offsets, threads=4)
    assert thr == 4 and set(stage_s) == {'sets', 'index', 'pair_count', 'lz'}
    key = lambda r: sorted(map(tuple, r.tolist()))
    assert len(r0) > 50 and key(r0) == key(r1)

"""CPU-side tests of libvclust_gpu.so: the C-ABI loads and exports every declared symbol, the
host-only entry points (ingest, writers, task lists) reproduce the reference's files from the
golden integers, and compute calls fail loudly without a GPU (no CPU fallback)."""
import collections
import ctypes as C
import filecmp
import re

import numpy as np
import pytest

import oracle_lib as orc
from vclust_amd import _lib, api


def test_header_symbols_all_exported_and_bound():
    header = (_lib.PKG_DIR.parent / 'include' / 'vclust_gpu.h').read_text()
    declared = set(re.findall(r'\b(vg_[a-z_0-9]+)\s*\(', header))
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in vclust_gpu.h but not exported'
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)


def test_no_gpu_means_loud_failure(golden_dir):
    if api.device_count() > 0:
        pytest.skip('a HIP device is visible')
    gs = api.GenomeSet.load([golden_dir / 'multifasta.fna'], multisample=True)
    with pytest.raises(_lib.VclustGpuError) as e:
        gs.kmer_shared(k=25)
    assert e.value.code == -3 and 'no CPU fallback' in str(e.value)
    with pytest.raises(_lib.VclustGpuError):
        gs.lz_align(np.zeros(2, dtype=api.TASK_DTYPE))
    with pytest.raises(_lib.VclustGpuError):
        api.set_device(0)


@pytest.mark.parametrize('inp,multi', [('multifasta.fna', True), ('multifasta.fna.gz', True), ('fna', False)])
def test_ingest(golden_dir, inp, multi):
    codes, offsets, names = orc.read_fasta_codes(golden_dir / 'multifasta.fna')
    paths = [golden_dir / inp] if multi else sorted((golden_dir / 'fna').iterdir())
    gs = api.GenomeSet.load(paths, multisample=multi, n_threads=4)
    assert len(gs) == 12
    if multi:
        assert gs.names() == names
        assert list(gs.lengths()) == list(np.diff(offsets))
    else:
        assert sorted(n.replace('.fna', '') for n in gs.names()) == sorted(names)
        assert sorted(gs.lengths()) == sorted(np.diff(offsets))
    assert gs.total_len == int(offsets[-1])


def test_ingest_errors(tmp_path):
    with pytest.raises(_lib.VclustGpuError) as e:
        api.GenomeSet.load([tmp_path / 'missing.fna'], multisample=True)
    assert e.value.code == -2


def test_write_fltr_from_oracle_counts_is_golden(tmp_path, golden_dir):
    """K3 + K4 of the product's host writer; accepts concatenated per-shard partial counts."""
    codes, offsets, names = orc.read_fasta_codes(golden_dir / 'multifasta.fna')
    sizes, pairs = orc.shared_all(codes, offsets, k=25)
    gs = api.GenomeSet.load([golden_dir / 'multifasta.fna'], multisample=True)
    arr = np.array([(a, b, s) for (a, b), s in pairs.items()], dtype=api.PAIR_DTYPE)
    gs.write_fltr(tmp_path / 'f.txt', sizes, arr)
    assert filecmp.cmp(tmp_path / 'f.txt', golden_dir / 'output' / 'fltr.txt', shallow=False)
    halves = np.concatenate([arr, arr]); halves['shared'] = np.concatenate([arr['shared'] // 2, arr['shared'] - arr['shared'] // 2])
    gs.write_fltr(tmp_path / 'g.txt', sizes, halves[::-1])
    assert filecmp.cmp(tmp_path / 'g.txt', golden_dir / 'output' / 'fltr.txt', shallow=False)


def test_fltr_thresholds_and_max_seqs(tmp_path, golden_dir):
    codes, offsets, names = orc.read_fasta_codes(golden_dir / 'multifasta.fna')
    sizes, pairs = orc.shared_all(codes, offsets, k=25)
    gs = api.GenomeSet.load([golden_dir / 'multifasta.fna'], multisample=True)
    arr = np.array([(a, b, s) for (a, b), s in pairs.items()], dtype=api.PAIR_DTYPE)
    gs.write_fltr(tmp_path / 'a.txt', sizes, arr, min_ident=0.99)
    orc.run_cli('prefilter', '--min-ident', '0.99', '-o', tmp_path / 'b.txt', golden_dir / 'multifasta.fna')
    assert filecmp.cmp(tmp_path / 'a.txt', tmp_path / 'b.txt', shallow=False)
    gs.write_fltr(tmp_path / 'c.txt', sizes, arr, max_seqs=1)
    orc.run_cli('prefilter', '--max-seqs', '1', '-o', tmp_path / 'd.txt', golden_dir / 'multifasta.fna')
    assert filecmp.cmp(tmp_path / 'c.txt', tmp_path / 'd.txt', shallow=False)
    assert sum(line.count(':') for line in open(tmp_path / 'c.txt')) - 1 <= 12


def test_read_filter_and_tasks(golden_dir):
    gs = api.GenomeSet.load([golden_dir / 'multifasta.fna'], multisample=True)
    allp = gs.read_filter(None)
    assert len(allp) == 66 and all(allp['a'] > allp['b'])
    flt = gs.read_filter(golden_dir / 'output' / 'fltr.txt', 0.0)
    assert len(flt) == 13
    assert len(gs.read_filter(golden_dir / 'output' / 'fltr.txt', 0.99)) == 8
    order = gs.align_order()
    ids = [l.split('\t')[0] for l in open(golden_dir / 'output' / 'ani.ids.tsv').read().splitlines()[1:]]
    assert [gs.names()[i] for i in order] == ids
    tasks = gs.align_tasks(allp)
    rank = {int(g): r for r, g in enumerate(order)}
    rows = [l.split('\t')[:2] for l in open(golden_dir / 'output' / 'ani.tsv').read().splitlines()[1:]]
    assert [[str(rank[int(t['q'])]), str(rank[int(t['r'])])] for t in tasks] == rows


@pytest.mark.parametrize('fmt', ['standard', 'lite', 'complete'])
def test_write_ani_from_golden_integers(tmp_path, golden_dir, fmt):
    """L6-L8 of the product's host writer fed with the integers summed from the golden
    alignment table: ani.tsv (all 792 numeric fields), ids file and the alignment table."""
    gs = api.GenomeSet.load([golden_dir / 'multifasta.fna'], multisample=True)
    names = gs.names(); idx = {n: i for i, n in enumerate(names)}
    tasks = gs.align_tasks(gs.read_filter(None))
    tpos = {(int(t['q']), int(t['r'])): i for i, t in enumerate(tasks)}
    stats = np.zeros(len(tasks), dtype=api.STAT_DTYPE)
    regions = []
    lens = gs.lengths()
    for line in open(golden_dir / 'output' / 'ani.aln.tsv').read().splitlines()[1:]:
        c = line.split('\t')
        t = tpos[(idx[c[0]], idx[c[1]])]
        stats[t]['n_match'] += int(c[8]); stats[t]['aln_len'] += int(c[3]); stats[t]['n_regions'] += 1
        L = int(lens[idx[c[1]]]); rs, re_ = int(c[6]), int(c[7])
        to_rr = (lambda p: p - 1) if rs <= re_ else (lambda p: L + 1 + (L - p))
        regions.append((t, int(c[4]) - 1, int(c[5]) - 1, to_rr(rs), to_rr(re_), int(c[8])))
    regions = np.array(regions, dtype=api.REGION_DTYPE)
    from vclust_amd.cli import ALIGN_OUTFMT
    out = tmp_path / 'ani.tsv'; aln = tmp_path / 'ani.aln.tsv'
    gs.write_ani(out, tasks, stats, regions=regions, columns=ALIGN_OUTFMT[fmt], out_aln=aln)
    assert filecmp.cmp(tmp_path / 'ani.ids.tsv', golden_dir / 'output' / 'ani.ids.tsv', shallow=False)
    if fmt == 'standard':
        assert filecmp.cmp(out, golden_dir / 'output' / 'ani.tsv', shallow=False)
        assert sorted(open(aln).read().splitlines()) == sorted(open(golden_dir / 'output' / 'ani.aln.tsv').read().splitlines())
    else:
        assert open(out).readline().split() == ALIGN_OUTFMT[fmt]
        assert sum(1 for _ in open(out)) == 133


def test_write_ani_output_filters(tmp_path, golden_dir):
    gs = api.GenomeSet.load([golden_dir / 'multifasta.fna'], multisample=True)
    tasks = gs.align_tasks(gs.read_filter(None))
    stats = np.zeros(len(tasks), dtype=api.STAT_DTYPE)
    stats['n_match'] = 30000; stats['aln_len'] = 35000; stats['n_regions'] = 5
    stats[0] = (100, 200, 1)
    gs.write_ani(tmp_path / 'o.tsv', tasks, stats, out_filters={'ani': 0.8})
    assert sum(1 for _ in open(tmp_path / 'o.tsv')) == 132        # header + 131 rows, row 0 (ani 0.5) dropped


def test_number_format_fast_path_equals_oracle(tmp_path, golden_dir):
    """The product's fast formatter against the oracle's exact one on 40 000 random quotients
    (plus the awkward cases: short decimals, exact ties, powers of ten)."""
    rng = np.random.default_rng(12)
    gs = api.GenomeSet.load([golden_dir / 'multifasta.fna'], multisample=True)
    tasks = np.tile(gs.align_tasks(gs.read_filter(None)), 160)[:20000]
    stats = np.zeros(len(tasks), dtype=api.STAT_DTYPE)
    stats['aln_len'] = rng.integers(1, 60000, size=len(tasks))
    stats['n_match'] = (stats['aln_len'] * rng.random(len(tasks))).astype(np.uint32)
    special = [(1, 2), (5, 8), (73, 128), (125, 128), (117, 128), (1, 1), (0, 7), (1, 3), (999999, 1000000), (1, 1000000),
               (9999995, 10000000), (64, 64000), (3, 64164), (12345, 100000)]
    for i, (m, a) in enumerate(special):
        stats[2 * i] = (m, a, 1); stats[2 * i + 1] = (m, a, 1)
    gs.write_ani(tmp_path / 'o.tsv', tasks, stats, columns=['tani', 'gani', 'ani', 'qcov', 'rcov'])
    lens = gs.lengths()
    rows = [l.split('\t') for l in open(tmp_path / 'o.tsv').read().splitlines()[1:]]
    assert len(rows) == len(tasks)
    bad = 0
    for t, (tk, s, row) in enumerate(zip(tasks, stats, rows)):
        rev = stats[t ^ 1]
        lq, lr = int(lens[tk['q']]), int(lens[tk['r']])
        want = [(int(s['n_match']) + int(rev['n_match'])) / (lq + lr), int(s['n_match']) / lq,
                int(s['n_match']) / int(s['aln_len']), int(s['aln_len']) / lq, int(rev['aln_len']) / lr]
        bad += [orc.fmt_num(v) for v in want] != row
    assert bad == 0


MESSY = (b">g1 first genome\tdescription\r\nACGTacgtNNnnRYKM\r\n\r\nacgtACGT\r\n"
         b">g2\nAC\nGT\n\n>empty_record\n>g4|pipes and spaces  \nTTTTGGGGCCCCAAAA-*.\nacgu")


def test_ingest_messy_fasta(tmp_path):
    """Lower case, IUPAC and other symbols (all non-ACGT -> N), CRLF, blank lines, an empty record, no
    final newline: names and lengths equal the oracle reader's (GPU test compares the bases too)."""
    import gzip
    p = tmp_path / 'messy.fna'; p.write_bytes(MESSY)
    pz = tmp_path / 'messy.fna.gz'; pz.write_bytes(gzip.compress(MESSY))
    codes, offsets, names = orc.read_fasta_codes(p)
    for path in (p, pz):
        gs = api.GenomeSet.load([path], multisample=True, n_threads=3)
        assert gs.names() == names
        assert list(gs.lengths()) == list(np.diff(offsets))
        for i in range(len(names)):
            assert np.array_equal(gs.codes(i), np.minimum(codes[offsets[i]:offsets[i + 1]], 4)), i


def test_ingest_random_fasta_equals_oracle_reader(tmp_path):
    """The packer (sixteen symbols per trip through a shift register, symbol-by-symbol fall-back) against the
    oracle's reader on random FASTA: line widths 1..200, CRLF or LF, lower case, N runs, IUPAC codes, blanks and tabs
    inside lines, empty lines, records of 0..5 000 symbols; multi-record (one genome per record) and directory mode
    (records of a file joined by one N)."""
    rng = np.random.default_rng(8)
    alphabet = np.frombuffer(b'ACGTacgtNnRYKMSWBDHV-*', dtype=np.uint8)
    weights = np.array([20, 20, 20, 20, 5, 5, 5, 5, 2, 1] + [0.25] * 12); weights = weights / weights.sum()
    files = []
    for fi in range(4):
        out = bytearray()
        for r in range(int(rng.integers(1, 40))):
            n = int(rng.integers(0, 5000)) if rng.random() > 0.1 else 0
            seq = rng.choice(alphabet, size=n, p=weights).tobytes()
            width = int(rng.integers(1, 200)); eol = b'\r\n' if rng.random() < 0.3 else b'\n'
            out += b'>f%d_r%d some text' % (fi, r) + eol
            for o in range(0, n, width):
                line = seq[o:o + width]
                if rng.random() < 0.05 and len(line) > 2:
                    k = int(rng.integers(1, len(line))); line = line[:k] + (b' ' if rng.random() < 0.5 else b'\t') + line[k:]
                out += line + eol
                if rng.random() < 0.03:
                    out += eol
        p = tmp_path / ('f%d.fna' % fi); p.write_bytes(bytes(out)); files.append(p)
    for path in files:
        codes, offsets, names = orc.read_fasta_codes(path)
        gs = api.GenomeSet.load([path], multisample=True, n_threads=4)
        assert gs.names() == names and list(gs.lengths()) == list(np.diff(offsets))
        for i in range(len(names)):
            assert np.array_equal(gs.codes(i), np.minimum(codes[offsets[i]:offsets[i + 1]], 4)), (path.name, i)
    gs = api.GenomeSet.load(files, multisample=False, n_threads=4)
    for i, path in enumerate(files):
        codes, offsets, names = orc.read_fasta_codes(path, multisample=False)
        assert int(gs.lengths()[i]) == len(codes) and np.array_equal(gs.codes(i), np.minimum(codes, 4)), path.name


def test_read_filter_with_genomes_in_another_order(tmp_path, golden_dir):
    """The filter names genomes, not positions: read against a set whose records come in another order (reversed, and
    with one genome the filter does not know), the same couples of NAMES come back; and a large filter (counting-sort
    path of the reader) equals the small-filter path on the same couples."""
    text = (golden_dir / 'multifasta.fna').read_text()
    recs = ['>' + r for r in text.split('>') if r]
    (tmp_path / 'rev.fna').write_text(''.join(reversed(recs)) + '>stranger\nACGTACGTACGTACGTACGTAGCTAGCTAGCATCGATCGATGCATGCAT\n')
    gs = api.GenomeSet.load([golden_dir / 'multifasta.fna'], multisample=True)
    rv = api.GenomeSet.load([tmp_path / 'rev.fna'], multisample=True)
    def couples(g, arr):
        nm = g.names()
        return sorted(tuple(sorted((nm[int(e['a'])], nm[int(e['b'])]))) for e in arr)
    for thr in (0.0, 0.99):
        a = gs.read_filter(golden_dir / 'output' / 'fltr.txt', thr)
        b = rv.read_filter(golden_dir / 'output' / 'fltr.txt', thr)
        assert couples(gs, a) == couples(rv, b) and len(a) in (13, 8)
        assert all(b['a'] > b['b']) and list(map(tuple, b[['a', 'b']])) == sorted(map(tuple, b[['a', 'b']]))
    # a dense filter over 60 genomes: 1 770 couples > 4 x 60 -> the counting-sort path; rows and columns shuffled
    rng = np.random.default_rng(3)
    n = 60
    names = ['g%03d' % i for i in range(n)]
    (tmp_path / 'many.fna').write_text(''.join('>%s\n%s\n' % (nm, 'ACGT' * 10) for nm in names))
    order = rng.permutation(n)
    lines = ['kmer-length: 25 fraction: 1,' + ','.join(names[i] for i in order) + ',']
    want = set()
    for r in rng.permutation(n):
        cells = []
        for ci, c in enumerate(order):
            if names[c] < names[r]:
                cells.append('%d:%.6f' % (ci + 1, 0.5 + 0.4 * rng.random())); want.add((max(r, c), min(r, c)))
        lines.append(names[r] + ',' + ','.join(cells))
    (tmp_path / 'many.txt').write_text('\n'.join(lines) + '\n')
    many = api.GenomeSet.load([tmp_path / 'many.fna'], multisample=True)
    got = many.read_filter(tmp_path / 'many.txt', 0.0)
    assert [(int(e['a']), int(e['b'])) for e in got] == sorted(want) and len(got) == n * (n - 1) // 2


def test_align_tasks_large_lists_take_the_parallel_path():
    """vg_align_tasks on 2.4 M couples (range partition over the threads) and on 50 000 (one thread) against the same
    restatement: couples sorted on the length ranks (lo, hi), two rows per couple, longer genome first as the reference."""
    rng = np.random.default_rng(5)
    n = 60000
    lens = rng.integers(50, 400, size=n)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    gs = api.GenomeSet.from_codes(np.zeros(int(lens.sum()), dtype=np.uint8), offsets, ['g%d' % i for i in range(n)])
    order = np.asarray(gs.align_order()); rank = np.empty(n, dtype=np.int64); rank[order] = np.arange(n)
    for m in (2400000, 50000):
        a = rng.integers(1, n, size=m).astype(np.uint32); b = (rng.random(m) * a).astype(np.uint32)
        pairs = np.zeros(m, dtype=api.PAIR_DTYPE); pairs['a'] = a; pairs['b'] = b
        pairs = np.unique(pairs)
        got = gs.align_tasks(pairs)
        x = rank[pairs['a']]; y = rank[pairs['b']]; lo = np.minimum(x, y); hi = np.maximum(x, y)
        o = np.lexsort((hi, lo)); lo = lo[o]; hi = hi[o]
        exp = np.zeros(2 * len(lo), dtype=api.TASK_DTYPE)
        exp['q'][0::2] = order[hi]; exp['r'][0::2] = order[lo]; exp['q'][1::2] = order[lo]; exp['r'][1::2] = order[hi]
        assert np.array_equal(got, exp), m
    bad = np.zeros(2200000, dtype=api.PAIR_DTYPE); bad['a'] = n + 5
    with pytest.raises(_lib.VclustGpuError):
        gs.align_tasks(bad)


def _bgzf(data: bytes, block: int = 65280) -> bytes:
    """bgzip's container: gzip members of <= 64 KiB with their compressed size in a 'BC' extra subfield, and the
    empty end-of-file member."""
    import struct
    import zlib
    out = bytearray()
    for o in list(range(0, len(data), block)) + [len(data)]:
        chunk = data[o:o + block] if o < len(data) else b''
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        cdata = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(cdata) + 8
        out += struct.pack('<BBBBIBBH', 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b'BC' + struct.pack('<HH', 2, bsize - 1)
        out += cdata + struct.pack('<II', zlib.crc32(chunk) & 0xffffffff, len(chunk))
    return bytes(out)


def test_ingest_bgzf_blocks_in_parallel(tmp_path, golden_dir):
    """A bgzip-compressed FASTA (block-parallel inflate) gives the genomes of the plain file and of the same text as
    an ordinary one-member gzip; a damaged block is a read error, not a short set."""
    import gzip
    text = (golden_dir / 'multifasta.fna').read_bytes()
    bg = _bgzf(text, block=4093)                      # odd block size: records and lines straddle members
    assert gzip.decompress(bg) == text               # (a valid multi-member gzip stream for everybody else)
    (tmp_path / 'b.fna.gz').write_bytes(bg)
    (tmp_path / 'g.fna.gz').write_bytes(gzip.compress(text))
    plain = api.GenomeSet.load([golden_dir / 'multifasta.fna'], multisample=True, n_threads=4)
    for name in ('b.fna.gz', 'g.fna.gz'):
        gs = api.GenomeSet.load([tmp_path / name], multisample=True, n_threads=4)
        assert gs.names() == plain.names() and list(gs.lengths()) == list(plain.lengths()), name
        for i in range(len(plain.names())):
            assert np.array_equal(gs.codes(i), plain.codes(i)), (name, i)
    bad = bytearray(bg); bad[len(bad) // 2] ^= 0x55
    (tmp_path / 'bad.fna.gz').write_bytes(bytes(bad))
    with pytest.raises(_lib.VclustGpuError):
        api.GenomeSet.load([tmp_path / 'bad.fna.gz'], multisample=True, n_threads=4)


def test_ingest_gzip_through_the_own_decoder(tmp_path, golden_dir):
    """One-member and multi-member gzip files (levels 1/6/9, stored blocks, a header with a file name) go through the
    library's own inflate (vg_inflate.cpp) and give the genomes of the plain text; with VG_GZ=zlib (a fresh process) the
    same; a damaged stream is an error from either decoder, never a short or altered set."""
    import gzip
    import io
    import subprocess
    import sys
    import zlib
    rng = np.random.default_rng(11)
    recs = []
    for r in range(60):
        n = int(rng.integers(0, 30000))
        seq = rng.choice(np.frombuffer(b'ACGTacgtN', dtype=np.uint8), size=n, p=[.24, .24, .24, .24, .01, .01, .005, .005, .01]).tobytes()
        w = int(rng.integers(20, 120))
        recs.append(b'>r%d x\n' % r + b'\n'.join(seq[o:o + w] for o in range(0, n, w)) + b'\n')
    text = b''.join(recs)
    (tmp_path / 'p.fna').write_bytes(text)
    plain = api.GenomeSet.load([tmp_path / 'p.fna'], multisample=True, n_threads=4)
    files = {}
    for lvl in (1, 6, 9):
        files['l%d.fna.gz' % lvl] = gzip.compress(text, lvl)
    files['stored.fna.gz'] = gzip.compress(text, 0)
    cut = text.index(b'>r30 ')
    files['multi.fna.gz'] = gzip.compress(text[:cut], 6) + gzip.compress(b'', 6) + gzip.compress(text[cut:], 2)
    bio = io.BytesIO()
    with gzip.GzipFile(filename='genomes.fna', mode='wb', fileobj=bio, mtime=12345) as fh:
        fh.write(text)
    files['named.fna.gz'] = bio.getvalue()
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)
    files['fixed.fna.gz'] = co.compress(text) + co.flush()
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
        gs = api.GenomeSet.load([tmp_path / name], multisample=True, n_threads=4)
        assert gs.names() == plain.names() and list(gs.lengths()) == list(plain.lengths()), name
        for i in range(len(plain.names())):
            assert np.array_equal(gs.codes(i), plain.codes(i)), (name, i)
    code = ("import sys; sys.path.insert(0, %r); from vclust_amd import api; "
            "gs = api.GenomeSet.load([%r], multisample=True, n_threads=2); print(sum(int(x) for x in gs.lengths()), len(gs.names()))"
            % (str(__import__('pathlib').Path(__file__).resolve().parent.parent), str(tmp_path / 'l6.fna.gz')))
    import os
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, VG_DEV_SWITCHES='1', VG_GZ='zlib'), stdout=subprocess.PIPE, text=True, check=True).stdout.split()
    assert [int(out[0]), int(out[1])] == [int(sum(plain.lengths())), len(plain.names())]
    bad = bytearray(files['l6.fna.gz']); bad[len(bad) // 3] ^= 0x10
    (tmp_path / 'bad.fna.gz').write_bytes(bytes(bad))
    with pytest.raises(_lib.VclustGpuError):
        api.GenomeSet.load([tmp_path / 'bad.fna.gz'], multisample=True, n_threads=2)


def test_filter_pairs_equals_fltr_file(tmp_path, golden_dir):
    """vg_filter_pairs keeps exactly the pairs vg_write_fltr prints (golden fltr.txt: 13 entries)."""
    codes, offsets, names = orc.read_fasta_codes(golden_dir / 'multifasta.fna')
    sizes, pairs = orc.shared_all(codes, offsets, k=25)
    arr = np.array([(a, b, s) for (a, b), s in pairs.items()], dtype=api.PAIR_DTYPE)
    gs = api.GenomeSet.load([golden_dir / 'multifasta.fna'], multisample=True)
    kept = gs.filter_pairs(sizes, arr, k=25, min_kmers=20, min_ident=0.7)
    out = tmp_path / 'f.txt'
    gs.write_fltr(out, sizes, arr, k=25, min_kmers=20, min_ident=0.7)
    printed = set()
    for row, line in enumerate(open(out).read().splitlines()[1:]):
        for ent in line.split(',')[1:]:
            if ent:
                printed.add((row, int(ent.split(':')[0]) - 1))
    assert {(int(p['a']), int(p['b'])) for p in kept} == printed and len(printed) == 13
    assert len(gs.filter_pairs(sizes, arr, k=25, min_kmers=20, min_ident=0.999)) < 13

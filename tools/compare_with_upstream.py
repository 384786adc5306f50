#!/usr/bin/env python3
"""Widen the parity evidence the day the upstream binaries are at hand -- one command:

    python tools/compare_with_upstream.py --kmerdb /path/to/kmer-db --lzani /path/to/lz-ani [--out DIR] [--sets example,phage,nrich]

Nothing of the reference is copied or built: the binaries are ARGUMENTS.  The tool runs them with exactly the argv lists the
reference's own front-end builds (tests/golden/argv_matrix.json: recorded from /root/reference/vclust.py's cmd_kmerdb_* /
cmd_lzani by tools/make_argv_matrix.py) on

  * example   the reference's 12-genome example (tests/golden/example/multifasta.fna) -- must reproduce its goldens,
  * phage     a slice of the phage-1k workload of SURVEY 8(d) (20 families x 10 x 40 kb),
  * nrich     diverged families with N runs, lower-case stretches and IUPAC codes (non-ACGT handling: unpinned upstream),

for every multi-FASTA case of the matrix (defaults, k / min-kmers / min-ident / fraction / max-seqs, every --outfmt and
--out-* filter, --filter with and without a threshold, --out-aln, the non-default LZ parameter set), runs the CPU oracle
(oracle/_build/oracle_cli: the restatement every HIP test is held to) with the same options, and diffs
`fltr.txt` (lines), `ani.tsv` (rows) and `ani.aln.tsv` (multiset of regions).  It prints the regression score of SURVEY 7
hard part 1 -- regions exactly equal / upstream regions, rows with equal fields / upstream rows -- per set and case, and
writes every differing pair under --out for oracle/lzfit + oracle/score_regions.py to re-fit on.

`--self-test` checks the plumbing without upstream binaries: stand-in executables that answer the upstream argv by calling
the oracle CLI (every score must then be 100 %).  It proves the tool, not parity."""
import argparse
import collections
import json
import os
import pathlib
import shutil
import stat
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
ORACLE_CLI = ROOT / 'oracle' / '_build' / 'oracle_cli'
MATRIX = ROOT / 'tests' / 'golden' / 'argv_matrix.json'
EXAMPLE = ROOT / 'tests' / 'golden' / 'example' / 'multifasta.fna'


def make_sets(which, work):
    """-> {name: fasta path}"""
    import numpy as np
    from vclust_amd import synth
    out = {}
    if 'example' in which:
        out['example'] = EXAMPLE
    if 'phage' in which:
        codes, offsets, names, _ = synth.make_workload('phage-1k', 20)
        p = work / 'phage.fna'; synth.write_fasta(p, codes, offsets, names); out['phage'] = p
    if 'nrich' in which:
        codes, offsets, names = synth.make_families(3, 6, length=15000, seed=41, p_lo=0.03, p_hi=0.25)
        rng = np.random.default_rng(41)
        p = work / 'nrich.fna'
        with open(p, 'w') as fh:
            for g, name in enumerate(names):
                s = np.frombuffer(b'ACGT', dtype=np.uint8)[codes[offsets[g]:offsets[g + 1]]].copy()
                for _ in range(6):                                        # N runs, IUPAC codes, lower case
                    a = int(rng.integers(0, len(s) - 400)); s[a:a + int(rng.integers(1, 300))] = ord('N')
                for a in rng.integers(0, len(s), 20): s[a] = ord(rng.choice(list('RYKMSWBDHV')))
                a = int(rng.integers(0, len(s) - 2000)); s[a:a + 1500] = np.frombuffer(s[a:a + 1500].tobytes().lower(), dtype=np.uint8)
                txt = s.tobytes().decode()
                fh.write(f'>{name} some description\n')
                for i in range(0, len(txt), 70): fh.write(txt[i:i + 70] + '\n')
        out['nrich'] = p
    return out


def subst(argv, m):
    out = []
    for a in argv:
        for k, v in m.items():
            a = a.replace(k, v)
        out.append(a)
    return out


def oracle_args(case_argv, tmp, fasta):
    """The oracle CLI's options for a front-end argv of the matrix (vclust.py option names -> oracle_cli's)."""
    a = subst(case_argv, {'<TMP>': str(tmp), '<FASTA>': str(fasta)})
    stage, rest = a[0], a[1:]
    out = [stage]; i = 0; thr = '0'; flt = None
    while i < len(rest):
        o = rest[i]
        if o == '-i': i += 2; continue
        if o in ('-v', '-t'): i += 2; continue
        if o == '--filter': flt = rest[i + 1]; i += 2; continue
        if o == '--filter-threshold': thr = rest[i + 1]; i += 2; continue
        out += [o, rest[i + 1]]; i += 2
    if flt: out += ['--filter', flt, thr]
    return out + [str(fasta)]


def run(cmd, log):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    log.write('$ ' + ' '.join(map(str, cmd)) + '\n' + p.stdout + '\n')
    return p.returncode


def lines(path):
    try:
        return open(path).read().splitlines()
    except OSError:
        return None


def score_lines(up, own):
    """-> (equal, n_upstream, n_own): rows of `up` found in `own` (multiset)"""
    if up is None or own is None:
        return 0, 0 if up is None else len(up), 0 if own is None else len(own)
    c = collections.Counter(own); eq = 0
    for l in up:
        if c[l] > 0: c[l] -= 1; eq += 1
    return eq, len(up), len(own)


def write_stubs(d):
    """Stand-ins for --self-test: executables that take the UPSTREAM argv and answer through the oracle CLI."""
    kd = d / 'kmer-db'; lz = d / 'lz-ani'
    kd.write_text(f'''#!{sys.executable}
import json, subprocess, sys
a = sys.argv[1:]
if not a: sys.exit(0)
def opt(name, default=None):
    return a[a.index(name) + 1] if name in a else default
if a[0] == 'build':
    json.dump(dict(k=opt('-k'), f=opt('-f'), fasta=open(a[-2]).read().split()), open(a[-1], 'w'))
elif a[0].startswith('all2all'):
    d = json.load(open(a[-2])); mins = [a[i + 1] for i, x in enumerate(a) if x == '-min']
    d['min_kmers'] = [m.split(':')[1] for m in mins if m.startswith('num-kmers')][0]
    d['max_seqs'] = (opt('-sample-rows') or 'x:0').split(':')[1]
    json.dump(d, open(a[-1], 'w'))
elif a[0] == 'distance':
    d = json.load(open(a[-2]))
    sys.exit(subprocess.call([{str(ORACLE_CLI)!r}, 'prefilter', '-o', a[-1], '-k', d['k'], '--kmers-fraction', d['f'], '--min-kmers', d['min_kmers'],
                              '--min-ident', opt('-min'), '--max-seqs', d['max_seqs'], *d['fasta']]))
''')
    lz.write_text(f'''#!{sys.executable}
import subprocess, sys
a = sys.argv[1:]
if not a: sys.exit(0)
def opt(name, default=None):
    return a[a.index(name) + 1] if name in a else default
cols = opt('--out-format').split(',')
fmt = 'lite' if len(cols) == 9 else ('complete' if len(cols) == 15 else 'standard')
cmd = [{str(ORACLE_CLI)!r}, 'align', '-o', opt('-o'), '--outfmt', fmt]
for p in ('--mal', '--msl', '--mrd', '--mqd', '--reg', '--aw', '--am', '--ar'): cmd += [p, opt(p)]
if '--out-alignment' in a: cmd += ['--out-aln', opt('--out-alignment')]
if '--flt-kmerdb' in a: i = a.index('--flt-kmerdb'); cmd += ['--filter', a[i + 1], a[i + 2]]
for i, x in enumerate(a):
    if x == '--out-filter': cmd += ['--out-' + a[i + 1], a[i + 2]]
sys.exit(subprocess.call(cmd + open(opt('--in-txt')).read().split()))
''')
    for f in (kd, lz): f.chmod(f.stat().st_mode | stat.S_IXUSR)
    return kd, lz


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--kmerdb'); ap.add_argument('--lzani')
    ap.add_argument('--out', default=None, help='directory for logs and differing files (default: a temp dir that is removed)')
    ap.add_argument('--sets', default='example,phage,nrich')
    ap.add_argument('--threads', default=str(min(os.cpu_count() or 1, 64)))
    ap.add_argument('--self-test', action='store_true')
    args = ap.parse_args()
    if not ORACLE_CLI.exists():
        subprocess.run(['make', '-C', str(ROOT / 'oracle')], check=True, stdout=subprocess.DEVNULL)
    keep = args.out is not None
    work = pathlib.Path(args.out or tempfile.mkdtemp(prefix='vclust_upstream_')); work.mkdir(parents=True, exist_ok=True)
    if args.self_test:
        args.kmerdb, args.lzani = map(str, write_stubs(work))
    if not args.kmerdb or not args.lzani:
        ap.error('--kmerdb and --lzani (paths of the upstream binaries) are required, or --self-test')
    for b in (args.kmerdb, args.lzani):
        if subprocess.run([b], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode != 0:      # validate_binary (vclust.py:663-670)
            sys.exit(f'{b}: does not run (the reference requires exit status 0 without arguments)')
    matrix = json.load(open(MATRIX))
    cases = [c for c in matrix['cases'] if '<FASTA>' in c['argv'] and '-v' not in c['argv']]
    sets = make_sets(args.sets.split(','), work)
    log = open(work / 'commands.log', 'w')
    total = collections.Counter(); report = []
    for sname, fasta in sets.items():
        base = None                                                      # the default prefilter's fltr.txt feeds the --filter cases
        for ci, case in enumerate(cases):
            tmp = work / f'{sname}_{ci:02d}'; otmp = work / f'{sname}_{ci:02d}_oracle'
            for d in (tmp, otmp): shutil.rmtree(d, ignore_errors=True); d.mkdir()
            m = {'<TMP>': str(tmp), '<FASTA>': str(fasta), '<KMERDB>': args.kmerdb, '<LZANI>': args.lzani, '<DEFAULT_THREADS>': args.threads}
            tag = ' '.join(case['argv'][5:]) or 'defaults'
            if case['stage'] == 'prefilter':
                (tmp / 'whole.txt').write_text(f'{fasta}\n')
                rc = 0
                for step in ('build', 'all2all', 'distance'):
                    rc = rc or run(subst(case[step], m), log)
                orc_rc = run([str(ORACLE_CLI), *oracle_args(case['argv'], otmp, fasta)], log)
                eq, nu, no = score_lines(lines(tmp / 'out.txt'), lines(otmp / 'out.txt'))
                if ci == 0: base = tmp / 'out.txt'
                report.append((sname, 'prefilter', tag, f'fltr.txt lines {eq}/{nu}' + (f' (oracle writes {no})' if no != nu else '') + ('' if rc == 0 and orc_rc == 0 else f'  [exit upstream {rc}, oracle {orc_rc}]')))
                total['fltr_eq'] += eq; total['fltr_n'] += nu
            else:
                (tmp / 'ids.txt').write_text(f'{fasta}\n')
                if base is not None and base.exists():
                    shutil.copy(base, tmp / 'fltr.txt'); shutil.copy(base, otmp / 'fltr.txt')
                rc = run(subst(case['lzani'], m), log)
                oa = oracle_args(case['argv'], otmp, fasta)
                orc_rc = run([str(ORACLE_CLI), *oa], log)
                eq, nu, no = score_lines((lines(tmp / 'out.txt') or [None])[1:] if lines(tmp / 'out.txt') else None,
                                         (lines(otmp / 'out.txt') or [None])[1:] if lines(otmp / 'out.txt') else None)
                line = f'ani.tsv rows {eq}/{nu}' + (f' (oracle writes {no})' if no != nu else '')
                total['rows_eq'] += eq; total['rows_n'] += nu
                if '--out-aln' in case['argv']:
                    req, rnu, rno = score_lines((lines(tmp / 'aln.tsv') or [None])[1:] if lines(tmp / 'aln.tsv') else None,
                                                (lines(otmp / 'aln.tsv') or [None])[1:] if lines(otmp / 'aln.tsv') else None)
                    line += f'; regions {req}/{rnu}, surplus in the oracle {rno - req}'
                    total['reg_eq'] += req; total['reg_n'] += rnu; total['reg_surplus'] += rno - req
                ids_u, ids_o = lines(str(tmp / 'out.txt').replace('.txt', '.ids.tsv')), lines(str(otmp / 'out.txt').replace('.txt', '.ids.tsv'))
                if ids_u is not None and ids_u != ids_o: line += '; ids file DIFFERS'
                report.append((sname, 'align', tag, line + ('' if rc == 0 and orc_rc == 0 else f'  [exit upstream {rc}, oracle {orc_rc}]')))
    for r in report:
        print('%-8s %-9s %-70s %s' % r)
    pct = lambda a, b: f'{a}/{b} = {100.0 * a / b:.2f} %' if b else 'n/a'
    print('\nregression score (upstream = truth, oracle = the restatement the HIP path is bit-identical to):')
    print('  fltr.txt lines  ', pct(total['fltr_eq'], total['fltr_n']))
    print('  ani.tsv rows    ', pct(total['rows_eq'], total['rows_n']))
    print('  regions         ', pct(total['reg_eq'], total['reg_n']), f"(surplus regions of the oracle: {total['reg_surplus']})")
    print(f'\nlogs and every output: {work}' if keep else '')
    ok = total['fltr_eq'] == total['fltr_n'] and total['rows_eq'] == total['rows_n'] and total['reg_eq'] == total['reg_n'] and total['reg_surplus'] == 0
    log.close()
    if not keep:
        shutil.rmtree(work, ignore_errors=True)
    sys.exit(0 if ok else 3)


if __name__ == '__main__':
    main()

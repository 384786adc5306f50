#!/usr/bin/env python3
"""Fixture generator (run in the BUILD container, where /root/reference exists): imports the reference's
vclust.py (v1.3.1) and records, for a matrix of command lines, the exact argv lists its cmd_kmerdb_build /
cmd_kmerdb_all2all / cmd_kmerdb_distance (vclust.py:915-1055) and cmd_lzani (vclust.py:1058-1181) produce
with the arguments handle_prefilter / handle_align pass them (vclust.py:1433-1471, 1497-1521).  The result,
tests/golden/argv_matrix.json, pins the CLI -> parameter mapping that vclust_amd/cli.py must reproduce
(tests/test_cli.py::test_cli_maps_to_the_reference_parameters).  Only argv DATA is stored; paths are replaced
by placeholders.

    python tools/make_argv_matrix.py [/root/reference]
"""
import json
import pathlib
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent
REF = pathlib.Path(sys.argv[1] if len(sys.argv) > 1 else '/root/reference')
sys.path.insert(0, str(REF))
import vclust as ref  # noqa: E402

FASTA = ROOT / 'tests' / 'golden' / 'example' / 'multifasta.fna'
FNA_DIR = ROOT / 'tests' / 'golden' / 'example' / 'fna'

PREFILTER = [
    [],
    ['-k', '20'],
    ['--min-kmers', '30'],
    ['--min-ident', '0.95'],
    ['--kmers-fraction', '0.2'],
    ['--max-seqs', '2'],
    ['--max-seqs', '0', '--min-kmers', '1', '--min-ident', '0.5', '-k', '15'],
    ['-k', '30', '--kmers-fraction', '0.05', '--max-seqs', '100', '-t', '3'],
    ['-v', '0'], ['-v', '2'],
]
ALIGN = [
    [],
    ['--outfmt', 'lite'], ['--outfmt', 'complete'],
    ['--out-ani', '0.95', '--out-qcov', '0.85'],
    ['--out-tani', '0.7', '--out-gani', '0.6', '--out-rcov', '0.5'],
    ['--out-ani', '0'],
    ['--filter', '<FLTR>'],
    ['--filter', '<FLTR>', '--filter-threshold', '0.8'],
    ['--out-aln', '<ALN>'],
    ['--mal', '14', '--msl', '8', '--mrd', '60', '--mqd', '70', '--reg', '50', '--aw', '25', '--am', '12', '--ar', '4'],
    ['-v', '0'], ['-v', '2'], ['-t', '5'],
]


def norm(cmd, repl):
    out = []
    for tok in cmd:
        for path, name in repl:
            tok = tok.replace(str(path), name)
        out.append(tok)
    for i, tok in enumerate(out[:-1]):          # the thread default depends on the machine: keep it symbolic
        if tok == '-t' and out[i + 1] == str(ref.DEFAULT_THREAD_COUNT):
            out[i + 1] = '<DEFAULT_THREADS>'
    return out


def main():
    sys.argv = ['vclust.py', 'prefilter', '-i', 'x', '-o', 'y']
    parser = ref.get_parser()
    cases = []
    with tempfile.TemporaryDirectory() as td:
        td = pathlib.Path(td)
        fltr = td / 'fltr.txt'; fltr.write_text('x')
        aln = td / 'aln.tsv'
        out = td / 'out.txt'
        repl = [(td, '<TMP>'), (ref.BIN_KMERDB, '<KMERDB>'), (ref.BIN_LZANI, '<LZANI>'), (FNA_DIR, '<FNA_DIR>'), (FASTA, '<FASTA>')]
        for inp, tag in ((FASTA, '<FASTA>'), (FNA_DIR, '<FNA_DIR>')):
            for extra in PREFILTER:
                argv = ['prefilter', '-i', str(inp), '-o', str(out)] + extra
                sys.argv = ['vclust.py'] + argv
                args = ref.get_parser().parse_args(argv)
                args = ref.validate_args_prefilter(args, parser)
                args = ref.validate_args_fasta_input(args, parser)
                build = ref.cmd_kmerdb_build(input_paths=[inp] if args.is_multifasta else args.fasta_paths, txt_path=td / 'whole.txt',
                                             db_path=td / 'whole.kdb', is_multisample_fasta=args.is_multifasta, kmer_size=args.k,
                                             kmers_fraction=args.kmers_fraction, num_threads=args.num_threads)
                a2a = ref.cmd_kmerdb_all2all(db_paths=[td / 'whole.kdb'], db_list_path=td / 'db_list.txt', outfile_all2all=td / 'all2all.txt',
                                             min_kmers=args.min_kmers, min_ident=args.min_ident, max_seqs=args.max_seqs,
                                             num_threads=args.num_threads)
                dist = ref.cmd_kmerdb_distance(infile_all2all=td / 'all2all.txt', outfile_distance=args.output_path,
                                               min_ident=args.min_ident, num_threads=args.num_threads)
                cases.append(dict(stage='prefilter', argv=['prefilter', '-i', tag, '-o', '<TMP>/out.txt'] + extra,
                                  n_inputs=len(args.fasta_paths), build=norm(build, repl), all2all=norm(a2a, repl), distance=norm(dist, repl)))
            for extra in ALIGN:
                ex = [str(fltr) if t == '<FLTR>' else str(aln) if t == '<ALN>' else t for t in extra]
                argv = ['align', '-i', str(inp), '-o', str(out)] + ex
                sys.argv = ['vclust.py'] + argv
                args = ref.get_parser().parse_args(argv)
                args = ref.validate_args_fasta_input(args, parser)
                cmd = ref.cmd_lzani(input_paths=args.fasta_paths, txt_path=td / 'ids.txt', output_path=args.output_path,
                                    out_format=ref.ALIGN_OUTFMT[args.outfmt], out_aln_path=args.aln_path, out_tani=args.tani,
                                    out_gani=args.gani, out_ani=args.ani, out_qcov=args.qcov, out_rcov=args.rcov,
                                    filter_file=args.filter_path, filter_threshold=args.filter_threshold, mal=args.mal, msl=args.msl,
                                    mrd=args.mrd, mqd=args.mqd, reg=args.reg, aw=args.aw, am=args.am, ar=args.ar,
                                    num_threads=args.num_threads, verbosity_level=args.verbosity_level)
                ex_n = ['<TMP>/fltr.txt' if t == '<FLTR>' else '<TMP>/aln.tsv' if t == '<ALN>' else t for t in extra]
                cases.append(dict(stage='align', argv=['align', '-i', tag, '-o', '<TMP>/out.txt'] + ex_n,
                                  n_inputs=len(args.fasta_paths), lzani=norm(cmd, repl)))
    doc = dict(source=f'reference vclust.py v{ref.__version__}: cmd_kmerdb_build/all2all/distance (vclust.py:915-1055), cmd_lzani (vclust.py:1058-1181)',
               generator='tools/make_argv_matrix.py', default_threads='<DEFAULT_THREADS>', cases=cases)
    text = json.dumps(doc, indent=1)
    (ROOT / 'tests' / 'golden' / 'argv_matrix.json').write_text(text + '\n')
    print(len(cases), 'cases')


if __name__ == '__main__':
    main()

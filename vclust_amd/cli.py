"""Vclust-compatible command line (drop-in for the reference's vclust.py, v1.3.1 surface).

`prefilter` and `align` keep the reference's options, defaults, messages, exit codes and
output files (vclust.py:178-421, 1380-1521) but call libvclust_gpu.so (HIP, MI355X) through
ctypes instead of running bin/kmer-db and bin/lz-ani as subprocesses.  `cluster`,
`deduplicate` and `info` stay what they are in the reference: thin wrappers over the CPU tools
bin/clusty and bin/mfasta-tool (out of scope of the GPU path), used if those binaries exist.

Multi-GPU: start one process per GPU, e.g.
    python -m torch.distributed.run --nproc-per-node 8 vclust.py align -i x.fna -o ani.tsv ...
Each rank computes its share of the integer work; rows are gathered on rank 0 over RCCL.
"""
import argparse
import logging
import multiprocessing
import os
import pathlib
import subprocess
import sys

__version__ = '1.3.1'

DEFAULT_THREAD_COUNT = min(multiprocessing.cpu_count(), 64)

SCRIPT_DIR = pathlib.Path(__file__).resolve().parent.parent
BIN_DIR = SCRIPT_DIR / 'bin'
BIN_CLUSTY = BIN_DIR / 'clusty'
BIN_MFASTA = BIN_DIR / 'mfasta-tool'

# LZ-ANI output columns and the three output formats (same sets as the reference, vclust.py:38-47)
ALIGN_FIELDS = ['qidx', 'ridx', 'query', 'reference', 'tani', 'gani', 'ani', 'qcov', 'rcov',
                'num_alns', 'len_ratio', 'qlen', 'rlen', 'nt_match', 'nt_mismatch']
ALIGN_OUTFMT = {
    'lite': ALIGN_FIELDS[:2] + ALIGN_FIELDS[4:11],
    'standard': ALIGN_FIELDS[:11],
    'complete': ALIGN_FIELDS[:],
}


# ------------------------------------------------------------------ argument parsing
class _HelpFormatter(argparse.RawDescriptionHelpFormatter):
    def _format_action_invocation(self, action):
        if not action.option_strings or action.nargs == 0:
            return super()._format_action_invocation(action)
        metavar = self._format_args(action, self._get_default_metavar_for_optional(action))
        return ', '.join(action.option_strings) + ' ' + metavar

    def _split_lines(self, text, width):
        out = []
        for part in text.splitlines():
            out.extend(argparse.HelpFormatter._split_lines(self, part, width))
        return out


def _existing_path(value):
    path = pathlib.Path(value)
    if not path.exists():
        raise argparse.ArgumentTypeError(f'input does not exist: {value}')
    return path


def _unit_float(value):
    f = float(value)
    if f < 0 or f > 1:
        raise argparse.ArgumentTypeError('must be between 0 and 1')
    return f


def get_parser() -> argparse.ArgumentParser:
    fmt = lambda prog: _HelpFormatter(prog, max_help_position=32, width=100)  # noqa: E731

    def common(p, threads=True):
        if threads:
            p.add_argument('-t', '--threads', metavar='<int>', dest='num_threads', type=int,
                           default=DEFAULT_THREAD_COUNT, help='Number of threads [%(default)s]')
        p.add_argument('-v', metavar='<int>', dest='verbosity_level', type=int, choices=[0, 1, 2], default=1,
                       help='Verbosity level [%(default)s]:\n0: Errors only\n1: Info\n2: Debug')
        p.add_argument('-h', '--help', action='help', help='Show this help message and exit')

    def io_args(p, in_help):
        req = p.add_argument_group('required arguments')
        req.add_argument('-i', '--in', metavar='<file>', type=_existing_path, dest='input_path', help=in_help,
                         required=True)
        req.add_argument('-o', '--out', metavar='<file>', type=pathlib.Path, dest='output_path',
                         help='Output filename', required=True)
        return req

    parser = argparse.ArgumentParser(
        description=f'%(prog)s v{__version__}: calculate ANI and cluster virus (meta)genome sequences '
                    '(MI355X-native prefilter/align)',
        formatter_class=fmt, add_help=False)
    parser.add_argument('-v', '--version', action='version', version=f'v{__version__}',
                        help="Display the tool's version and exit")
    parser.add_argument('-h', '--help', action='help', help='Show this help message and exit')
    sub = parser.add_subparsers(dest='command')

    # deduplicate (CPU tool wrapper, out of scope of the GPU path)
    dd = sub.add_parser('deduplicate', help='Deduplicate and merge genome sequences from multiple FASTA files',
                        formatter_class=fmt, add_help=False)
    ddr = dd.add_argument_group('required arguments')
    ddr.add_argument('-i', '--in', metavar='<file>', type=_existing_path, dest='input_path', nargs='+',
                     help='Space-separated input FASTA files (gzipped or uncompressed)')
    ddr.add_argument('-o', '--out', metavar='<file>', type=pathlib.Path, dest='output_path', required=True,
                     help='Output FASTA file with unique, nonredundant sequences')
    dd.add_argument('--add-prefixes', metavar='<str>', nargs='*', default=False,
                    help='Add prefixes to sequence IDs [%(default)s]')
    dd.add_argument('--gzip-output', action='store_true', help='Compress the output FASTA file with gzip')
    dd.add_argument('--gzip-level', metavar='<int>', type=int, default=4, help='Compression level (1-9) [%(default)s]')
    common(dd)

    # prefilter
    pf = sub.add_parser('prefilter', help='Prefilter genome pairs for alignment', formatter_class=fmt, add_help=False)
    io_args(pf, 'Input FASTA file or directory of files (gzipped or uncompressed)')
    pf.add_argument('-k', '--k', metavar='<int>', type=int, default=25, choices=range(15, 31),
                    help='Size of k-mer for Kmer-db [%(default)s]')
    pf.add_argument('--min-kmers', metavar='<int>', type=int, default=20,
                    help='Minimum number of shared k-mers between two genomes [%(default)s]')
    pf.add_argument('--min-ident', metavar='<float>', type=_unit_float, default=0.7,
                    help='Minimum sequence identity (0-1) between two genomes, relative to the shorter one '
                         '[%(default)s]')
    pf.add_argument('--batch-size', metavar='<int>', type=int, default=0,
                    help='Process a multifasta file in batches of <int> sequences (accepted for '
                         'compatibility; the GPU path produces identical results) [%(default)s]')
    pf.add_argument('--kmers-fraction', metavar='<float>', type=_unit_float, default=1.0,
                    help='Fraction of k-mers to analyze in each genome (0-1) [%(default)s]')
    pf.add_argument('--max-seqs', metavar='<int>', type=int, default=0,
                    help='Maximum number of sequences allowed to pass the prefilter per query; 0 = all '
                         '[%(default)s]')
    common(pf)

    # align
    al = sub.add_parser('align', help='Align genome sequence pairs and calculate ANI measures',
                        formatter_class=fmt, add_help=False)
    io_args(al, 'Input FASTA file or directory of files (gzipped or uncompressed)')
    al.add_argument('--filter', metavar='<file>', type=_existing_path, dest='filter_path',
                    help='Path to filter file (output of prefilter)')
    al.add_argument('--filter-threshold', metavar='<float>', dest='filter_threshold', type=_unit_float, default=0,
                    help='Align genome pairs above the threshold (0-1) [%(default)s]')
    al.add_argument('--outfmt', metavar='<str>', choices=ALIGN_OUTFMT.keys(), dest='outfmt', default='standard',
                    help='Output format [%(default)s]\nchoices: ' + ','.join(ALIGN_OUTFMT.keys()))
    al.add_argument('--out-aln', metavar='<file>', type=pathlib.Path, dest='aln_path',
                    help='Write alignments to the tsv <file>')
    for name, label in (('ani', 'ANI'), ('tani', 'tANI'), ('gani', 'gANI'),
                        ('qcov', 'query coverage (aligned fraction)'), ('rcov', 'reference coverage (aligned fraction)')):
        al.add_argument(f'--out-{name}', dest=name, metavar='<float>', type=_unit_float, default=0,
                        help=f'Min. {label} to output (0-1) [%(default)s]')
    for name, default, text in (('mal', 11, 'Min. anchor length'), ('msl', 7, 'Min. seed length'),
                                ('mrd', 40, 'Max. dist. between approx. matches in reference'),
                                ('mqd', 40, 'Max. dist. between approx. matches in query'),
                                ('reg', 35, 'Min. considered region length'), ('aw', 15, 'Approx. window length'),
                                ('am', 7, 'Max. no. of mismatches in approx. window'),
                                ('ar', 3, 'Min. length of run ending approx. extension')):
        al.add_argument(f'--{name}', metavar='<int>', type=int, default=default, help=f'{text} [%(default)s]')
    common(al)

    # cluster (CPU tool wrapper)
    cl = sub.add_parser('cluster', help='Cluster genomes based on ANI thresholds', formatter_class=fmt, add_help=False)
    clr = io_args(cl, 'Input file with ANI metrics (tsv)')
    clr.add_argument('--ids', metavar='<file>', type=_existing_path, dest='ids_path', required=True,
                     help='Input file with sequence identifiers (tsv)')
    cl.add_argument('-r', '--out-repr', action='store_true', dest='representatives',
                    help='Output a representative genome for each cluster [%(default)s]')
    cl.add_argument('--algorithm', metavar='<str>', dest='algorithm', default='single',
                    choices=['single', 'complete', 'uclust', 'cd-hit', 'set-cover', 'leiden'],
                    help='Clustering algorithm [%(default)s]')
    cl.add_argument('--metric', metavar='<str>', dest='metric', choices=['tani', 'gani', 'ani'], default='tani',
                    help='Similarity metric for clustering [%(default)s]')
    for name in ('tani', 'gani', 'ani', 'qcov', 'rcov', 'len_ratio'):
        cl.add_argument(f'--{name}', metavar='<float>', dest=name, type=_unit_float, default=0,
                        help=f'Min. {name} (0-1) [%(default)s]')
    cl.add_argument('--num_alns', metavar='<int>', dest='num_alns', type=int, default=0,
                    help='Max. number of local alignments between two genomes; 0 = all [%(default)s]')
    cl.add_argument('--leiden-resolution', metavar='<float>', type=_unit_float, default=0.7)
    cl.add_argument('--leiden-beta', metavar='<float>', type=_unit_float, default=0.01)
    cl.add_argument('--leiden-iterations', metavar='<int>', type=int, default=2)
    common(cl, threads=False)

    sub.add_parser('info', help='Show information about the tool and its dependencies', formatter_class=fmt,
                   add_help=False)

    # no arguments: help on stdout, exit 0 (test.py:41-55)
    if len(sys.argv[1:]) == 0:
        parser.print_help()
        parser.exit()
    for name, sp in (('prefilter', pf), ('align', al), ('cluster', cl), ('deduplicate', dd)):
        if sys.argv[-1] == name:
            sp.print_help()
            parser.exit()
    return parser


# ------------------------------------------------------------------ logging
class _LogFormatter(logging.Formatter):
    FMT = '{asctime} [{levelname:^7}] {message}'
    COLORS = {logging.DEBUG: '{}', logging.INFO: '\33[36m{}\33[0m', logging.WARNING: '\33[33m{}\33[0m',
              logging.ERROR: '\33[31m{}\33[0m'}

    def format(self, record):
        fmt = self.COLORS.get(record.levelno, '{}').replace('{}', self.FMT)
        return logging.Formatter(fmt, style='{').format(record)


def create_logger(name: str, verbosity_level: int) -> logging.Logger:
    level = {0: logging.ERROR, 1: logging.INFO, 2: logging.DEBUG}.get(verbosity_level, logging.ERROR)
    logger = logging.getLogger(name)
    logger.setLevel(level)
    if not logger.handlers:
        handler = logging.StreamHandler()
        handler.setLevel(level)
        handler.setFormatter(_LogFormatter())
        logger.addHandler(handler)
    return logger


# ------------------------------------------------------------------ helpers
def validate_args_fasta_input(args, parser):
    """file -> one multi-FASTA; directory -> one genome per file, sorted, at least two."""
    args.is_multifasta = True
    args.fasta_paths = [args.input_path]
    if args.input_path.is_dir():
        args.is_multifasta = False
        args.fasta_paths = sorted(f for f in args.input_path.iterdir() if f.is_file())
    if not args.is_multifasta and len(args.fasta_paths) < 2:
        parser.error(f'Too few fasta files found in {args.input_path}. '
                     f'Expected at least 2, found {len(args.fasta_paths)}.')
    return args


def validate_args_prefilter(args, parser):
    if args.batch_size and args.input_path.is_dir():
        parser.error('--batch-size only handles a multi-fasta file, not a directory.')
    return args


def _dist_env():
    """(rank, world, local_rank) of a torchrun launch, (0, 1, 0) otherwise."""
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def run_native(description, fn, verbosity_level, logger):
    """The reference's run(): log 'Running', call, log 'Completed'; errors -> ERROR log + exit 1."""
    logger.info(f'Running: {description}')
    try:
        fn()
    except Exception as e:      # library error codes arrive as VclustGpuError
        logger.error(f'Process {description} failed with message: {e}')
        sys.exit(1)
    logger.info('Completed')


def run_subprocess(cmd, verbosity_level, logger):
    logger.info(f'Running: {" ".join(cmd)}')
    try:
        subprocess.run(cmd, stdout=None if verbosity_level else subprocess.DEVNULL,
                       stderr=None if verbosity_level else subprocess.PIPE, text=True, check=True)
    except subprocess.CalledProcessError as e:
        logger.error(f'Process {" ".join(cmd)} failed with message: {e.stderr}')
        sys.exit(1)
    except OSError as e:
        logger.error(f'OSError: {" ".join(cmd)} failed with message: {e}')
        sys.exit(1)
    logger.info('Completed')


def _require_binary(path):
    if not path.exists() or not os.access(path, os.X_OK):
        sys.exit(f'error: File not found: {path} (CPU tool, not part of the GPU path; build it from the '
                 'reference repository and place it in bin/)')


# ------------------------------------------------------------------ handlers
def prefilter_call(args):
    """What the front-end hands to vg_prefilter for validated prefilter arguments -- the counterpart of the
    reference's three argv lists (cmd_kmerdb_build / _all2all / _distance, vclust.py:915-1055)."""
    return dict(paths=args.fasta_paths, out_path=args.output_path, is_multifasta=args.is_multifasta, k=args.k,
                min_kmers=args.min_kmers, min_ident=args.min_ident, kmers_fraction=args.kmers_fraction,
                max_seqs=args.max_seqs, num_threads=args.num_threads)


def align_call(args):
    """What the front-end hands to vg_align for validated align arguments -- the counterpart of cmd_lzani
    (vclust.py:1058-1181): `--out-filter` only for values > 0 (:1170-1176), `--multisample-fasta true` iff there
    is exactly one input path (:1159-1160)."""
    lz = {k: getattr(args, k) for k in ('mal', 'msl', 'mrd', 'mqd', 'reg', 'aw', 'am', 'ar')}
    out_filters = {k: getattr(args, k) for k in ('tani', 'gani', 'ani', 'qcov', 'rcov') if getattr(args, k) > 0}
    return dict(paths=args.fasta_paths, out_path=args.output_path, is_multifasta=args.is_multifasta,
                columns=ALIGN_OUTFMT[args.outfmt], filter_path=args.filter_path, filter_threshold=args.filter_threshold,
                out_aln=args.aln_path, lz=lz, out_filters=out_filters, num_threads=args.num_threads)


def handle_prefilter(args, parser, logger):
    args = validate_args_prefilter(args, parser)
    args = validate_args_fasta_input(args, parser)
    from . import stages
    rank, world, local_rank = _dist_env()
    desc = (f'libvclust_gpu prefilter -k {args.k} --min-kmers {args.min_kmers} --min-ident {args.min_ident} '
            f'--kmers-fraction {args.kmers_fraction} --max-seqs {args.max_seqs} [{world} GPU] -> {args.output_path}')

    def work():
        kw = prefilter_call(args)
        paths, out_path, multi = kw.pop('paths'), kw.pop('out_path'), kw.pop('is_multifasta')
        if world == 1:
            stages.prefilter(paths, out_path, multi, batch_size=args.batch_size, verbosity=args.verbosity_level, **kw)
        else:
            from . import distributed
            distributed.prefilter(paths, out_path, multi, **kw)
    run_native(desc, work, args.verbosity_level, logger)


def handle_align(args, parser, logger):
    args = validate_args_fasta_input(args, parser)
    from . import stages
    rank, world, local_rank = _dist_env()
    call = align_call(args)
    desc = ('libvclust_gpu align ' + ' '.join(f'--{k} {v}' for k, v in call['lz'].items())
            + (f' --filter {args.filter_path} {args.filter_threshold}' if args.filter_path else '')
            + f' [{world} GPU] -> {args.output_path}')

    def work():
        kw = dict(call)
        paths, out_path, multi = kw.pop('paths'), kw.pop('out_path'), kw.pop('is_multifasta')
        if world == 1:
            stages.align(paths, out_path, multi, verbosity=args.verbosity_level, **kw)
        else:
            from . import distributed
            distributed.align(paths, out_path, multi, **kw)
    run_native(desc, work, args.verbosity_level, logger)


def handle_cluster(args, parser, logger):
    _require_binary(BIN_CLUSTY)
    threshold = vars(args).get(args.metric, 0)
    if not threshold:
        parser.error(f'{args.metric} threshold must be above 0. Specify the option: --{args.metric}')
    cmd = [str(BIN_CLUSTY), '--objects-file', str(args.ids_path), '--algo', args.algorithm, '--id-cols', 'qidx', 'ridx',
           '--distance-col', args.metric, '--similarity', '--numeric-ids']
    for name in ('tani', 'gani', 'ani', 'qcov', 'rcov', 'len_ratio'):
        if getattr(args, name) > 0:
            cmd += ['--min', name, str(getattr(args, name))]
    if args.num_alns > 0:
        cmd += ['--max', 'num_alns', str(args.num_alns)]
    if args.representatives:
        cmd.append('--out-representatives')
    if args.algorithm == 'leiden':
        cmd += ['--leiden-resolution', str(args.leiden_resolution), '--leiden-beta', str(args.leiden_beta),
                '--leiden-iterations', str(args.leiden_iterations)]
    cmd += [str(args.input_path), str(args.output_path)]
    run_subprocess(cmd, args.verbosity_level, logger)


def handle_deduplicate(args, parser, logger):
    _require_binary(BIN_MFASTA)
    out_dup = pathlib.Path(f'{args.output_path}.duplicates.txt')
    cmd = [str(BIN_MFASTA), 'mrds', '-i', ','.join(str(f) for f in args.input_path), '-o', str(args.output_path),
           '--out-duplicates', str(out_dup), '--remove-duplicates', '--mark-duplicates-orientation',
           '--rev-comp-as-equivalent', '-t', str(args.num_threads)]
    if args.verbosity_level:
        cmd += ['--verbosity', '1']
    if args.add_prefixes:
        cmd += ['--in-prefixes', ','.join(args.add_prefixes)]
    if args.gzip_output:
        cmd += ['--gzipped-output', '--gzip-level', str(args.gzip_level)]
    run_subprocess(cmd, args.verbosity_level, logger)


def handle_info(args, parser, logger):
    lines = [f'Vclust (MI355X-native prefilter/align) version {__version__}', '', 'GPU path:']
    ok = True
    try:
        from . import api
        lines.append(f'   libvclust_gpu        {api.version()}')
        n = api.device_count()
        lines.append(f'   HIP devices          {n}')
        ok = n > 0
    except Exception as e:      # noqa: BLE001
        lines.append(f'   libvclust_gpu        [error] {e}')
        ok = False
    lines += ['', 'Parity with the reference (golden example of refresh-bio/vclust, DESIGN.md section 2):',
              '   fltr.txt, ani.ids.tsv            byte-identical',
              '   ani.aln.tsv                      5693 / 5693 regions identical, none surplus',
              '   ani.tsv                          132 / 132 rows byte-identical']
    lines += ['', 'CPU tools (optional):']
    for name, path in (('Clusty', BIN_CLUSTY), ('mfasta', BIN_MFASTA)):
        lines.append(f'   {name:<20} {"found" if path.exists() else "not installed"} ({path})')
    lines += ['', '\033[32;1mStatus: ready\033[0m' if ok else '\033[31mStatus: error\033[0m']
    print('\n'.join(lines))
    if not ok:
        sys.exit(1)


def main():
    parser = get_parser()
    args = parser.parse_args()
    rank, world, _ = _dist_env()
    logger = create_logger('Vclust', getattr(args, 'verbosity_level', 0) if rank == 0 else 0)
    handlers = {'info': handle_info, 'deduplicate': handle_deduplicate, 'prefilter': handle_prefilter,
                'align': handle_align, 'cluster': handle_cluster}
    if args.command in handlers:
        handlers[args.command](args, parser, logger)


if __name__ == '__main__':
    main()

"""Command-line surface of the drop-in front-end, mirroring the reference's black-box tests
(test.py:41-133): help on no arguments, parser errors with exit code 2 and the same messages."""
import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
VCLUST = ROOT / 'vclust.py'
EX = ROOT / 'tests' / 'golden' / 'example'


def run(*args):
    return subprocess.run([sys.executable, str(VCLUST), *map(str, args)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def test_no_arguments_prints_help():
    p = run()
    assert p.returncode == 0 and p.stdout and not p.stderr


@pytest.mark.parametrize('sub', ['deduplicate', 'prefilter', 'align', 'cluster'])
def test_subcommand_without_arguments_prints_help(sub):
    p = run(sub)
    assert p.returncode == 0 and not p.stderr and f'vclust.py {sub}' in p.stdout


def test_version():
    p = run('--version')
    assert p.returncode == 0 and 'v1.3.1' in (p.stdout + p.stderr)


@pytest.mark.parametrize('args,msg', [
    (['-i', EX / 'fna', '-o', 'x', '--batch-size', '4'], 'error: --batch-size'),
    (['-i', EX / 'multifasta.fna', '-o', 'x', '--min-ident', '95'], 'between 0 and 1'),
    (['-i', EX / 'multifasta.fna', '-o', 'x', '--kmers-fraction', '10'], 'between 0 and 1'),
    (['-i', EX / 'multifasta.fna', '-o', 'x', '--k', '2'], 'invalid choice'),
    (['-i', 'missing.fna', '-o', 'x'], 'does not exist'),
])
def test_parser_error_prefilter(args, msg):
    p = run('prefilter', *args)
    assert p.returncode == 2 and msg in p.stderr


@pytest.mark.parametrize('args,msg', [
    (['-i', EX / 'multifasta.fna', '-o', 'x', '--out-tani', '40'], 'between 0 and 1'),
    (['-i', 'missing.fna', '-o', 'x'], 'does not exist'),
    (['-i', EX / 'multifasta.fna', '-o', 'x', '--outfmt', 'fancy'], 'invalid choice'),
])
def test_parser_error_align(args, msg):
    p = run('align', *args)
    assert p.returncode == 2 and msg in p.stderr


def test_outfmt_columns_match_reference_sets():
    sys.path.insert(0, str(ROOT))
    import vclust
    assert vclust.ALIGN_OUTFMT['standard'] == ['qidx', 'ridx', 'query', 'reference', 'tani', 'gani', 'ani', 'qcov',
                                               'rcov', 'num_alns', 'len_ratio']
    assert vclust.ALIGN_OUTFMT['lite'] == ['qidx', 'ridx', 'tani', 'gani', 'ani', 'qcov', 'rcov', 'num_alns', 'len_ratio']
    assert vclust.ALIGN_OUTFMT['complete'][-4:] == ['qlen', 'rlen', 'nt_match', 'nt_mismatch']


def test_without_gpu_the_stage_fails_loudly(tmp_path):
    """No CPU fallback: on a box without a HIP device prefilter exits 1 with an ERROR log line."""
    from vclust_amd import api
    if api.device_count() > 0:
        pytest.skip('a HIP device is visible')
    p = run('prefilter', '-i', EX / 'multifasta.fna', '-o', tmp_path / 'f.txt')
    assert p.returncode == 1
    assert 'Running' in p.stderr and 'ERROR' in p.stderr and 'no CPU fallback' in p.stderr
    assert not (tmp_path / 'f.txt').exists()

import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

GOLDEN = ROOT / 'tests' / 'golden' / 'example'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: full-size configuration (about two minutes on the GPU box); deselect with -m "gpu and not slow"')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN

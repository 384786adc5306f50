"""Developer micro-benchmark: one divergence level, for PMC runs."""
import sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vclust_amd import api
from tools.lz_micro import ident, run
if __name__ == '__main__':
    api.set_device(0)
    rate = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4500
    c, o, p = ident(n, 40000, rate)
    run(f'subst rate {rate}', c, o, p, reps=2)

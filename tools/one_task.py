"""Developer tool: duration of single parse tasks run alone (VG_LZ_ABLATE=32 timing)."""
import os, sys, pathlib
import numpy as np
os.environ['VG_LZ_ABLATE'] = os.environ.get('VG_LZ_ABLATE', '32')
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vclust_amd import api, synth
api.set_device(0)
c, o, n = synth.make_families(100, 10, 40000, seed=1)
gs = api.GenomeSet.from_codes(c, o, n); gs.to_device()
for (q, r) in ((910, 912), (912, 910), (0, 1), (5, 3)):
    t = np.array([(q, r)], dtype=api.TASK_DTYPE)
    gs.lz_align(t)
    st = gs.lz_align(t)
    print((q, r), 'ticks/100 = %.1f us' % (st['n_regions'][0] / 100.0), 'M', st['n_match'][0], 'A', st['aln_len'][0])

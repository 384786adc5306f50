"""Developer micro-benchmark of the LZ parse kernel (not part of the product or the tests)."""
import sys, pathlib, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vclust_amd import api, synth

def run(label, codes, offsets, pairs, reps=3):
    gs = api.GenomeSet.from_codes(codes, offsets)
    gs.to_device()
    tasks = gs.align_tasks(pairs)
    gs.lz_align(tasks)
    api.profile_enable(True); api.profile_reset()
    for _ in range(reps):
        st = gs.lz_align(tasks)
    prof = {e['name']: e['total_ms'] / e['launches'] for e in api.profile_get()}
    api.profile_enable(False)
    print(f'{label:28s} tasks {len(tasks):6d}  parse {prof.get("lz_parse",0):8.3f} ms  build {prof.get("lz_build_index",0):7.3f} ms  '
          f'per-task {prof.get("lz_parse",0)*1e3/len(tasks):7.3f} us  sumM {int(st["n_match"].sum())} regs {int(st["n_regions"].sum())}')

def ident(n, L, rate, seed=5):
    rng = np.random.default_rng(seed)
    seqs = []
    for i in range(n):
        a = rng.integers(0, 4, size=L, dtype=np.uint8)
        b = a.copy()
        m = rng.random(L) < rate
        b[m] = (b[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
        seqs += [a, b]
    off = np.zeros(len(seqs) + 1, dtype=np.int64); off[1:] = np.cumsum([len(s) for s in seqs])
    pairs = np.array([(2 * i + 1, 2 * i, 0) for i in range(n)], dtype=api.PAIR_DTYPE)
    return np.concatenate(seqs), off, pairs

if __name__ == '__main__':
    api.set_device(0)
    for rate in (0.0, 0.01, 0.05, 0.10, 0.20, 0.30, 0.75):
        c, o, p = ident(4500, 40000, rate)
        run(f'subst rate {rate}', c, o, p)
    c, o, n = synth.make_families(100, 10, 40000, seed=1)
    run('phage-1k families', c, o, synth.family_pairs(100, 10))

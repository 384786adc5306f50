"""Developer tool: per-pair trip / event / time statistics of the LZ parse through the dev kernel (VG_LZ_ABLATE).
Needs the developer build (VG_DEV=1 python -m vclust_amd.build --force; rebuild without VG_DEV afterwards).
Run as: python tools/micro/lz_stats.py [families]  (spawns one process per setting: the knob is read once per process)."""
import os, sys, subprocess, pathlib, json
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
CODE = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
from vclust_amd import api, synth
nf = int(sys.argv[1])
codes, offsets, names, _ = synth.make_workload('phage-100k', nf)
gs = api.GenomeSet.from_codes(codes, offsets, names)
sizes, pairs = gs.kmer_shared(k=25, min_shared=20)
tasks = gs.align_tasks(gs.filter_pairs(sizes, pairs))
st = gs.lz_align(tasks)
print(json.dumps(dict(M=st['n_match'].astype(int).tolist(), A=st['aln_len'].astype(int).tolist(), N=st['n_regions'].astype(int).tolist())))
''' % str(ROOT)
nf = sys.argv[1] if len(sys.argv) > 1 else '100'
import numpy as np
def run(abl):
    env = dict(os.environ); 
    env['VG_DEV_SWITCHES'] = '1'
    if abl is not None: env['VG_LZ_ABLATE'] = str(abl)
    p = subprocess.run([sys.executable, '-c', CODE, nf], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    d = json.loads(p.stdout.strip().splitlines()[-1]); return {k: np.array(v) for k, v in d.items()}
base = run(None)
print('pairs', len(base['M']), 'regions/pair mean', base['N'].mean())
d = run(1024)      # M = probe trips, A = events, N = wall ticks (100 MHz)
def pct(x): return 'mean %.1f p50 %.0f p90 %.0f p99 %.0f max %.0f' % (x.mean(), *np.percentile(x, [50, 90, 99]), x.max())
print('probe trips / pair  ', pct(d['M'])); print('events / pair       ', pct(d['A'])); print('wave time us / pair ', pct(d['N'] / 100.0))
print('trips per event', d['M'].sum() / d['A'].sum())
d2 = run(1024 | 2048)   # M = max anchor-bucket trips, A = seed ... (bucket loop trips)
print('bucket-loop trips / pair', pct(d2['M']))
for ps, nm in enumerate(['probe', 'ext loads', 'left ext', 'fwd ext + gap']):
    d3 = run(128 | (ps << 8))
    print('cycles in %-14s' % nm, pct(d3['M'] * 16.0), ' per event %.0f' % ((d3['M'] * 16.0).sum() / d3['A'].sum()))

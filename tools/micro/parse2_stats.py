#!/usr/bin/env python3
"""Developer tool: loop statistics of the two-pairs-per-wave parse (VG_DEV_SWITCHES=1 VG_LZ_KERNEL=two_stats): events per pair,
iterations of a wave while a pair is resident, iterations in which both halves had an event."""
import os, sys, pathlib
os.environ['VG_DEV_SWITCHES'] = '1'; os.environ['VG_LZ_KERNEL'] = 'two_stats'
SEL = os.environ.get('VG_LZ_P2SEL')
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import numpy as np
from vclust_amd import api, synth
api.set_device(0)
NF = int(os.environ.get('NF', '2000'))
codes, offsets, names, _ = synth.make_workload('phage-100k', NF)
gs = api.GenomeSet.from_codes(codes, offsets, names); gs.to_device()
sizes, pairs = gs.kmer_shared(k=25, min_shared=20)
tasks = gs.align_tasks(gs.filter_pairs(sizes, pairs))
st = gs.lz_align(tasks)
ev, it, both = st['n_match'].astype(np.int64), st['aln_len'].astype(np.int64), st['n_regions'].astype(np.int64)
print(f'{len(tasks)} pairs: events per pair {ev.mean():.1f}; iterations while resident {it.mean():.1f} (a wave of one pair per wave would run {ev.mean():.1f} event trips + misses); '
      f'iterations with an event in BOTH halves {both.mean():.1f} per pair = {both.sum() / max(ev.sum(), 1):.2f} of its events; wave iterations per event {it.sum() / 2 / max(ev.sum(), 1):.2f}')
if SEL is not None:
    names = ['done / take', 'probe', 'event set-up (broadcasts, first-round loads)', 'backward extension', 'forward extension', 'gap score + state']
    print(f'section {SEL} ({names[int(SEL)]}): {ev.sum() * 16 / max(it.sum(), 1):.0f} clock ticks (100 MHz) per resident iteration')

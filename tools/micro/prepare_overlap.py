#!/usr/bin/env python3
"""Developer experiment: the reference indexes of ALL candidate genomes built on a queue of their own BESIDE the prefilter
pass (VG_DEV_SWITCHES=1 VG_LZ_PREPARE_QUEUE=own, MODE=early: the references of the previous step are prepared before the
prefilter of this one) against the normal order (MODE=late: prepare after the thresholds).  Prints ms per step."""
import os, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from vclust_amd import api, synth
api.set_device(0)
NF = int(os.environ.get('NF', '10000')); MODE = os.environ.get('MODE', 'late'); STEPS = int(os.environ.get('STEPS', '6'))
codes, offsets, names, _ = synth.make_workload('phage-100k', NF)
gs = api.GenomeSet.from_codes(codes, offsets, names); gs.to_device()
sizes, pairs = gs.kmer_shared(k=25, min_shared=20); cand = gs.filter_pairs(sizes, pairs)
ts = []
for it in range(STEPS + 2):
    if it == 2: api.profile_enable(True); api.profile_reset()
    t0 = time.perf_counter()
    if MODE == 'early': gs.lz_prepare(cand)
    sizes, pairs = gs.kmer_shared(k=25, min_shared=20)
    cand = gs.filter_pairs(sizes, pairs)
    if MODE == 'late': gs.lz_prepare(cand)
    tasks = gs.align_tasks(cand)
    st = gs.lz_align(tasks)
    ts.append((time.perf_counter() - t0) * 1e3)
prof = {e['name']: round(e['total_ms'] / STEPS, 1) for e in api.profile_get()}
print(MODE, os.environ.get('VG_LZ_PREPARE_QUEUE', 'library'), 'ms per step', [round(t, 1) for t in ts[2:]], prof, int(st['n_match'].sum()), flush=True)

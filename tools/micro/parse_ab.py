#!/usr/bin/env python3
"""Developer tool: the LZ parse alone on the phage-100k set (NF families), REPS times: ms of the lz_parse scope and a hash
of the rows -- run once per kernel variant (the switches are read once per process) and compare:
  VG_DEV_SWITCHES=1 python tools/micro/parse_ab.py ; VG_DEV_SWITCHES=1 VG_LZ_INDEX=fused python tools/micro/parse_ab.py"""
import hashlib, os, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from vclust_amd import api, synth
api.set_device(0)
NF = int(os.environ.get('NF', '10000')); REPS = int(os.environ.get('REPS', '3'))
codes, offsets, names, _ = synth.make_workload('phage-100k', NF)
gs = api.GenomeSet.from_codes(codes, offsets, names); gs.to_device()
sizes, pairs = gs.kmer_shared(k=25, min_shared=20)
tasks = gs.align_tasks(gs.filter_pairs(sizes, pairs))
st = gs.lz_align(tasks)
api.profile_enable(True); api.profile_reset()
for _ in range(REPS):
    st = gs.lz_align(tasks)
prof = {e['name']: round(e['total_ms'] / REPS, 2) for e in api.profile_get()}
print(os.environ.get('VG_LZ_KERNEL', 'default'), 'index=' + os.environ.get('VG_LZ_INDEX', 'default'), len(tasks), 'tasks', prof, hashlib.sha256(st.tobytes()).hexdigest()[:16], flush=True)

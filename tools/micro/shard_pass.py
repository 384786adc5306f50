#!/usr/bin/env python3
"""Developer tool: one k-mer RANGE shard pass of the prefilter (shard 0 of WORLD) on the phage-100k set, REPS times -- the
command to put under `rocprofv3 --kernel-trace --stats` when the per-rank cost of the sharded prefilter is looked at.
  WORLD=8 REPS=5 NF=10000 python tools/micro/shard_pass.py"""
import os, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from vclust_amd import api, synth
api.set_device(0)
NF = int(os.environ.get('NF', '10000')); WORLD = int(os.environ.get('WORLD', '8')); REPS = int(os.environ.get('REPS', '5'))
codes, offsets, names, _ = synth.make_workload('phage-100k', NF)
gs = api.GenomeSet.from_codes(codes, offsets, names); gs.to_device()
if os.environ.get('SCAN', 'replicated') == 'sliced':
    api.set_range_scan(1)         # the multi-GPU form: 1 / WORLD of the bases scanned, the peers' slices computed here (scope emulated_peer_scan)
for it in range(REPS):
    t0 = time.perf_counter()
    s, p = gs.kmer_shared(k=25, shard=0, n_shards=WORLD, min_shared=1 if WORLD > 1 else 20)
    print(f'pass {it}: {(time.perf_counter() - t0) * 1e3:.1f} ms, {len(p)} pairs', flush=True)

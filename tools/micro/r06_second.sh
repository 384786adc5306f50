#!/bin/bash
# round 6, second GPU call: host-side phase marks of one warm step and of one shard pass, counters of the fused-slot probe A/B
set -u
OUT=gpurun_out/r06b; mkdir -p $OUT
export TMPDIR=/tmp
VG_HOST_TRACE=1 STEP_ITERS=3 timeout 600 python tools/timing.py step > $OUT/step_trace.txt 2>&1
VG_HOST_TRACE=1 SCAN=sliced WORLD=8 RANK_SIM=1 REPS=2 timeout 600 python tools/micro/shard_pass.py > $OUT/shard_trace.txt 2>&1
SHORT=1 KERNELS="default fused" timeout 1200 bash tools/micro/parse_pmc.sh $OUT/parse_pmc > $OUT/parse_pmc.log 2>&1
timeout 300 python - > $OUT/out_aln.txt 2>&1 <<'PY'
import sys, time; sys.path.insert(0, '.')
from vclust_amd import api, synth
api.set_device(0)
codes, offsets, names, _ = synth.make_workload('phage-100k', 10000)
gs = api.GenomeSet.from_codes(codes, offsets, names); gs.to_device()
sizes, pairs = gs.kmer_shared(k=25, min_shared=20)
tasks = gs.align_tasks(gs.filter_pairs(sizes, pairs))
for rep in range(3):
    api.profile_enable(True); api.profile_reset(); t0 = time.perf_counter()
    st, rg = gs.lz_align(tasks, want_regions=True)
    print(rep, round((time.perf_counter() - t0) * 1e3, 1), 'ms', {e['name']: round(e['total_ms'], 2) for e in api.profile_get()}, len(rg))
PY
grep -v "^\[vg" $OUT/step_trace.txt | tail -4; tail -5 $OUT/out_aln.txt; tail -40 $OUT/parse_pmc.log

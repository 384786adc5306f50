#!/bin/bash
# placement: fresh processes, trials off, plain hipMalloc against VMM blocks (2 MiB-granular hipMemCreate chunks); CPU baseline threads
set -u
OUT=gpurun_out/r06e; mkdir -p $OUT; export TMPDIR=/tmp
python - > $OUT/cpus.txt 2>&1 <<'PY'
import os
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for p in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(p, open(p).read().strip())
    except OSError as e: print(p, e)
PY
for i in 1 2 3; do
  for v in hipmalloc vmm; do
    if [ $v = vmm ]; then export VG_ALLOC=vmm; else unset VG_ALLOC; fi
    echo -n "$v run $i: " >> $OUT/placement.txt
    VG_DEV_SWITCHES=1 timeout 300 python bench.py --steps 4 --warmup 2 --placement-trials 1 --no-cpu-baseline --no-cli-wall --no-other-workloads --no-out-aln 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['placement_trials'], d['roofline']['ms_per_step_by_scope'])" >> $OUT/placement.txt 2>&1
  done
done
unset VG_ALLOC
for i in 1 2; do
  echo -n "trials 4 run $i: " >> $OUT/placement.txt
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-cli-wall --no-other-workloads --no-out-aln 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['placement_trials'], d['roofline']['ms_per_step_by_scope'])" >> $OUT/placement.txt 2>&1
done
python - > $OUT/cpu_threads.txt 2>&1 <<'PY'
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle_lib as orc
from vclust_amd import synth
codes, offsets, names = synth.make_families(1000, 10, length=40000, seed=3)
for thr in (16, 32, 64, 256, 16):
    t0 = time.perf_counter(); rows, st, ran = orc.path_rows_mt(codes, offsets, threads=thr); dt = time.perf_counter() - t0
    cpu = orc.last_stage_cpu()
    print(thr, 'threads', round(dt, 2), 's', round(len(rows) / 2 / dt), 'pairs/s', st, {k: round(cpu[k] / st[k], 1) for k in st}, flush=True)
PY
cat $OUT/cpus.txt $OUT/placement.txt $OUT/cpu_threads.txt

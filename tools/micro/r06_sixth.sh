#!/bin/bash
set -u
OUT=gpurun_out/r06g; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "plan_differently or placement_trials_are or eight_processes or exchange_kept_masks" > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log
for seg in default 1 4; do
  if [ $seg = default ]; then unset VG_LZ_SEGMENTS; else export VG_LZ_SEGMENTS=$seg; fi
  echo -n "imgvr segments=$seg: " >> $OUT/imgvr_seg.txt
  VG_DEV_SWITCHES=1 bash tools/micro/bench_scopes.sh imgvr-10k 8 >> $OUT/imgvr_seg.txt 2>&1
done
unset VG_LZ_SEGMENTS
timeout 1500 python bench.py --workload contigs-1M --steps 2 --warmup 1 --no-cpu-baseline --no-cli-wall --no-other-workloads --no-out-aln > $OUT/bench_contigs-1M.json 2> $OUT/bench_contigs-1M.err
tail -3 $OUT/tests.log; cat $OUT/imgvr_seg.txt; python -c "
import json; d=json.loads(open('$OUT/bench_contigs-1M.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['pairs_per_step'], d['roofline']['ms_per_step_by_scope'], d['roofline']['frac'])"

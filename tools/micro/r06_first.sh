#!/bin/bash
# round 6, first GPU call: the new tests, the fused-slot probe A/B, the working-set sweep, one bench line, all-rank emulation
set -u
OUT=gpurun_out/r06a; mkdir -p $OUT
export TMPDIR=/tmp
python -c "from vclust_amd import _lib; _lib.load(); print('lib ok')" > $OUT/load.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "out_aln_one_parse or accuracy_against or eight_processes or many_regions or reference_goldens or two_ranks_write or exchange_kept_masks or variants_write" > $OUT/tests.log 2>&1
echo "tests rc $?" >> $OUT/tests.log
hipcc --offload-arch=gfx950 -O3 tools/micro/pool_sweep.hip -o /tmp/pool_sweep && timeout 300 /tmp/pool_sweep 2000 > $OUT/pool_sweep.txt 2>&1
for v in default fused default fused; do
  VG_DEV_SWITCHES=1 VG_LZ_INDEX=$v timeout 300 python tools/micro/parse_ab.py >> $OUT/parse_ab.txt 2>&1
done
timeout 900 python bench.py --steps 5 --warmup 2 --cpu-reps 2 > $OUT/bench.json 2> $OUT/bench.err
timeout 900 python tools/strong_scaling_sim.py > $OUT/scaling.txt 2>&1
tail -3 $OUT/tests.log; cat $OUT/pool_sweep.txt; cat $OUT/parse_ab.txt; grep "^==" $OUT/scaling.txt; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['ms_per_step_by_scope'], 'host', d['roofline']['host_ms_per_step'])
print('out_aln', d['out_aln'])
print('others', {k:(v.get('ms_per_step'), v.get('roofline',{}).get('frac') if v.get('roofline') else None, v.get('error')) for k,v in (d['other_workloads'] or {}).items()})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['runs_pairs_per_s'], d['cpu_baseline']['busy_threads_per_stage'], d['cpu_baseline']['stage_seconds'])
print('cli', d['cli_wall'].get('runs_total_s'))
"

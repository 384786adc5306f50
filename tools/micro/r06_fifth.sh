#!/bin/bash
# full GPU test suite, the driver's bench command, rocprofv3 stats + PMC traffic of the three one-GPU workloads
set -u
OUT=gpurun_out/r06f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/tests_full.log 2>&1; echo "rc $?" >> $OUT/tests_full.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
bash tools/collect_profiles.sh r06 phage-100k "" 3 > $OUT/prof_100k.log 2>&1
bash tools/collect_profiles.sh r06 phage-1k "" 10 > $OUT/prof_1k.log 2>&1
bash tools/collect_profiles.sh r06 imgvr-10k "" 5 > $OUT/prof_imgvr.log 2>&1
tail -3 $OUT/tests_full.log; python -c "
import json; d=json.loads(open('$OUT/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['ms_per_step_by_scope'], 'host', d['roofline']['host_ms_per_step'])
print('out_aln', {k:v for k,v in d['out_aln'].items() if k!='note'})
print('others', {k:(v.get('ms_per_step'), v.get('roofline',{}).get('frac') if v.get('roofline') else None, v.get('error')) for k,v in (d['other_workloads'] or {}).items()})
c=d['cpu_baseline']; print('cpu', c['value'], c['cores'], c['runs_pairs_per_s'], c['spread'], c['busy_threads_per_stage'], c['stage_seconds'], c['host_cpus'])
print('cli', d['cli_wall'].get('runs_total_s'), d['vs_cpu_baseline'])
"; ls gpurun_out/prof_r06_*/summary/
